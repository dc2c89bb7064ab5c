"""Tensor containers and manifold helpers of the calibration API.

API-compatible with the reference's geocalib/misc.py (TensorWrapper :50-154, EuclideanManifold
:157-168, SphericalManifold :171-259, J_vecnorm :263, J_focal2fov :285, J_up_projection :291) so
`Camera` / `Gravity` objects returned by the HIP optimiser behave like the reference's.  These
are host-side torch helpers; the per-pixel math of the hot path lives in csrc/gclm_pass.hip.
"""
import functools

import numpy as np
import torch


def autocast(method):
    """Let a wrapper method accept numpy arrays (cast to the wrapper's dtype / device)."""

    @functools.wraps(method)
    def inner(self, *args):
        is_wrapper = isinstance(self, TensorWrapper)
        if not is_wrapper and not (isinstance(self, type) and issubclass(self, TensorWrapper)):
            raise ValueError(self)
        dev, dt = torch.device("cpu"), None
        if is_wrapper and self._data is not None:
            dev, dt = self.device, self.dtype
        args = [torch.from_numpy(a).to(device=dev, dtype=dt) if isinstance(a, np.ndarray) else a for a in args]
        return method(self, *args)

    return inner


class TensorWrapper:
    """A tensor whose last dimension is a record; everything before it is the batch shape.

    Same surface as the reference's TensorWrapper (misc.py:50-154).  Tensor methods that keep the record
    dimension (`to`, `cpu`, `float`, `unsqueeze`, ...) are forwarded to the wrapped tensor and re-wrapped;
    the `new_*` factories are forwarded as they are."""

    _data = None
    _REWRAPPED = ("to", "cpu", "cuda", "pin_memory", "float", "double", "detach", "unsqueeze", "squeeze")
    _FACTORIES = ("new_tensor", "new_zeros", "new_ones", "new_full", "new_empty")

    @autocast
    def __init__(self, data: torch.Tensor):
        self._data = data

    shape = property(lambda self: self._data.shape[:-1], doc="Batch shape (record dimension dropped).")
    device = property(lambda self: self._data.device)
    dtype = property(lambda self: self._data.dtype)

    def __getitem__(self, index):
        return type(self)(self._data[index])

    def __setitem__(self, index, item):
        self._data[index] = item.data

    def numpy(self):
        return self._data.detach().cpu().numpy()

    @classmethod
    def stack(cls, objects, dim=0, *, out=None):
        return cls(torch.stack([o._data for o in objects], dim=dim, out=out))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is torch.stack:
            return cls.stack(*args, **(kwargs or {}))
        return NotImplemented


def _rewrapped(name):
    def method(self, *args, **kwargs):
        return type(self)(getattr(self._data, name)(*args, **kwargs))
    method.__name__ = name
    method.__doc__ = f"`torch.Tensor.{name}` on the wrapped tensor, wrapped again."
    return method


def _factory(name):
    def method(self, *args, **kwargs):
        return getattr(self._data, name)(*args, **kwargs)
    method.__name__ = name
    method.__doc__ = f"`torch.Tensor.{name}` with the wrapper's dtype and device."
    return method


for _name in TensorWrapper._REWRAPPED:
    setattr(TensorWrapper, _name, _rewrapped(_name))
for _name in TensorWrapper._FACTORIES:
    setattr(TensorWrapper, _name, _factory(_name))


class EuclideanManifold:
    """x [+] delta = x + delta."""

    @staticmethod
    def J_plus(x: torch.Tensor) -> torch.Tensor:
        return torch.eye(x.shape[-1]).to(x)

    @staticmethod
    def plus(x: torch.Tensor, delta: torch.Tensor) -> torch.Tensor:
        return x + delta


class SphericalManifold:
    """Unit-sphere [+] through a Householder reflection with the LAST component as pivot
    (Hertzberg et al. B.2; Golub & Van Loan Alg. 5.1.1).  Same numerics as the device code
    (csrc/gclm_update.hip: householder / tangent_sphere / grav_update)."""

    @staticmethod
    def householder_vector(x: torch.Tensor):
        """v (v[-1] = 1) and beta with (I - beta v v^T) x = |x| e_n."""
        sigma = (x[..., :-1] ** 2).sum(-1)
        pivot = x[..., -1]
        norm = x.norm(dim=-1)
        sigma = torch.where(sigma < 1e-7, sigma + 1e-7, sigma)
        vpiv = torch.where(pivot < 0, pivot - norm, -sigma / (pivot + norm))
        beta = 2 * vpiv**2 / (sigma + vpiv**2)
        v = torch.cat([x[..., :-1] / vpiv[..., None], torch.ones_like(vpiv)[..., None]], -1)
        return v, beta

    @staticmethod
    def apply_householder(y, v, beta):
        return y - v * (beta * (v * y).sum(-1))[..., None]

    @classmethod
    def J_plus(cls, x: torch.Tensor) -> torch.Tensor:
        """d(x [+] delta)/d(delta) at delta = 0: the first n-1 columns of the reflection."""
        v, beta = cls.householder_vector(x)
        n = x.shape[-1]
        Hm = torch.eye(n).to(x) - beta[..., None, None] * v[..., :, None] * v[..., None, :]
        return Hm[..., :-1]

    @classmethod
    def plus(cls, x: torch.Tensor, delta: torch.Tensor) -> torch.Tensor:
        eps = 1e-7
        nx = x.norm(dim=-1, keepdim=True)
        nd = delta.norm(dim=-1, keepdim=True)
        nd_safe = torch.where(nd < eps, nd + eps, nd)
        sinc = torch.where(nd < eps, torch.ones_like(nd), nd_safe.sin() / nd_safe)
        exp_delta = torch.cat([sinc * delta, nd.cos()], -1)
        v, beta = cls.householder_vector(x)
        return nx * cls.apply_householder(exp_delta, v, beta)


def J_vecnorm(vec: torch.Tensor) -> torch.Tensor:
    """Jacobian of vec / |vec|: I/|v| - v v^T/|v|^3, shape (..., D, D)."""
    n = vec.norm(dim=-1, keepdim=True).unsqueeze(-1)
    if (n == 0).any():
        n = n + 1e-6
    eye = torch.eye(vec.shape[-1], device=vec.device, dtype=vec.dtype)
    return eye / n - vec[..., :, None] * vec[..., None, :] / n**3


def J_focal2fov(focal: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """d(2 atan(h / 2f))/df."""
    return -4 * h / (4 * focal**2 + h**2)


def J_up_projection(uv: torch.Tensor, abc: torch.Tensor, wrt: str = "uv") -> torch.Tensor:
    """Jacobian of the projected up vector (a, b) - c (u, v) wrt "uv" (2x2) or "abc" (2x3)."""
    if wrt == "uv":
        eye = torch.eye(2, device=uv.device, dtype=uv.dtype).expand(uv.shape[:-1] + (2, 2))
        return -abc[..., 2][..., None, None, None] * eye
    if wrt == "abc":
        J = uv.new_zeros(uv.shape[:-1] + (2, 3))
        J[..., 0, 0] = J[..., 1, 1] = 1
        J[..., :, 2] = -uv
        return J
    raise ValueError(f"Unknown wrt: {wrt}")
