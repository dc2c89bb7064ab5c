"""Camera tensors: pinhole, simple radial, radial and simple divisional models.

API-compatible with the reference's geocalib/camera.py (BaseCamera :18, Pinhole :522,
SimpleRadial :565, Radial :663, SimpleDivisional :789, camera_models :945): a `(..., 8)` tensor
`{w, h, fx, fy, cx, cy, k1, k2}` wrapped with accessors, resize / crop bookkeeping and the
(un)distortion maps.  Every model here is *radial*: distort(p) = p * s(r2) and
undistort(p) = p * t(r2) with r2 = |p|^2, so a model only supplies s, t and their derivatives
and all Jacobians follow in closed form (the reference falls back to torch.func.jacfwd for the
generic cases).  Host-side torch code: the per-pixel hot path is csrc/gclm_pass.hip.
"""
from typing import Dict, Tuple, Union

import torch
from torch.nn import functional as F

from .misc import TensorWrapper, autocast
from .utils import deg2rad, focal2fov, fov2focal


def _outer(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return a[..., :, None] * b[..., None, :]


def _one_of(pts, p2d):
    """The positional / `pts=` argument or its `p2d=` alias (exactly one of them)."""
    if (pts is None) == (p2d is None):
        raise TypeError("pass the points once: positionally, as pts= or as p2d=")
    return pts if p2d is None else p2d


def _eye_like(p2d: torch.Tensor) -> torch.Tensor:
    return torch.eye(2, device=p2d.device, dtype=p2d.dtype).expand(p2d.shape[:-1] + (2, 2))


class BaseCamera(TensorWrapper):
    """(..., {w, h, fx, fy, cx, cy, k1, k2}) camera parameters."""

    eps = 1e-3
    dist_range = (-0.7, 0.7)

    @autocast
    def __init__(self, data: torch.Tensor):
        assert data.shape[-1] in {6, 7, 8}, data.shape
        if data.shape[-1] != 8:
            data = torch.cat([data, data.new_zeros(data.shape[:-1] + (8 - data.shape[-1],))], -1)
        super().__init__(data)

    # ------------------------------------------------------------------ construction
    @classmethod
    def name(cls) -> str:
        raise NotImplementedError

    @classmethod
    def from_dict(cls, param_dict: Dict[str, torch.Tensor]) -> "BaseCamera":
        """Keys: height, width, one of {f, vfov}, optional cx, cy, scales, and one of
        {dist, k1_hat, k1[, k2]} (same precedence as the reference, camera.py:49-93)."""
        d = {k: v if isinstance(v, torch.Tensor) else torch.tensor(v) for k, v in param_dict.items()}
        param_dict.update(d)
        h, w = d["height"], d["width"]
        cx, cy = d.get("cx", w / 2), d.get("cy", h / 2)
        if "f" in d:
            f = d["f"]
        elif "vfov" in d:
            f = fov2focal(d["vfov"], h)
        else:
            raise ValueError("Focal length or vertical field of view must be provided.")
        if "dist" in d:
            k1 = d["dist"][..., (0,)]
            k2 = d["dist"][..., (1,)] if d["dist"].shape[-1] == 2 else torch.zeros_like(k1)
        elif "k1_hat" in d:
            k1 = d["k1_hat"] * (f / h) ** 2
            k2 = d.get("k2", torch.zeros_like(k1))
        else:
            k1 = d.get("k1", torch.zeros_like(f))
            k2 = d.get("k2", torch.zeros_like(f))
        fx, fy = f, f
        if "scales" in d:
            fx = fx * d["scales"][..., 0] / d["scales"][..., 1]
        return cls(torch.stack([w, h, fx, fy, cx, cy, k1, k2], dim=-1))

    def pinhole(self):
        """Same intrinsics without distortion."""
        return self.__class__(self._data[..., :6])

    def _rebuild(self, size=None, f=None, c=None, dist=None):
        size = self.size if size is None else size
        f = self.f if f is None else f
        c = self.c if c is None else c
        if dist is None:
            dist = self.dist if hasattr(self, "dist") else self.new_zeros(self.f.shape)
        return self.__class__(torch.cat([size, f, c, dist], -1))

    # ------------------------------------------------------------------ accessors
    @property
    def size(self) -> torch.Tensor:
        return self._data[..., :2]

    @property
    def f(self) -> torch.Tensor:
        return self._data[..., 2:4]

    @property
    def c(self) -> torch.Tensor:
        return self._data[..., 4:6]

    @property
    def vfov(self) -> torch.Tensor:
        return focal2fov(self.f[..., 1], self.size[..., 1])

    @property
    def hfov(self) -> torch.Tensor:
        return focal2fov(self.f[..., 0], self.size[..., 0])

    @property
    def K(self) -> torch.Tensor:
        K = self._data.new_zeros(self.shape + (3, 3))
        K[..., 0, 0], K[..., 1, 1] = self.f[..., 0], self.f[..., 1]
        K[..., 0, 2], K[..., 1, 2] = self.c[..., 0], self.c[..., 1]
        K[..., 2, 2] = 1
        return K

    # ------------------------------------------------------------------ parameter updates
    def update_focal(self, delta: torch.Tensor, as_log: bool = False):
        """f <- f + delta (or exp(log f + delta)), clamped to a vertical fov in [5, 150] degrees of
        the image HEIGHT for both axes; fx follows fy with the previous aspect (camera.py:136-152)."""
        f = torch.exp(torch.log(self.f) + delta) if as_log else self.f + delta
        ones = self.new_ones(self.shape[0])
        lo = fov2focal(ones * deg2rad(150), self.size[..., 1]).unsqueeze(-1)
        hi = fov2focal(ones * deg2rad(5), self.size[..., 1]).unsqueeze(-1)
        fy = torch.minimum(torch.maximum(f, lo), hi)[..., 1]
        return self._rebuild(f=torch.stack([fy * self.f[..., 0] / self.f[..., 1], fy], -1))

    def update_dist(self, delta: torch.Tensor, dist_range: Tuple[float, float] = None):
        lo, hi = dist_range or self.dist_range
        return self._rebuild(dist=(self.dist + self.new_ones(self.dist.shape) * delta).clamp(lo, hi))

    def scale(self, scales: Union[float, int, Tuple[Union[float, int]]]):
        """Intrinsics after resizing the image by `scales` (sx, sy)."""
        scales = (scales, scales) if isinstance(scales, (int, float)) else scales
        s = scales if isinstance(scales, torch.Tensor) else self.new_tensor(scales)
        return self._rebuild(size=self.size * s, f=self.f * s, c=self.c * s)

    def crop(self, pad: Tuple[float]):
        """Intrinsics after padding (+) / cropping (-) the image by `pad` pixels in total per axis."""
        pad = pad if isinstance(pad, torch.Tensor) else self.new_tensor(pad)
        return self._rebuild(size=self.size + pad.to(self.size), c=self.c + pad.to(self.c) / 2)

    def undo_scale_crop(self, data: Dict[str, torch.Tensor]):
        cam = self.crop(-data["crop_pad"]) if "crop_pad" in data else self
        return cam.scale(1.0 / data["scales"])

    # ------------------------------------------------------------------ radial model hooks
    def _k(self, i: int) -> torch.Tensor:
        return self._data[..., 6 + i][..., None, None]

    def _distort_scale(self, r2: torch.Tensor) -> torch.Tensor:
        """s(r2) with distort(p) = p * s(|p|^2)."""
        raise NotImplementedError

    def _distort_scale_dr2(self, r2: torch.Tensor) -> torch.Tensor:
        """ds/dr2."""
        raise NotImplementedError

    def _undistort_scale(self, r2: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def _undistort_scale_dr2(self, r2: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def check_valid(self, p2d: torch.Tensor) -> torch.Tensor:
        return p2d.new_ones(p2d.shape[:-1]).bool()

    # The reference names the argument `pts` on BaseCamera (camera.py:212,242) and `p2d` on the distortion models
    # (camera.py:611,631,712,737,829,863); one body serves them all here, so both keywords are accepted.
    def distort(self, pts: torch.Tensor = None, return_scale: bool = False, *, p2d: torch.Tensor = None):
        """Distort normalised coordinates; (scale, None) with return_scale."""
        pts = _one_of(pts, p2d)
        s = self._distort_scale((pts**2).sum(-1, keepdim=True))
        return (s, None) if return_scale else (pts * s, self.check_valid(pts))

    def undistort(self, pts: torch.Tensor = None, *, p2d: torch.Tensor = None):
        pts = _one_of(pts, p2d)
        return pts * self._undistort_scale((pts**2).sum(-1, keepdim=True)), self.check_valid(pts)

    def J_distort(self, p2d: torch.Tensor, wrt: str = "pts") -> torch.Tensor:
        r2 = (p2d**2).sum(-1, keepdim=True)
        if wrt == "pts":                      # d(p s)/dp = s I + 2 s' p p^T
            return self._distort_scale(r2)[..., None] * _eye_like(p2d) + \
                2 * self._distort_scale_dr2(r2)[..., None] * _outer(p2d, p2d)
        if wrt == "scale2pts":                # ds/dp = 2 s' p
            return 2 * self._distort_scale_dr2(r2) * p2d
        if wrt == "scale2dist":               # ds/dk_j  (..., N, nd)
            return self._hook_ddist("_distort_scale", p2d)
        raise NotImplementedError(f"Jacobian not implemented for wrt={wrt}")

    def J_undistort(self, p2d: torch.Tensor, wrt: str = "pts") -> torch.Tensor:
        r2 = (p2d**2).sum(-1, keepdim=True)
        if wrt == "pts":
            return self._undistort_scale(r2)[..., None] * _eye_like(p2d) + \
                2 * self._undistort_scale_dr2(r2)[..., None] * _outer(p2d, p2d)
        raise NotImplementedError(f"Jacobian not implemented for wrt={wrt}")

    @autocast
    def up_projection_offset(self, p2d: torch.Tensor) -> torch.Tensor:
        """ds/dp of the distortion scale (enters the distorted up field, perspective_fields.py:72)."""
        return self.J_distort(p2d, wrt="scale2pts")

    # Generic derivatives of the radial hooks by autograd (the models with closed forms override the public
    # methods; the reference's generic versions use torch.func.jacfwd per point, camera.py:216-298).
    def _per_point(self, p2d: torch.Tensor):
        """One camera row per point, distortion parameters as a differentiable leaf: (camera, r2, dist)."""
        n = p2d.shape[-2]
        data = self._data.reshape(-1, self._data.shape[-1])
        rows = data[:, None, :].expand(-1, n, -1).reshape(-1, data.shape[-1])
        nd = self.num_dist_params() if hasattr(self, "num_dist_params") else 0
        dist = rows[:, 6:6 + nd].detach().clone().requires_grad_(True)
        cam = self.__class__(torch.cat([rows[:, :6].detach(), dist, rows[:, 6 + nd:].detach()], -1))
        r2 = (p2d.detach() ** 2).sum(-1).reshape(-1, 1, 1)
        return cam, r2, dist

    def _hook_dr2(self, hook_name: str, r2: torch.Tensor) -> torch.Tensor:
        """d hook(r2) / d r2 (elementwise)."""
        x = r2.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            (g,) = torch.autograd.grad((getattr(self, hook_name)(x) + 0 * x).sum(), x)     # 0*x: constant hooks
        return g

    def _hook_ddist(self, hook_name: str, p2d: torch.Tensor) -> torch.Tensor:
        """d hook(|p|^2) / d dist per point: (..., N, num_dist)."""
        with torch.enable_grad():
            cam, r2, dist = self._per_point(p2d)
            (g,) = torch.autograd.grad(getattr(cam, hook_name)(r2).sum() + 0 * dist.sum(), dist)
        return g.reshape(*p2d.shape[:-1], -1)

    def J_up_projection_offset(self, p2d: torch.Tensor, wrt: str = "uv") -> torch.Tensor:
        """Jacobian of up_projection_offset(p) = 2 s'(r2) p: wrt "uv" (..., N, 2, 2), wrt "dist" (..., N, 2, nd)."""
        r2 = (p2d**2).sum(-1, keepdim=True)
        if wrt == "uv":        # 2 s' I + 4 s'' p p^T
            s2 = self._hook_dr2("_distort_scale_dr2", r2)
            return 2 * self._distort_scale_dr2(r2)[..., None] * _eye_like(p2d) + 4 * s2[..., None] * _outer(p2d, p2d)
        if wrt == "dist":      # 2 p ds'/dk_j
            return 2 * p2d[..., None] * self._hook_ddist("_distort_scale_dr2", p2d)[..., None, :]
        raise NotImplementedError(f"Jacobian not implemented for wrt={wrt}")

    # ------------------------------------------------------------------ projection chain
    @autocast
    def in_image(self, p2d: torch.Tensor):
        assert p2d.shape[-1] == 2
        return torch.all((p2d >= 0) & (p2d <= (self.size.unsqueeze(-2) - 1)), -1)

    @autocast
    def project(self, p3d: torch.Tensor):
        """(x, y, z) -> (x/z, y/z) and the z > eps visibility mask."""
        z = p3d[..., -1]
        return p3d[..., :-1] / z.clamp(min=self.eps).unsqueeze(-1), z > self.eps

    def J_project(self, p3d: torch.Tensor):
        x, y, z = p3d[..., 0], p3d[..., 1], p3d[..., 2].clamp(min=self.eps)
        zero = torch.zeros_like(z)
        return torch.stack([1 / z, zero, -x / z**2, zero, 1 / z, -y / z**2], -1).reshape(p3d.shape[:-1] + (2, 3))

    @autocast
    def denormalize(self, p2d: torch.Tensor) -> torch.Tensor:
        return p2d * self.f.unsqueeze(-2) + self.c.unsqueeze(-2)

    def J_denormalize(self):
        return torch.diag_embed(self.f)

    @autocast
    def normalize(self, p2d: torch.Tensor) -> torch.Tensor:
        return (p2d - self.c.unsqueeze(-2)) / self.f.unsqueeze(-2)

    def J_normalize(self, p2d: torch.Tensor, wrt: str = "f"):
        if wrt == "f":
            return torch.diag_embed(-(p2d - self.c.unsqueeze(-2)) / self.f.unsqueeze(-2) ** 2)
        if wrt == "pts":
            return torch.diag_embed(1 / self.f)
        raise NotImplementedError(f"Jacobian not implemented for wrt={wrt}")

    def pixel_coordinates(self) -> torch.Tensor:
        """(B, h*w, 2) integer pixel grid, row-major, x in [0, w), y in [0, h) (no half-pixel offset)."""
        w, h = (int(v) for v in self.size[0].round().tolist())
        ys, xs = torch.meshgrid(torch.arange(h, dtype=self.dtype, device=self.device),
                                torch.arange(w, dtype=self.dtype, device=self.device), indexing="ij")
        xy = torch.stack((xs, ys), -1).reshape(-1, 2)
        return xy.unsqueeze(0).expand(self.shape[0], -1, -1)

    @autocast
    def pixel_bearing_many(self, p3d: torch.Tensor) -> torch.Tensor:
        return F.normalize(p3d, dim=-1)

    @autocast
    def world2image(self, p3d: torch.Tensor):
        p2d, visible = self.project(p3d)
        p2d, mask = self.distort(p2d)
        p2d = self.denormalize(p2d)
        return p2d, visible & mask & self.in_image(p2d)

    @autocast
    def J_world2image(self, p3d: torch.Tensor):
        p2d, valid = self.project(p3d)
        J = self.J_denormalize().unsqueeze(-3) @ self.J_distort(p2d) @ self.J_project(p3d)
        return J, valid

    @autocast
    def image2world(self, p2d: torch.Tensor):
        p2d, valid = self.undistort(self.normalize(p2d))
        return torch.cat([p2d, p2d.new_ones(p2d.shape[:-1] + (1,))], -1), valid

    @autocast
    def J_image2world(self, p2d: torch.Tensor, wrt: str = "f"):
        if wrt == "dist":
            return self.J_undistort(self.normalize(p2d), wrt)
        if wrt == "f":
            return self.J_undistort(self.normalize(p2d), "pts") @ self.J_normalize(p2d, wrt)
        raise ValueError(f"Unknown wrt: {wrt}")

    def __repr__(self):
        return f"{self.__class__.__name__} {self.shape} {self.dtype} {self.device}"


class Pinhole(BaseCamera):
    """No distortion."""

    @classmethod
    def name(cls) -> str:
        return "pinhole"

    def distort(self, p2d: torch.Tensor, return_scale: bool = False):
        if return_scale:
            return p2d.new_ones(p2d.shape[:-1] + (1,))
        return p2d, p2d.new_ones((p2d.shape[0], 1)).bool()

    def undistort(self, pts: torch.Tensor):
        return pts, pts.new_ones((pts.shape[0], 1)).bool()

    def J_distort(self, p2d: torch.Tensor, wrt: str = "pts") -> torch.Tensor:
        if wrt == "pts":
            return _eye_like(p2d)
        raise ValueError(f"Unknown wrt: {wrt}")

    J_undistort = J_distort

    def J_up_projection_offset(self, p2d: torch.Tensor, wrt: str = "uv") -> torch.Tensor:
        if wrt == "uv":
            return torch.zeros(p2d.shape[:-1] + (2, 2), device=p2d.device, dtype=p2d.dtype)
        raise ValueError(f"Unknown wrt: {wrt}")


class _OneParam(BaseCamera):
    """Models with a single coefficient k1 (`dist` still spans both storage slots, as upstream)."""

    @property
    def dist(self) -> torch.Tensor:
        return self._data[..., 6:]

    @classmethod
    def num_dist_params(cls) -> int:
        return 1

    @property
    def k1(self) -> torch.Tensor:
        return self._data[..., 6]


class SimpleRadial(_OneParam):
    """s = 1 + k1 r2; inverse approximated by t = 1 - k1 r2 (Drap & Lefevre)."""

    @classmethod
    def name(cls) -> str:
        return "simple_radial"

    @property
    def k1_hat(self) -> torch.Tensor:
        return self.k1 / (self.f[..., 1] / self.size[..., 1]) ** 2

    def _distort_scale(self, r2):
        return 1 + self._k(0) * r2

    def _distort_scale_dr2(self, r2):
        return self._k(0).expand_as(r2)

    def _undistort_scale(self, r2):
        return 1 - self._k(0) * r2

    def _undistort_scale_dr2(self, r2):
        return (-self._k(0)).expand_as(r2)

    def J_distort(self, p2d, wrt: str = "pts"):
        if wrt == "scale2dist":
            return (p2d**2).sum(-1, keepdim=True)
        return super().J_distort(p2d, wrt)

    def J_undistort(self, p2d, wrt: str = "pts"):
        if wrt == "dist":
            return (-(p2d**2).sum(-1, keepdim=True) * p2d)[..., None]
        return super().J_undistort(p2d, wrt)

    def J_up_projection_offset(self, p2d, wrt: str = "uv"):
        if wrt == "uv":
            return 2 * self._k(0)[..., None] * _eye_like(p2d)
        if wrt == "dist":
            return (2 * p2d)[..., None]
        raise NotImplementedError(f"Jacobian not implemented for wrt={wrt}")


class Radial(BaseCamera):
    """s = 1 + k1 r2 + k2 r4; t = 1 - k1 r2 + (3 k1^2 - k2) r4."""

    @classmethod
    def name(cls) -> str:
        return "radial"

    @property
    def dist(self) -> torch.Tensor:
        return self._data[..., 6:8]

    @classmethod
    def num_dist_params(cls) -> int:
        return 2

    @property
    def k1(self) -> torch.Tensor:
        return self._data[..., 6]

    @property
    def k2(self) -> torch.Tensor:
        return self._data[..., 7]

    def _b(self):
        k1, k2 = self._k(0), self._k(1)
        return -k1, 3 * k1**2 - k2

    def _distort_scale(self, r2):
        return 1 + self._k(0) * r2 + self._k(1) * r2**2

    def _distort_scale_dr2(self, r2):
        return self._k(0) + 2 * self._k(1) * r2

    def _undistort_scale(self, r2):
        b1, b2 = self._b()
        return 1 + b1 * r2 + b2 * r2**2

    def _undistort_scale_dr2(self, r2):
        b1, b2 = self._b()
        return b1 + 2 * b2 * r2

    def J_distort(self, p2d, wrt: str = "pts"):
        if wrt == "scale2dist":
            r2 = (p2d**2).sum(-1, keepdim=True)
            return torch.cat([r2, r2**2], -1)
        return super().J_distort(p2d, wrt)

    def J_undistort(self, p2d, wrt: str = "pts"):
        if wrt == "dist":
            r2 = (p2d**2).sum(-1, keepdim=True)
            return torch.stack([(6 * r2**2 * self._k(0) - r2) * p2d, -(r2**2) * p2d], -1)
        return super().J_undistort(p2d, wrt)

    def J_up_projection_offset(self, p2d, wrt: str = "uv"):
        r2 = (p2d**2).sum(-1, keepdim=True)
        if wrt == "uv":
            return 8 * self._k(1)[..., None] * _outer(p2d, p2d) + \
                (2 * self._k(0) + 4 * self._k(1) * r2)[..., None] * _eye_like(p2d)
        if wrt == "dist":
            return torch.stack([2 * p2d, 4 * r2 * p2d], -1)
        raise NotImplementedError(f"Jacobian not implemented for wrt={wrt}")


class SimpleDivisional(_OneParam):
    """t = 1 / (1 + k1 r2); s = (1 - sqrt(1 - 4 k1 r2)) / (2 k1 r2)."""

    dist_range = (-3.0, 3.0)

    @classmethod
    def name(cls) -> str:
        return "simple_divisional"

    def _distort_scale(self, r2):
        k1 = self._k(0)
        num = 1 - torch.sqrt((1 - 4 * k1 * r2).clamp(min=0))
        den = 2 * k1 * r2
        return torch.where(den == 0, torch.ones_like(num), num / den.masked_fill(den == 0, 1e6))

    def _distort_scale_dr2(self, r2):
        k1 = self._k(0)
        t = torch.sqrt((1 - 4 * k1 * r2).clamp(min=1e-6))
        den = 2 * k1 * r2**2 * t
        return (2 * k1 * r2 - (1 - t) * t) / den.masked_fill(den == 0, 1e6)

    def _undistort_scale(self, r2):
        den = 1 + self._k(0) * r2
        return 1 / den.masked_fill(den == 0, 1e6)

    def _undistort_scale_dr2(self, r2):
        den = 1 + self._k(0) * r2
        den = den.masked_fill(den == 0, 1e6)
        return -self._k(0) / den**2

    def J_undistort(self, p2d, wrt: str = "pts"):
        if wrt == "dist":
            r2 = (p2d**2).sum(-1, keepdim=True)
            den = (1 + self._k(0) * r2) ** 2
            return (-r2 / den.masked_fill(den == 0, 1e6) * p2d)[..., None]
        return super().J_undistort(p2d, wrt)


camera_models = {
    "pinhole": Pinhole,
    "radial": Radial,
    "simple_radial": SimpleRadial,
    "simple_divisional": SimpleDivisional,
}
