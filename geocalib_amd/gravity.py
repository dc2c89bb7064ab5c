"""Unit gravity direction in the camera frame.

API-compatible with the reference's `geocalib.gravity.Gravity` (gravity.py:12-131): construction from a
3-vector or from (roll, pitch), the `x / y / z / roll / pitch / rp / R` views, the (roll, pitch) Jacobians
and the manifold update.  The angle conventions are the reference's:
    g = (-sin r cos p, -cos r cos p, sin p),   pitch = asin(g_z),
    roll = asin(-g_x / (sqrt(1 - g_z^2) + 1e-4)), mirrored to (-pi, pi] when g_y >= 0.
The same expressions run on the device in csrc/gclm_device.h (grav_roll, grav_pitch, tangent_rp, from_rp).
"""
import math

import torch
from torch.nn import functional as F

from .misc import EuclideanManifold, SphericalManifold, TensorWrapper, autocast
from .utils import rad2rotmat


def _as_tensor(v):
    return v if isinstance(v, torch.Tensor) else torch.tensor(v)


def _component(index: int, doc: str):
    return property(lambda self: self._data[..., index], doc=doc)


class Gravity(TensorWrapper):
    """(..., 3) unit vectors."""

    eps = 1e-4        # regulariser of the roll denominator

    @autocast
    def __init__(self, data: torch.Tensor) -> None:
        assert data.shape[-1] == 3, data.shape
        super().__init__(F.normalize(data, dim=-1))

    @classmethod
    def from_rp(cls, roll, pitch) -> "Gravity":
        """Gravity of a camera rolled by `roll` and pitched by `pitch` (radians)."""
        r, p = _as_tensor(roll), _as_tensor(pitch)
        horizontal = -torch.cos(p)
        return cls(torch.stack([torch.sin(r) * horizontal, torch.cos(r) * horizontal, torch.sin(p)], dim=-1))

    vec3d = property(lambda self: self._data, doc="The (..., 3) unit vector itself.")
    x = _component(0, "First component.")
    y = _component(1, "Second component.")
    z = _component(2, "Third component.")

    # ---- angles
    def _angles(self):
        gx, gy, gz = self._data.unbind(-1)
        front = torch.asin(-gx / (torch.sqrt(1 - gz * gz) + self.eps))
        back = -front - math.pi * torch.sign(gx)          # pointing "up" in the image: mirror
        return torch.where(gy < 0, front, back), torch.asin(gz)

    @property
    def roll(self) -> torch.Tensor:
        return self._angles()[0]

    @property
    def pitch(self) -> torch.Tensor:
        return self._angles()[1]

    @property
    def rp(self) -> torch.Tensor:
        return torch.stack(self._angles(), dim=-1)

    @property
    def R(self) -> torch.Tensor:
        roll, pitch = self._angles()
        return rad2rotmat(roll=roll, pitch=pitch)

    # ---- derivatives
    def J_rp(self) -> torch.Tensor:
        """d(vec3d)/d(roll, pitch) as (..., 3, 2): columns (-cr cp, sr cp, 0) and (sr sp, cr sp, cp)."""
        roll, pitch = self._angles()
        sr, cr, sp, cp = roll.sin(), roll.cos(), pitch.sin(), pitch.cos()
        d_roll = torch.stack([-cr * cp, sr * cp, torch.zeros_like(cp)], dim=-1)
        d_pitch = torch.stack([sr * sp, cr * sp, cp], dim=-1)
        return torch.stack([d_roll, d_pitch], dim=-1)

    def J_roll(self) -> torch.Tensor:
        return self.J_rp()[..., 0]

    def J_pitch(self) -> torch.Tensor:
        return self.J_rp()[..., 1]

    def J_R(self) -> torch.Tensor:
        raise NotImplementedError

    # ---- manifold step
    def update(self, delta: torch.Tensor, spherical: bool = False) -> "Gravity":
        """Apply a 2-vector step: on the sphere (Householder retraction) or on (roll, pitch)."""
        if spherical:
            return type(self)(SphericalManifold.plus(self._data, delta))
        stepped = EuclideanManifold.plus(self.rp, delta)
        return self.from_rp(*stepped.unbind(-1))

    def J_update(self, spherical: bool = False) -> torch.Tensor:
        manifold = SphericalManifold if spherical else EuclideanManifold
        return manifold.J_plus(self._data)

    def __repr__(self):
        return f"{type(self).__name__} {self.shape} {self.dtype} {self.device}"
