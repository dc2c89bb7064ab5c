"""Unit gravity direction in the camera frame (API of the reference's geocalib/gravity.py:12-131)."""
import math

import torch
from torch.nn import functional as F

from .misc import EuclideanManifold, SphericalManifold, TensorWrapper, autocast
from .utils import rad2rotmat


class Gravity(TensorWrapper):
    """(..., 3) unit vectors; roll / pitch are derived views."""

    eps = 1e-4

    @autocast
    def __init__(self, data: torch.Tensor) -> None:
        assert data.shape[-1] == 3, data.shape
        super().__init__(F.normalize(data, dim=-1))

    @classmethod
    def from_rp(cls, roll, pitch) -> "Gravity":
        roll = roll if isinstance(roll, torch.Tensor) else torch.tensor(roll)
        pitch = pitch if isinstance(pitch, torch.Tensor) else torch.tensor(pitch)
        cp = pitch.cos()
        return cls(torch.stack([-roll.sin() * cp, -roll.cos() * cp, pitch.sin()], dim=-1))

    @property
    def vec3d(self) -> torch.Tensor:
        return self._data

    @property
    def x(self) -> torch.Tensor:
        return self._data[..., 0]

    @property
    def y(self) -> torch.Tensor:
        return self._data[..., 1]

    @property
    def z(self) -> torch.Tensor:
        return self._data[..., 2]

    @property
    def roll(self) -> torch.Tensor:
        """Roll in (-pi, pi]; the eps in the denominator is the reference's (gravity.py:65)."""
        r = torch.asin(-self.x / (torch.sqrt(1 - self.z**2) + self.eps))
        return torch.where(self.y < 0, r, -r - math.pi * torch.sign(self.x))

    @property
    def pitch(self) -> torch.Tensor:
        return torch.asin(self.z)

    @property
    def rp(self) -> torch.Tensor:
        return torch.stack([self.roll, self.pitch], dim=-1)

    def J_roll(self) -> torch.Tensor:
        r, p = self.roll, self.pitch
        return torch.stack([-r.cos() * p.cos(), r.sin() * p.cos(), torch.zeros_like(r)], -1)

    def J_pitch(self) -> torch.Tensor:
        r, p = self.roll, self.pitch
        return torch.stack([r.sin() * p.sin(), r.cos() * p.sin(), p.cos()], -1)

    def J_rp(self) -> torch.Tensor:
        """d(vec3d)/d(roll, pitch), shape (..., 3, 2)."""
        return torch.stack([self.J_roll(), self.J_pitch()], dim=-1)

    @property
    def R(self) -> torch.Tensor:
        return rad2rotmat(roll=self.roll, pitch=self.pitch)

    def J_R(self) -> torch.Tensor:
        raise NotImplementedError

    def update(self, delta: torch.Tensor, spherical: bool = False) -> "Gravity":
        if spherical:
            return self.__class__(SphericalManifold.plus(self.vec3d, delta))
        rp = EuclideanManifold.plus(self.rp, delta)
        return self.from_rp(rp[..., 0], rp[..., 1])

    def J_update(self, spherical: bool = False) -> torch.Tensor:
        return (SphericalManifold if spherical else EuclideanManifold).J_plus(self.vec3d)

    def __repr__(self):
        return f"{self.__class__.__name__} {self.shape} {self.dtype} {self.device}"
