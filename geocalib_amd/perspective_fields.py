"""Perspective fields of a camera (host-side torch renderer).

Forward model of the reference's geocalib/perspective_fields.py (get_up_field :47, get_latitude_field
:185, get_perspective_field :278, get_horizon_line :18), written on the radial-model hooks of
geocalib_amd.camera.  Used to render ground-truth / visualisation fields; the optimiser does NOT
call this module -- residuals and Jacobians are evaluated per pixel in csrc/gclm_pass.hip.
"""
from typing import Tuple

import torch
from torch.nn import functional as F

from .camera import BaseCamera
from .gravity import Gravity


def _batched(camera: BaseCamera, gravity: Gravity):
    camera = camera.unsqueeze(0) if len(camera.shape) == 0 else camera
    gravity = gravity.unsqueeze(0) if len(gravity.shape) == 0 else gravity
    w, h = (int(v) for v in camera.size[0].round().tolist())
    return camera, gravity, h, w


def get_up_field(camera: BaseCamera, gravity: Gravity, normalize: bool = True) -> torch.Tensor:
    """Projected up direction per pixel, (..., h, w, 2): p = (a, b) - c (u, v), pushed through the
    distortion differential  s p + (ds/duv . ... ) i.e. q = s p + (off . uv-weighted p)."""
    camera, gravity, h, w = _batched(camera, gravity)
    uv = camera.normalize(camera.pixel_coordinates())
    abc = gravity.vec3d
    up = abc[..., None, :2] - abc[..., 2, None, None] * uv
    if hasattr(camera, "dist"):
        scale = camera.distort(uv, return_scale=True)[0]
        off = camera.up_projection_offset(uv)
        up = scale * up + off * (uv * up).sum(-1, keepdim=True)
    if normalize:
        up = F.normalize(up, dim=-1)
    return up.reshape(camera.shape[0], h, w, 2)


def get_latitude_field(camera: BaseCamera, gravity: Gravity) -> torch.Tensor:
    """Latitude (radians) of every pixel's viewing ray wrt the gravity direction, (..., h, w, 1)."""
    camera, gravity, h, w = _batched(camera, gravity)
    rays = camera.pixel_bearing_many(camera.image2world(camera.pixel_coordinates())[0])
    s = (rays * gravity.vec3d[..., None, :]).sum(-1)
    eps = 1e-6
    return torch.asin(s.clamp(min=-1 + eps, max=1 - eps)).reshape(camera.shape[0], h, w, 1)


def get_perspective_field(camera: BaseCamera, gravity: Gravity, use_up: bool = True,
                          use_latitude: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(up (..., 2, h, w), latitude (..., 1, h, w)); a disabled field is returned as zeros."""
    assert use_up or use_latitude, "At least one of use_up or use_latitude must be True."
    camera, gravity, h, w = _batched(camera, gravity)
    B = camera.shape[0]
    up = get_up_field(camera, gravity).permute(0, 3, 1, 2) if use_up else camera.new_zeros((B, 2, h, w))
    lat = get_latitude_field(camera, gravity).permute(0, 3, 1, 2) if use_latitude else camera.new_zeros((B, 1, h, w))
    return up, lat


def get_horizon_line(camera: BaseCamera, gravity: Gravity, relative: bool = True) -> torch.Tensor:
    """Left / right image-border intersections of the horizon (fractions of the height if relative)."""
    camera = camera.unsqueeze(0) if len(camera.shape) == 0 else camera
    gravity = gravity.unsqueeze(0) if len(gravity.shape) == 0 else gravity
    mid = camera.K @ gravity.R @ camera.new_tensor([0, 0, 1])
    mid = mid[:2] / mid[2]
    t = torch.tan(gravity.roll)
    horizon = camera.new_tensor([mid[1] + mid[0] * t, mid[1] - (camera.size[0] - mid[0]) * t])
    return horizon / camera.size[1] if relative else horizon
