"""Perspective fields of a camera (host-side torch renderer).

Forward model of the reference's geocalib/perspective_fields.py (get_up_field :47, get_latitude_field
 :185, get_perspective_field :278), written on the radial-model hooks of
geocalib_amd.camera.  Used to render ground-truth fields; the optimiser does NOT
call this module -- residuals and Jacobians are evaluated per pixel in csrc/gclm_pass.hip.

J_up_field / J_latitude_field / J_perspective_field (:84, :214, :323) return the per-pixel Jacobian fields
from the same HIP pixel code through gclm_jacobian_fields (device tensors only, no CPU fallback).
"""
from typing import Tuple

import torch
from torch.nn import functional as F

from . import _lib
from .camera import BaseCamera
from .gravity import Gravity


def _batched(camera: BaseCamera, gravity: Gravity):
    camera = camera.unsqueeze(0) if len(camera.shape) == 0 else camera
    gravity = gravity.unsqueeze(0) if len(gravity.shape) == 0 else gravity
    w, h = (int(v) for v in camera.size[0].round().tolist())
    return camera, gravity, h, w


def get_horizon_line(camera: BaseCamera, gravity: Gravity, relative: bool = True) -> torch.Tensor:
    """Heights at which the horizon meets the left and right image border (reference: perspective_fields.py:18-44):
    the horizon passes through the projection m of the direction R e_z and is tilted by the roll.  `relative`
    divides by the image height.  One (unbatched or single-element) camera.  (The reference's version cannot run: it
    indexes the batch dimension it has just added, :29-36; this follows its definition.)"""
    camera, gravity, _, _ = _batched(camera, gravity)
    m = (camera.K @ gravity.R)[0, :, 2]                       # K R e_z
    mx, my = m[0] / m[2], m[1] / m[2]
    slope = torch.tan(gravity.roll)[0]
    ends = torch.stack([my + mx * slope, my - (camera.size[0, 0] - mx) * slope])
    return ends / camera.size[0, 1] if relative else ends


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


def _jacobian_fields(camera: BaseCamera, gravity: Gravity, spherical: bool, log_focal: bool, want_up: bool,
                     want_lat: bool):
    camera, gravity, h, w = _batched(camera, gravity)
    cam, grav = camera._data, gravity._data
    if not (cam.is_cuda and grav.is_cuda):
        raise RuntimeError("geocalib_amd Jacobian fields need HIP device tensors: they are evaluated by the HIP "
                           "extension (gclm_jacobian_fields), there is no CPU fallback")
    lead = cam.shape[:-1]
    cam = cam.detach().reshape(-1, 8).to(torch.float32).contiguous()
    grav = grav.detach().reshape(-1, 3).to(torch.float32).contiguous()
    assert cam.shape[0] == grav.shape[0], (cam.shape, grav.shape)
    B = cam.shape[0]
    model = _lib.CAMERA_MODEL_IDS[camera.name()]
    P = 3 + (camera.num_dist_params() if hasattr(camera, "num_dist_params") else 0)
    J_up = cam.new_empty((B, h, w, 2, P)) if want_up else None
    J_lat = cam.new_empty((B, h, w, 1, P)) if want_lat else None
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(cam.device):
        rc = _lib.load().gclm_jacobian_fields(model, cam.data_ptr(), grav.data_ptr(), B, h, w, int(spherical),
                                              int(log_focal), ptr(J_up), ptr(J_lat),
                                              torch.cuda.current_stream(cam.device).cuda_stream)
    if rc != 0:
        raise _lib.GclmError(f"gclm_jacobian_fields failed ({rc})")
    shape = lambda t: None if t is None else t.reshape(*lead, *t.shape[1:])  # noqa: E731
    return shape(J_up), shape(J_lat)


def J_up_field(camera: BaseCamera, gravity: Gravity, spherical: bool = False, log_focal: bool = False) -> torch.Tensor:
    """Jacobian of the up field wrt (gravity[2], focal[, dist]): (..., h, w, 2, P)   (perspective_fields.py:84-182)."""
    return _jacobian_fields(camera, gravity, spherical, log_focal, True, False)[0]


def J_latitude_field(camera: BaseCamera, gravity: Gravity, spherical: bool = False,
                     log_focal: bool = False) -> torch.Tensor:
    """Jacobian of sin(latitude) wrt (gravity[2], focal[, dist]): (..., h, w, 1, P)   (perspective_fields.py:214-275)."""
    return _jacobian_fields(camera, gravity, spherical, log_focal, False, True)[1]


def J_perspective_field(camera: BaseCamera, gravity: Gravity, use_up: bool = True, use_latitude: bool = True,
                        spherical: bool = False, log_focal: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(J_up (..., h, w, 2, P), J_lat (..., h, w, 1, P)); a disabled field is zeros (perspective_fields.py:323-365)."""
    assert use_up or use_latitude, "At least one of use_up or use_latitude must be True."
    J_up, J_lat = _jacobian_fields(camera, gravity, spherical, log_focal, True, True)
    return (J_up if use_up else torch.zeros_like(J_up)), (J_lat if use_latitude else torch.zeros_like(J_lat))
