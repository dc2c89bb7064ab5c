"""Synthetic perspective fields generated on the device (measurement / test helper).

Thin wrapper over gclm_synth_fields_grouped (include/gclm.h): image i depends on (seed, global index)
only, so any sharding of a batch sees identical data."""
from typing import Dict, Tuple

import torch

from . import _lib


def synth_fields(camera_model: str, B: int, H: int, W: int, device, seed: int = 0, first_index: int = 0,
                 noise: float = 0.02, group_size: int = 1, run: int = 0, run_stride: int = 0,
                 confidences: bool = True) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, torch.Tensor]:
    """Returns (data dict, gt_camera (B,8), gt_gravity (B,3)) on `device`."""
    device = torch.device(device)
    up = torch.empty((B, 2, H, W), device=device)
    lat = torch.empty((B, 1, H, W), device=device)
    upc = torch.empty((B, H, W), device=device) if confidences else None
    latc = torch.empty((B, H, W), device=device) if confidences else None
    gt_cam = torch.empty((B, 8), device=device)
    gt_grav = torch.empty((B, 3), device=device)
    with torch.cuda.device(device):
        rc = _lib.load().gclm_synth_fields_grouped(
            _lib.CAMERA_MODEL_IDS[camera_model], seed, first_index, B, H, W, noise, group_size, run, run_stride,
            up.data_ptr(), lat.data_ptr(), upc.data_ptr() if confidences else None,
            latc.data_ptr() if confidences else None, gt_cam.data_ptr(), gt_grav.data_ptr(),
            torch.cuda.current_stream(device).cuda_stream)
    if rc != 0:
        raise _lib.GclmError(f"gclm_synth_fields_grouped failed ({rc})")
    data = {"up_field": up, "latitude_field": lat}
    if confidences:
        data |= {"up_confidence": upc, "latitude_confidence": latc}
    return data, gt_cam, gt_grav
