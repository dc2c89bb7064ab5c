"""`GeoCalib.calibrate()` on MI355X: fields from a caller-supplied network, LM on the HIP path.

Mirrors the reference's geocalib/extractor.py (GeoCalib :15, _post_process :51-69, calibrate :72-127).
The CNN (MSCAN backbone + decoders, geocalib/geocalib.py:92-121) is out of scope of this package
and stays plain PyTorch-ROCm: pass any callable `field_model(image_dict) -> dict` that returns
`up_field (B,2,h,w)`, `latitude_field (B,1,h,w)` and optionally the two confidences -- e.g. the
upstream `geocalib.geocalib.GeoCalib` with its `.optimizer` replaced by this package's LMOptimizer,
or its `perspective_decoder` stack.  Image resizing (kornia in the reference) is the caller's too:
`preprocess(img) -> {"image", "scales", optional "crop_pad"}`.
"""
from typing import Callable, Dict, Optional

import torch
import torch.nn as nn
from torch.nn.functional import interpolate

from .camera import BaseCamera
from .fields import upsample_fields
from .lm_optimizer import LMOptimizer


def default_preprocess(img: torch.Tensor, resize: int = 320, edge_divisible_by: int = 32) -> Dict[str, torch.Tensor]:
    """Short side to `resize` px (antialiased bilinear), centre-crop to multiples of `edge_divisible_by`.
    Same bookkeeping keys as the reference's ImagePreprocessor (geocalib/utils.py:68-160)."""
    h, w = img.shape[-2:]
    s = resize / min(h, w)
    nh, nw = int(round(h * s)), int(round(w * s))
    out = interpolate(img, size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
    scales = torch.tensor([nw / w, nh / h], dtype=img.dtype, device=img.device)
    ch, cw = nh // edge_divisible_by * edge_divisible_by, nw // edge_divisible_by * edge_divisible_by
    top, left = (nh - ch) // 2, (nw - cw) // 2
    out = out[..., top: top + ch, left: left + cw]
    crop_pad = torch.tensor([cw - nw, ch - nh], dtype=img.dtype, device=img.device)
    return {"image": out, "scales": scales, "crop_pad": crop_pad}


class GeoCalib(nn.Module):
    """Single-image (or shared-intrinsics multi-image) calibration front-end."""

    def __init__(self, field_model: Callable, preprocess: Callable = default_preprocess, paced_launches: int = 3,
                 **optimizer_options):
        """`paced_launches` (no reference counterpart; include/gclm.h: gclm_set_paced_launches): calibrate() reads the camera
        on the host right after the solve (_post_process), so a single-image solve may pace its launches against the
        device's early stop -- the call then blocks for about the LM loop (~0.15 ms) instead of returning at once, and the
        result is there ~20 % earlier.  Pass 0 in a serving loop that keeps other streams busy while it calibrates: the
        wait is bounded (2 ms, then the handle stops pacing for its next 64 solves), but it is a wait."""
        super().__init__()
        self.field_model = field_model
        self.preprocess = preprocess
        self.optimizer = LMOptimizer({**optimizer_options})
        self.optimizer.eval()              # the reference puts its model in eval mode at construction (extractor.py:43): a
                                           # freshly built GeoCalib returns the uncertainties; .train() / .eval() on the wrapper
                                           # propagate to the optimiser like to any child module
        self.optimizer.paced_launches = int(paced_launches)

    def _post_process(self, camera: BaseCamera, img_data: Dict[str, torch.Tensor], out: Dict[str, torch.Tensor]):
        """Undo scaling / cropping and bring the fields back to the input resolution."""
        camera = camera.undo_scale_crop(img_data)
        w, h = (int(v) for v in camera.size[0].round().tolist())
        for k in ("latitude_field", "up_field", "up_confidence", "latitude_confidence"):
            if k in out:      # bilinear, align_corners=False (extractor.py:60-63), one HIP launch per tensor
                out[k] = upsample_fields(out[k], (h, w))
        zero = camera.new_zeros(camera.f.shape[0])
        out["focal_uncertainty"] = out.get("focal_uncertainty", zero) * (1.0 / img_data["scales"])[1]
        return camera, out

    @torch.no_grad()
    def calibrate(self, img: torch.Tensor, camera_model: str = "pinhole",
                  priors: Optional[Dict[str, torch.Tensor]] = None,
                  shared_intrinsics: bool = False) -> Dict[str, torch.Tensor]:
        """img: (C,H,W) or (B,C,H,W) in [0,1] RGB; B must be 1 unless shared_intrinsics."""
        if len(img.shape) == 3:
            img = img[None]
        if not shared_intrinsics:
            assert len(img.shape) == 4 and img.shape[0] == 1
        img_data = self.preprocess(img)
        priors = priors or {}
        prior_values = {}
        if (pf := priors.get("focal")) is not None:
            pf = pf[None] if len(pf.shape) == 0 else pf
            prior_values["prior_focal"] = pf * img_data["scales"][1]
        if "gravity" in priors:
            pg = priors["gravity"]
            prior_values["prior_gravity"] = pg[None] if len(pg.shape) == 0 else pg
        self.optimizer.set_camera_model(camera_model)
        self.optimizer.shared_intrinsics = shared_intrinsics
        out = dict(self.field_model(img_data))
        out |= {k: img_data[k] for k in ("image", "scales") if k in img_data}
        out |= prior_values
        out |= self.optimizer(out)
        camera, out = self._post_process(out["camera"], img_data, out)
        res = {"camera": camera, "gravity": out["gravity"]}
        if "covariance" in out:
            res["covariance"] = out["covariance"]
        for tag in ("field", "confidence", "uncertainty"):
            res |= {k: out[k] for k in out if tag in k}
        return res
