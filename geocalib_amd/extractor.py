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
from .fields import upsample_fields_multi
from .lm_optimizer import LMOptimizer


_PRE_CACHE: Dict[tuple, tuple] = {}     # (h, w, resize, edge, device, dtype) -> the bookkeeping tensors of default_preprocess


def default_preprocess(img: torch.Tensor, resize: int = 320, edge_divisible_by: int = 32) -> Dict[str, torch.Tensor]:
    """Short side to `resize` px (antialiased bilinear), centre-crop to multiples of `edge_divisible_by`.
    Same bookkeeping keys as the reference's ImagePreprocessor (geocalib/utils.py:68-160), plus `_host`: the same numbers
    as Python floats (they only depend on the image SHAPE), which lets `_post_process` skip a device-to-host read and a
    dozen tiny tensor kernels.  The bookkeeping numbers are cached on the device per input shape (no host-to-device copy per
    call); every call returns its own copy of them."""
    h, w = img.shape[-2:]
    s = resize / min(h, w)
    nh, nw = int(round(h * s)), int(round(w * s))
    out = interpolate(img, size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
    ch, cw = nh // edge_divisible_by * edge_divisible_by, nw // edge_divisible_by * edge_divisible_by
    top, left = (nh - ch) // 2, (nw - cw) // 2
    out = out[..., top: top + ch, left: left + cw]
    key = (h, w, resize, edge_divisible_by, img.device, img.dtype)
    cached = _PRE_CACHE.get(key)
    if cached is None:
        if len(_PRE_CACHE) > 64:
            _PRE_CACHE.clear()
        cached = _PRE_CACHE[key] = torch.tensor([nw / w, nh / h, cw - nw, ch - nh], dtype=img.dtype, device=img.device)
    # the caller gets its OWN copy (one 16-byte device copy, no host-to-device transfer): an in-place edit of
    # data["scales"] must not reach the next image of this shape
    own = cached.clone()
    scales, crop_pad = own[:2], own[2:]
    return {"image": out, "scales": scales, "crop_pad": crop_pad,
            "_host": {"scales": (nw / w, nh / h), "crop_pad": (float(cw - nw), float(ch - nh)), "size": (int(w), int(h))}}


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

    _UNDO_CACHE: Dict[tuple, tuple] = {}

    def _post_process(self, camera: BaseCamera, img_data: Dict[str, torch.Tensor], out: Dict[str, torch.Tensor]):
        """Undo scaling / cropping and bring the fields back to the input resolution (extractor.py:51-69).

        With the host-side bookkeeping of `default_preprocess` (`img_data["_host"]`) this is two launches and no
        device-to-host read: ONE fused multiply-add on the camera rows -- the operations of `crop(-crop_pad).scale(1/scales)`
        in their order, so the same bits -- and ONE bilinear launch for all four tensors.  A caller's own preprocess
        without `_host` takes the reference's route (`undo_scale_crop`, the size read back from the camera)."""
        host = img_data.get("_host")
        tensors = [k for k in ("latitude_field", "up_field", "up_confidence", "latitude_confidence") if k in out]
        if host is not None and "crop_pad" in img_data:
            dev = camera._data.device
            key = (host["scales"], host["crop_pad"], dev)
            cached = self._UNDO_CACHE.get(key)
            if cached is None:
                inv = 1.0 / torch.tensor(host["scales"], dtype=torch.float32)            # float32 like `1.0 / data["scales"]`
                px, py = host["crop_pad"]
                add = torch.tensor([-px, -py, 0.0, 0.0, -px / 2, -py / 2, 0.0, 0.0], dtype=torch.float32)
                mul = torch.stack([inv[0], inv[1], inv[0], inv[1], inv[0], inv[1], inv.new_tensor(1.0), inv.new_tensor(1.0)])
                if len(self._UNDO_CACHE) > 64:
                    self._UNDO_CACHE.clear()
                cached = self._UNDO_CACHE[key] = (add.to(dev), mul.to(dev), float(inv[1]))
            add, mul, inv_sy = cached
            camera = camera.__class__((camera._data + add) * mul)
            w, h = host["size"]
            zero_or_fu = out.get("focal_uncertainty")
            out["focal_uncertainty"] = (camera.new_zeros(camera.f.shape[0]) if zero_or_fu is None else zero_or_fu) * inv_sy
        else:
            camera = camera.undo_scale_crop(img_data)
            w, h = (int(v) for v in camera.size[0].round().tolist())
            zero = camera.new_zeros(camera.f.shape[0])
            out["focal_uncertainty"] = out.get("focal_uncertainty", zero) * (1.0 / img_data["scales"])[1]
        if tensors:        # bilinear, align_corners=False (extractor.py:60-63): one HIP launch for all of them
            for k, t in zip(tensors, upsample_fields_multi([out[k] for k in tensors], (h, w))):
                out[k] = t
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
