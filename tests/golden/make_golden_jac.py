"""Goldens for the per-pixel Jacobian fields (run in the build container only):

    python tests/golden/make_golden_jac.py      ->  tests/golden/golden_jac.npz

The reference's J_perspective_field (perspective_fields.py:323-365 -> J_up_field :84-182, J_latitude_field
:214-275) for all four camera models, both parametrisations of the LM loop (spherical manifold + log focal)
and of the uncertainty pass (roll/pitch + plain focal), two cameras each on a 12 x 16 image (float64, so that
the fixture is the reference's formulas and not their float32 rounding); plus calculate_residuals / calculate_costs
(lm_optimizer.py:248-315, float32) on noisy fields at a perturbed estimate."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ref = ref_import.load()
H, W = 12, 16
CAMS = {   # w, h, fx, fy, cx, cy, k1, k2
    "pinhole": [[W, H, 14.0, 14.0, 8.0, 6.0, 0.0, 0.0], [W, H, 9.5, 11.0, 7.2, 6.4, 0.0, 0.0]],
    "simple_radial": [[W, H, 14.0, 14.0, 8.0, 6.0, -0.12, -0.12], [W, H, 9.5, 11.0, 7.2, 6.4, 0.08, 0.08]],
    "radial": [[W, H, 14.0, 14.0, 8.0, 6.0, -0.12, 0.02], [W, H, 9.5, 11.0, 7.2, 6.4, 0.08, -0.015]],
    "simple_divisional": [[W, H, 14.0, 14.0, 8.0, 6.0, -0.12, -0.12], [W, H, 9.5, 11.0, 7.2, 6.4, 0.08, 0.08]],
}
RP = [[0.2, -0.3], [-0.5, 0.45]]


def main():
    out = {}
    for model, cams in CAMS.items():
        cam = ref.camera.camera_models[model](torch.tensor(cams, dtype=torch.float64))
        grav = ref.gravity.Gravity.from_rp(torch.tensor([r for r, _ in RP], dtype=torch.float64),
                                           torch.tensor([p for _, p in RP], dtype=torch.float64))
        out[f"{model}/camera"] = cam._data.numpy().copy()
        out[f"{model}/gravity"] = grav._data.numpy().copy()
        for tag, (sph, logf) in {"loop": (True, True), "rpf": (False, False)}.items():
            J_up, J_lat = ref.perspective_fields.J_perspective_field(cam, grav, spherical=sph, log_focal=logf)
            out[f"{model}/{tag}/J_up"] = J_up.numpy().copy()
            out[f"{model}/{tag}/J_lat"] = J_lat.numpy().copy()
            print(model, tag, tuple(J_up.shape), tuple(J_lat.shape), float(J_up.abs().max()), float(J_lat.abs().max()))
    # calculate_residuals / calculate_costs (lm_optimizer.py:248-315) on noisy fields of the same cameras
    rng = np.random.default_rng(3)
    for model, cams in CAMS.items():
        cam = ref.camera.camera_models[model](torch.tensor(cams, dtype=torch.float32))
        grav = ref.gravity.Gravity.from_rp(torch.tensor([r for r, _ in RP]), torch.tensor([p for _, p in RP]))
        up, lat = ref.perspective_fields.get_perspective_field(cam, grav)
        up = torch.nn.functional.normalize(up + torch.from_numpy(rng.normal(0, 0.03, up.shape).astype(np.float32)), dim=1)
        lat = (lat + torch.from_numpy(rng.normal(0, 0.03, lat.shape).astype(np.float32))).clamp(-1.5, 1.5)
        data = {"up_field": up, "latitude_field": lat,
                "up_confidence": torch.from_numpy(rng.uniform(0, 1, (2, H, W)).astype(np.float32)),
                "latitude_confidence": torch.from_numpy(rng.uniform(0, 1, (2, H, W)).astype(np.float32))}
        # evaluate at a PERTURBED estimate so that residuals straddle the Huber threshold
        cam2 = ref.camera.camera_models[model](cam._data * torch.tensor([1, 1, 1.04, 1.04, 1, 1, 0.9, 0.9]))
        grav2 = ref.gravity.Gravity.from_rp(grav.roll + 0.02, grav.pitch - 0.015)
        opt = ref.lm_optimizer.LMOptimizer({"camera_model": model})
        res = opt.calculate_residuals(cam2, grav2, data)
        costs, weights = opt.calculate_costs(res, data)
        for k, v in data.items():
            out[f"{model}/res/{k}"] = v.numpy().copy()
        out[f"{model}/res/camera"], out[f"{model}/res/gravity"] = cam2._data.numpy().copy(), grav2._data.numpy().copy()
        for d in (res, costs, weights):
            for k, v in d.items():
                out[f"{model}/res/{k}"] = v.numpy().copy()
        print(model, "residuals", {k: tuple(v.shape) for k, v in res.items()}, "frac beyond Huber threshold",
              float((weights["up_weights"] < data["up_confidence"].reshape(2, -1) * 0.999).float().mean()))
    np.savez_compressed(os.path.join(HERE, "golden_jac.npz"), **out)


if __name__ == "__main__":
    main()
