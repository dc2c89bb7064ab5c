"""Extra goldens for the training-time knobs that live in siclib (run in the build container only):

    python tests/golden/make_golden_extra.py      ->  tests/golden/golden_extra.npz

  * heuristic initialisation: siclib.models.optimization.utils.get_heuristic_estimation (utils.py:27-82)
    produces the initial (camera, gravity); the geocalib LMOptimizer.optimize (lm_optimizer.py:551) runs from it;
  * squared loss: siclib.models.optimization.losses.squared_loss (losses.py:26) substituted for huber_loss in
    the reference's calculate_costs (lm_optimizer.py:293,305).
Inputs are the committed small sets (tests/golden/inputs_<model>.npz)."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ref = ref_import.load()
for name in ("omegaconf", "h5py", "hydra"):
    if name not in sys.modules:
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = ref_import._Anything(name)
sic_utils = importlib.import_module("siclib.models.optimization.utils")
sic_losses = importlib.import_module("siclib.models.optimization.losses")
BENCH = {"num_steps": 20, "early_stop": False}


def pack(out):
    res = {}
    for k, v in out.items():
        res[k] = (v._data if hasattr(v, "_data") else v).numpy().copy() if not isinstance(v, (int, float)) else np.asarray(v)
    return res


def main():
    extra = {}
    for model in ("pinhole", "simple_radial"):
        inp = np.load(os.path.join(HERE, f"inputs_{model}.npz"))
        data = {k: torch.from_numpy(inp[k]) for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
        # ---- heuristic init (siclib camera classes share the (B,8) layout)
        sic_cam_cls = importlib.import_module("siclib.geometry.camera").camera_models[model]
        with torch.no_grad():
            cam_s, grav_s = sic_utils.get_heuristic_estimation(data, sic_cam_cls)
            opt = ref.lm_optimizer.LMOptimizer({"camera_model": model, **BENCH}).eval()
            opt.setup_optimization_and_priors(data, shared_intrinsics=False)
            cam0 = opt.camera_model(cam_s._data.clone())
            grav0 = ref.gravity.Gravity(grav_s._data.clone())
            cam, grav, infos = opt.optimize(data, cam0, grav0)
        extra[f"{model}/heuristic/init_camera"] = cam_s._data.numpy().copy()
        extra[f"{model}/heuristic/init_gravity"] = grav_s._data.numpy().copy()
        for k, v in pack({"camera": cam, "gravity": grav, **infos}).items():
            extra[f"{model}/heuristic/{k}"] = v
        print(model, "heuristic init f", cam_s._data[:, 3].numpy(), "->", cam._data[:, 3].numpy())
        # ---- squared loss
        orig = ref.lm_optimizer.huber_loss
        ref.lm_optimizer.huber_loss = sic_losses.squared_loss
        try:
            with torch.no_grad():
                out = ref.lm_optimizer.LMOptimizer({"camera_model": model, **BENCH}).eval()(dict(data))
        finally:
            ref.lm_optimizer.huber_loss = orig
        for k, v in pack(out).items():
            extra[f"{model}/squared_loss/{k}"] = v
        print(model, "squared loss f", out["camera"]._data[:, 3].numpy())
    # ---- shared intrinsics with the 5-parameter radial model (3 shared intrinsics)
    inp = np.load(os.path.join(HERE, "inputs_radial.npz"))
    data = {k: torch.from_numpy(inp[k]) for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    with torch.no_grad():
        out = ref.lm_optimizer.LMOptimizer({"camera_model": "radial", "shared_intrinsics": True, **BENCH}).eval()(data)
    for k, v in pack(out).items():
        extra[f"radial/shared/{k}"] = v
    print("radial shared f", out["camera"]._data[:, 3].numpy(), "k", out["camera"]._data[0, 6:].numpy())
    np.savez_compressed(os.path.join(HERE, "golden_extra.npz"), **extra)


if __name__ == "__main__":
    main()
