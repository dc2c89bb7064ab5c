"""GPU box: the row-pair walk of the sweep (gclm_set_row_pairs; gclm_pass.hip: row_math_mirror) against the one-row walk on the
same fields -- radial / simple_divisional, several shapes and batch sizes, library-initialised (centred) cameras and explicit
off-centre ones.  Prints the largest distance of every output tensor; the two walks evaluate the same per-pixel values and
differ in the order of a lane's additions only (~1e-7 relative on the sums; what the solve makes of that is printed).
usage: row_pairs_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from geocalib_amd import LMOptimizer, _lib  # noqa: E402
from geocalib_amd.synth import synth_fields  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()


def raw(out):
    return {k: (v._data if hasattr(v, "_data") else v).float().cpu().numpy() for k, v in out.items() if torch.is_tensor(v) or hasattr(v, "_data")}


def dist(a, b):
    """focal relative, distortion / gravity absolute, costs relative, the rest relative to the tensor's largest entry"""
    worst = {}
    for k in a:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        if not x.size:
            continue
        if k == "camera":
            worst["focal"] = float(np.abs(x[:, 2:4] / y[:, 2:4] - 1).max())
            worst["dist"] = float(np.abs(x[:, 6:8] - y[:, 6:8]).max())
        elif k == "gravity":
            worst[k] = float(np.abs(x - y).max())
        elif "cost" in k:
            worst[k] = float(np.abs(x / np.maximum(np.abs(y), 1e-30) - 1).max())
        else:
            worst[k] = float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30))
    return worst


for model in ("radial", "simple_divisional"):
    for (B, H, W, steps) in ((16, 480, 640, 1), (16, 480, 640, 20), (3, 230, 324, 1), (3, 230, 324, 10), (300, 120, 160, 1), (300, 120, 160, 20), (2, 64, 2600, 1),
                             (5, 6, 8, 1), (5, 6, 8, 4)):
        data, gt_cam, gt_grav = synth_fields(model, B, H, W, dev, seed=7)
        res = {}
        for mode in (0, 1):
            opt = LMOptimizer({"camera_model": model, "num_steps": steps, "early_stop": False}).eval()
            opt.overlap_streams = 1
            h = opt._handle(dev)
            assert lib.gclm_set_row_pairs(h.ptr, mode) == 0
            res[mode] = raw(opt(data))
            # single-sweep system at the ground truth through an explicit, OFF-CENTRE camera (the unshared pair walk)
        d = dist(res[1], res[0])
        top = sorted(d.items(), key=lambda kv: -kv[1])[:4]
        f_err = float(np.median(np.abs(res[1]["camera"][:, 3] / gt_cam[:, 3].cpu().numpy() - 1)))
        print(f"{model:18s} B={B:4d} {W}x{H} steps {steps:2d}: pairs vs one-row, worst " + ", ".join(f"{k} {v:.2e}" for k, v in top) + f" | median focal err vs gt {f_err:.1e}")
    # explicit cameras: optimize() from the ground truth with the principal point moved off the centre
    B, H, W = 8, 240, 320
    data, gt_cam, gt_grav = synth_fields(model, B, H, W, dev, seed=9)
    from geocalib_amd.camera import camera_models  # noqa: E402
    from geocalib_amd.gravity import Gravity  # noqa: E402
    cam0 = gt_cam.clone()
    cam0[:, 4] += 3.0
    cam0[:, 5] -= 2.0
    res = {}
    for mode in (0, 1):
        opt = LMOptimizer({"camera_model": model, "num_steps": 5, "early_stop": False}).eval()
        opt.overlap_streams = 1
        h = opt._handle(dev)
        assert lib.gclm_set_row_pairs(h.ptr, mode) == 0
        opt.setup_optimization_and_priors(data)
        c, g, info = opt.optimize(data, camera_models[model](cam0.clone()), Gravity(gt_grav.clone()))
        res[mode] = raw({"camera": c, "gravity": g, **{k: v for k, v in info.items() if torch.is_tensor(v)}})
    d = dist(res[1], res[0])
    print(f"{model:18s} explicit off-centre cameras, B={B} {W}x{H}: worst " + ", ".join(f"{k} {v:.2e}" for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:4]))
