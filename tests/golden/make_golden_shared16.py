"""Golden vectors for BASELINE configs[4] AT ITS STATED SHAPE, produced by the REFERENCE itself.

    python tests/golden/make_golden_shared16.py        (build container only: needs /root/reference)

One shared-intrinsics group = 16 frames of one camera at 640x480.  The reference has no group dimension
("512 groups x 16 frames" = 512 calls of LMOptimizer(shared_intrinsics=True) with B = 16,
lm_optimizer.py:350-383, 597-603), so each golden case is one such call: CPU, float32, eval mode, no_grad,
num_steps = 20, early_stop = False.  Inputs are regenerated from the seed by the tests
(oracle/synth.make_shared_group); only outputs and an input checksum are stored:

    golden_shared16.npz    <model>/g<group>/{camera, gravity, stop_at, initial_*, final_*, covariance, *_uncertainty,
                           input_checksum, gt_camera, gt_gravity}      model in {pinhole, simple_radial}, group in {0, 1}
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 1234
FULL = (480, 640)
FRAMES = 16
GROUPS = (0, 1)
BENCH = {"num_steps": 20, "early_stop": False}


def checksum(data):
    return np.array([np.float64(np.asarray(v, np.float64).sum()) for _, v in sorted(data.items())])


def main():
    ref = ref_import.load()
    torch.set_num_threads(os.cpu_count())
    out = {}
    for model in ("pinhole", "simple_radial"):
        for g in GROUPS:
            data, cams, gravs = synth.make_shared_group(SEED, g, model, *FULL, frames=FRAMES)
            opt = ref.lm_optimizer.LMOptimizer({"camera_model": model, "shared_intrinsics": True, **BENCH}).eval()
            t0 = time.time()
            with torch.no_grad():
                res = opt({k: torch.from_numpy(v) for k, v in data.items()})
            dt = time.time() - t0
            for k, v in res.items():
                out[f"{model}/g{g}/{k}"] = (v._data if k in ("camera", "gravity") else v).numpy().copy()
            out[f"{model}/g{g}/input_checksum"] = checksum(data)
            out[f"{model}/g{g}/gt_camera"], out[f"{model}/g{g}/gt_gravity"] = cams, gravs
            cam = out[f"{model}/g{g}/camera"]
            print(f"{model} group {g}: reference f = {cam[0, 2]:.4f} (gt {cams[0, 2]:.4f}), k1 = {cam[0, 6]:.5f} "
                  f"(gt {cams[0, 6]:.5f}), final cost {out[f'{model}/g{g}/final_cost'].mean():.6e}, {dt:.1f} s", flush=True)
    np.savez_compressed(os.path.join(HERE, "golden_shared16.npz"), **out)


if __name__ == "__main__":
    main()
