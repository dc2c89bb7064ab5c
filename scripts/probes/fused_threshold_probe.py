"""GPU box: up to which batch does ONE launch per LM step beat two?  20 fixed steps, 640x480 and 320x240, B = 2 ... 12,
gclm_set_fused_steps 0 / 1 (median of 40)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer, _lib
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0"); lib = _lib.load()
for model in ("pinhole", "simple_radial"):
    for (H, W) in ((480, 640), (240, 320)):
        for B in (2, 4, 6, 8, 12):
            d, _, _ = synth_fields(model, B, H, W, dev, seed=1)
            res = {}
            for mode in (0, 1):
                opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
                h = opt._handle(dev); lib.gclm_set_fused_steps(h.ptr, mode)
                for _ in range(5): opt(d)
                torch.cuda.synchronize(); ts = []
                for _ in range(40):
                    t = time.perf_counter(); opt(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
                res[mode] = sorted(ts)[20] * 1e6
            print(f"{model:14s} {W}x{H} B={B:2d}: two launches {res[0]:7.1f} us, one launch {res[1]:7.1f} us  ({(res[1]/res[0]-1)*100:+5.1f} %)", flush=True)
