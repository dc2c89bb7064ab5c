"""GPU box: one launch per LM step (gclm_set_fused_steps 1) against the two-launch sequence (0) at LARGE batches, where the
fused prologue repeats every image's update in each of its 15 workgroups.  usage: fused_probe.py [models] [sizes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from geocalib_amd import LMOptimizer, _lib  # noqa: E402
LMOptimizer.overlap_streams = 1      # these probes time single launches (the library default would split a large batch over two streams)
from geocalib_amd.synth import synth_fields  # noqa: E402

lib, dev = _lib.load(), torch.device("cuda:0")
models = sys.argv[1].split(",") if len(sys.argv) > 1 else ["pinhole", "simple_radial"]
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 256, 1024]
for model in models:
    for B in sizes:
        d, _, _ = synth_fields(model, B, 480, 640, dev, seed=1)
        res = {}
        for rep in range(2):
            for mode in (0, 1):
                opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
                h = opt._handle(dev)
                _lib.check(lib.gclm_set_fused_steps(h.ptr, mode), h.ptr)
                opt(d); torch.cuda.synchronize()
                n = 8 if B >= 256 else 30
                t = time.perf_counter()
                for _ in range(n):
                    out = opt(d)
                torch.cuda.synchronize()
                res.setdefault(mode, []).append((time.perf_counter() - t) / n * 1e3)
        print(f"{model:14s} B={B:5d}: two launches per step {min(res[0]):8.3f} ms | one launch per step {min(res[1]):8.3f} ms "
              f"({(min(res[0]) / min(res[1]) - 1) * 100:+.1f} %)", flush=True)
        del d
