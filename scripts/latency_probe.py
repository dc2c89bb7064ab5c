"""Probe: end-to-end latency of LMOptimizer.forward for small batches (GPU box).  `--json OUT` keeps the numbers."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geocalib_amd import LMOptimizer, _lib
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
rows = []
for model in ("pinhole", "simple_radial", "radial", "simple_divisional"):
    for (B, H, W) in ((1, 320, 480), (1, 480, 640), (4, 480, 640), (16, 480, 640), (64, 480, 640)):
        d, gtc, _ = synth_fields(model, B, H, W, dev, seed=1)
        for conf, fused in (({"num_steps": 20, "early_stop": False}, 0), ({"num_steps": 20, "early_stop": False}, 1),
                            ({"num_steps": 20, "early_stop": False}, -1), ({}, 0), ({}, 1), ({}, -1)):
            opt = LMOptimizer({"camera_model": model, **conf}).eval()
            h = opt._handle(dev)
            _lib.check(_lib.load().gclm_set_fused_steps(h.ptr, fused), h.ptr, "gclm_set_fused_steps")   # 0: two launches per step, 1: one (where valid), -1: the library's choice
            for _ in range(3): out = opt(d)
            torch.cuda.synchronize()
            n = 50; ts = []
            for _ in range(n):
                t = time.perf_counter(); out = opt(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            ts.sort(); dt = ts[n // 2]
            rows.append({"camera_model": model, "batch": B, "height": H, "width": W, "fused_steps_mode": fused,
                         "conf": "num_steps=20, early_stop=False" if conf else "default (30 steps, early stop on the device)",
                         "median_us_per_solve": round(dt * 1e6, 1), "p10_us": round(ts[n // 10] * 1e6, 1),
                         "p90_us": round(ts[(9 * n) // 10] * 1e6, 1), "stop_at": out["stop_at"][0].item()})
            print(f"{model:14s} B={B:3d} {W}x{H} conf={'bench20' if conf else 'default(early stop)'} fused={fused:2d}: {dt*1e6:8.1f} us/solve  stop_at={out['stop_at'][0].item():.0f}", flush=True)
if "--json" in sys.argv:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[sys.argv.index("--json") + 1])), exist_ok=True)
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
        json.dump({"what": "host wall time of LMOptimizer.forward + synchronize, median of 50", "rows": rows}, fh, indent=1)
