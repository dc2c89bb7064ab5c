"""Probe (GPU box): single-image latency with and without paced launches (gclm_set_paced_launches), median of 200."""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geocalib_amd import LMOptimizer
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
rows = []
_d, _, _ = synth_fields("pinhole", 1, 480, 640, dev, seed=1)          # the process's first few hundred launches are slower: not measured
_o = LMOptimizer({"camera_model": "pinhole"}).eval()
for _ in range(300): _o(_d)
torch.cuda.synchronize()
for model in ("pinhole", "simple_radial", "radial", "simple_divisional"):
    for (H, W) in ((320, 480), (480, 640)):
        d, _, _ = synth_fields(model, 1, H, W, dev, seed=1)
        for depth in (0, 1, 2, 3, 4, 6):
            opt = LMOptimizer({"camera_model": model}).eval()
            opt.paced_launches = depth
            for _ in range(5): out = opt(d)
            torch.cuda.synchronize()
            n = 200; ts = []; ret = []
            for _ in range(n):
                t = time.perf_counter(); out = opt(d); r = time.perf_counter(); torch.cuda.synchronize(); e = time.perf_counter()
                ts.append(e - t); ret.append(r - t)
            ts.sort(); ret.sort()
            rows.append({"camera_model": model, "height": H, "width": W, "paced_depth": depth, "median_us_per_solve": round(ts[n // 2] * 1e6, 1),
                         "p10_us": round(ts[n // 10] * 1e6, 1), "p90_us": round(ts[(9 * n) // 10] * 1e6, 1),
                         "median_us_until_the_call_returns": round(ret[n // 2] * 1e6, 1), "stop_at": out["stop_at"][0].item()})
            print(f"{model:18s} {W}x{H} depth={depth}: {ts[n//2]*1e6:7.1f} us/solve (p10 {ts[n//10]*1e6:6.1f}, p90 {ts[(9*n)//10]*1e6:6.1f}), call returns after {ret[n//2]*1e6:6.1f} us, stop_at={out['stop_at'][0].item():.0f}", flush=True)
if "--json" in sys.argv:
    path = sys.argv[sys.argv.index("--json") + 1]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as fh:
        json.dump({"what": "host wall time of LMOptimizer.forward + synchronize for ONE image, default conf (30 steps, early stop), median of 200; "
                           "paced_depth 0 = every launch issued at once", "rows": rows}, fh, indent=1)
