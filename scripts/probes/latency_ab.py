"""GPU box: single-image latency (default conf: 30 steps, early stop on the device) of the library named by GCLM_LIB_PATH
(or the in-tree one), 640x480, paced depths 0 and 3, median / p10 of 300 after 300 warm-up solves.  Run once per library
in the SAME gpurun call for a same-box A/B:
    for l in a.so b.so; do GCLM_LIB_PATH=$l python scripts/probes/latency_ab.py; done"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer, _lib
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
models = sys.argv[1].split(",") if len(sys.argv) > 1 else ["pinhole", "simple_radial"]
d0, _, _ = synth_fields("pinhole", 1, 480, 640, dev, seed=1)
o = LMOptimizer({"camera_model": "pinhole"}).eval()
for _ in range(300): o(d0)
torch.cuda.synchronize()
for model in models:
    d, _, _ = synth_fields(model, 1, 480, 640, dev, seed=1)
    for depth in (0, 3):
        opt = LMOptimizer({"camera_model": model}).eval()
        opt.paced_launches = depth
        for _ in range(20): out = opt(d)
        torch.cuda.synchronize()
        n, ts = 300, []
        for _ in range(n):
            t = time.perf_counter(); out = opt(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        ts.sort()
        print(f"{os.path.basename(_lib.LIB_PATH):24s} {model:18s} depth={depth}: median {ts[n//2]*1e6:6.1f} us  p10 {ts[n//10]*1e6:6.1f}  p90 {ts[9*n//10]*1e6:6.1f}  "
              f"stop_at={out['stop_at'][0].item():.0f}  f={out['camera']._data[0,3].item():.4f}", flush=True)
