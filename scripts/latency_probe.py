"""Probe: end-to-end latency of LMOptimizer.forward for small batches (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geocalib_amd import LMOptimizer
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
for model in ("pinhole", "simple_radial"):
    for (B, H, W) in ((1, 320, 480), (1, 480, 640), (4, 480, 640), (16, 480, 640)):
        d, gtc, _ = synth_fields(model, B, H, W, dev, seed=1)
        for conf in ({"num_steps": 20, "early_stop": False}, {}):
            opt = LMOptimizer({"camera_model": model, **conf}).eval()
            for _ in range(3): out = opt(d)
            torch.cuda.synchronize()
            n = 20; t = time.perf_counter()
            for _ in range(n):
                out = opt(d); torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
            print(f"{model:14s} B={B:3d} {W}x{H} conf={'bench20' if conf else 'default(early stop)'}: {dt*1e6:8.1f} us/solve  stop_at={out['stop_at'][0].item():.0f}", flush=True)
