"""Probe (GPU box): where the host-side microseconds of a single-image LMOptimizer.forward go (cProfile, 2000 calls)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
d, _, _ = synth_fields("pinhole", 1, 480, 640, dev, seed=1)
opt = LMOptimizer({"camera_model": "pinhole", "num_steps": 0}).eval()      # num_steps = 0: three launches, the host side dominates
for _ in range(20): opt(d)
torch.cuda.synchronize()
n = 2000
t = time.perf_counter()
for _ in range(n): out = opt(d)
host = (time.perf_counter() - t) / n
torch.cuda.synchronize()
print(f"host time per forward (num_steps=0, no sync): {host*1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): out = opt(d)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative")
st.print_stats(28)
