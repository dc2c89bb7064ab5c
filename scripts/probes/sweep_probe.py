"""Probe: mean sweep time / solve throughput at a few batch sizes (GPU box)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer, _lib
LMOptimizer.overlap_streams = 1      # these probes time single launches (the library default would split a large batch over two streams)
from geocalib_amd.synth import synth_fields
lib = _lib.load(); dev = torch.device("cuda:0")
H, W = 480, 640
models = sys.argv[1].split(",") if len(sys.argv) > 1 else ["pinhole", "simple_radial"]
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 256, 1024]
for model in models:
  for B in sizes:
    d, gtc, _ = synth_fields(model, B, H, W, dev, seed=1)
    opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
    out = opt(d); torch.cuda.synchronize()
    h = opt._handle(dev); lib.gclm_set_timing(h.ptr, 1)
    t = time.perf_counter(); n = 5
    for _ in range(n): out = opt(d)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    k, ms = C.c_int(0), C.c_float(0); lib.gclm_last_pass_timing(h.ptr, C.byref(k), C.byref(ms))
    avg = ms.value / k.value
    ferr = (out["camera"]._data[:, 3] / gtc[:, 3] - 1).abs().nanmedian().item()
    print(f"{model:14s} B={B:5d}: sweep {avg*1e3:8.1f} us = {B*H*W*20/avg/1e9:6.2f} TB/s | solve {dt*1e3:7.2f} ms = {B/dt:8.0f} img/s | f err {ferr:.1e}", flush=True)
    del d
