"""Probe: sweep time vs batch size (does a batch that fits the 256 MiB Infinity Cache sweep faster?)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer, _lib
LMOptimizer.overlap_streams = 1      # these probes time single launches (the library default would split a large batch over two streams)
from geocalib_amd.synth import synth_fields
lib = _lib.load(); dev = torch.device("cuda:0")
H, W = 480, 640
model = sys.argv[1] if len(sys.argv) > 1 else "pinhole"
for B in (4, 8, 16, 24, 32, 40, 48, 64, 96, 128, 256, 512, 1024):
    d, gtc, _ = synth_fields(model, B, H, W, dev, seed=1)
    opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
    opt(d); torch.cuda.synchronize()
    h = opt._handle(dev); lib.gclm_set_timing(h.ptr, 1)
    t = time.perf_counter(); n = 5
    for _ in range(n): opt(d)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    k, ms = C.c_int(0), C.c_float(0); lib.gclm_last_pass_timing(h.ptr, C.byref(k), C.byref(ms))
    avg = ms.value / k.value
    print(f"{model} B={B:5d} ({B*6.144:.0f} MB): sweep {avg*1e3:8.1f} us = {B*H*W*20/avg/1e9:7.2f} TB/s-equiv | solve {dt*1e3:7.2f} ms = {B/dt:8.0f} img/s  (sweeps {avg*21/dt/1e3*100:.0f}% of wall)")
    del d
