"""GPU box: does the RELATIVE placement of the input tensors (not their content) change the sweep time?
All four tensors are views into ONE slab; tensor k starts `pad * k` bytes after the 2 MiB-aligned end of its predecessor.
(Each tensor stays contiguous NCHW, as the boundary requires.)  usage: layout_probe.py [model] [B]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer, _lib
LMOptimizer.overlap_streams = 1      # these probes time single launches (the library default would split a large batch over two streams)
from geocalib_amd.synth import synth_fields
lib = _lib.load(); dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "pinhole"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H, W = 480, 640
src, gtc, _ = synth_fields(model, B, H, W, dev, seed=1)
keys = ("up_field", "latitude_field", "up_confidence", "latitude_confidence")
MB2 = 2 << 20
for pad in (0, 256, 1024, 4096, 16384, 65536, 4096 + 256, 16384 + 1024 + 256, (1 << 20) + 4096 + 256):
    sizes = [src[k].numel() * 4 for k in keys]
    offs, o = [], 0
    for i, sz in enumerate(sizes):
        o = (o + MB2 - 1) // MB2 * MB2 + pad * i
        offs.append(o); o += sz
    slab = torch.empty(o + MB2, dtype=torch.uint8, device=dev)
    base = (slab.data_ptr() + MB2 - 1) // MB2 * MB2 - slab.data_ptr()
    d = {}
    for k, off in zip(keys, offs):
        v = slab[base + off: base + off + src[k].numel() * 4].view(torch.float32).view(src[k].shape)
        v.copy_(src[k]); d[k] = v
    opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
    out = opt(d); torch.cuda.synchronize()
    h = opt._handle(dev); lib.gclm_set_timing(h.ptr, 1)
    for _ in range(5): out = opt(d)
    torch.cuda.synchronize()
    k, ms = C.c_int(0), C.c_float(0); lib.gclm_last_pass_timing(h.ptr, C.byref(k), C.byref(ms))
    avg = ms.value / k.value
    print(f"{model} B={B} pad {pad:8d} B per tensor: sweep {avg*1e3:7.1f} us = {B*H*W*20/avg/1e9:5.2f} TB/s  (slab at {slab.data_ptr():#x})", flush=True)
    del d, slab, opt
