"""GPU box: the placement effect as a function of the RELATIVE position of the four input tensors inside ONE allocation
(one slab = one hipMalloc: physically as contiguous as the driver makes it).  pads: bytes inserted between consecutive
tensors (up_field | latitude_field | up_confidence | latitude_confidence)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer, _lib
LMOptimizer.overlap_streams = 1      # these probes time single launches (the library default would split a large batch over two streams)
from geocalib_amd.synth import synth_fields
lib, dev = _lib.load(), torch.device("cuda:0")
B, H, W = 1024, 480, 640
model = "pinhole"


def sweep_us(opt, d, n=3):
    opt(d); torch.cuda.synchronize()
    h = opt._handle(dev); lib.gclm_set_timing(h.ptr, 1)
    for _ in range(n): opt(d)
    torch.cuda.synchronize()
    k, ms = C.c_int(0), C.c_float(0); lib.gclm_last_pass_timing(h.ptr, C.byref(k), C.byref(ms)); lib.gclm_set_timing(h.ptr, 0)
    return ms.value / k.value * 1e3


opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
ref, _, _ = synth_fields(model, B, H, W, dev, seed=1)
print(f"separate allocations: {sweep_us(opt, ref):7.1f} us")
order = ("up_field", "latitude_field", "up_confidence", "latitude_confidence")
MB = 1 << 20
for rep in range(2):
    for pad in (0, 4096, 64 << 10, 1 * MB, 2 * MB, 3 * MB, 8 * MB, 34 * MB, 130 * MB, 514 * MB):
        total = sum(ref[k].numel() for k in order) + 4 * (pad // 4) + 1024
        slab = torch.empty(total, dtype=torch.float32, device=dev)
        d, off = {}, 0
        for k in order:
            n = ref[k].numel()
            d[k] = slab[off:off + n].view(ref[k].shape)
            d[k].copy_(ref[k])
            off += n + pad // 4
        print(f"slab rep {rep} pad {pad / MB:9.3f} MiB: {sweep_us(opt, d):7.1f} us   base {hex(slab.data_ptr())}", flush=True)
        del d, slab
        torch.cuda.empty_cache()
