"""GPU box: WHICH allocation decides the fast / slow placement of the memory-bound pinhole sweep (0.927 vs 0.960 ms)?
One process; several input sets kept alive side by side (different physical placements), each timed with the same
handle; then single tensors of a set are re-allocated one at a time, and the handle (workspace) is re-created."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer, _lib
LMOptimizer.overlap_streams = 1      # these probes time single launches (the library default would split a large batch over two streams)
from geocalib_amd.synth import synth_fields
lib, dev = _lib.load(), torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "pinhole"
B, H, W = 1024, 480, 640


def sweep_us(opt, d, n=3):
    opt(d); torch.cuda.synchronize()
    h = opt._handle(dev); lib.gclm_set_timing(h.ptr, 1)
    for _ in range(n): opt(d)
    torch.cuda.synchronize()
    k, ms = C.c_int(0), C.c_float(0); lib.gclm_last_pass_timing(h.ptr, C.byref(k), C.byref(ms)); lib.gclm_set_timing(h.ptr, 0)
    return ms.value / k.value * 1e3


opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
sets = []
for i in range(7):
    d, _, _ = synth_fields(model, B, H, W, dev, seed=1)
    sets.append(d)
    ptrs = {k[:6]: hex(v.data_ptr()) for k, v in d.items()}
    print(f"set {i}: sweep {sweep_us(opt, d):7.1f} us  {ptrs}", flush=True)
print("re-timed in the same order:", [round(sweep_us(opt, d), 1) for d in sets], flush=True)
slow = max(range(len(sets)), key=lambda i: sweep_us(opt, sets[i]))
fast = min(range(len(sets)), key=lambda i: sweep_us(opt, sets[i]))
print(f"slowest set {slow}, fastest set {fast}")
for key in list(sets[slow]):
    d = dict(sets[slow]); d[key] = sets[slow][key].clone()       # the same values in a NEW allocation
    print(f"  slow set with `{key}` re-allocated: {sweep_us(opt, d):7.1f} us")
d = {k: v.clone() for k, v in sets[slow].items()}
print(f"  slow set, all four tensors re-allocated: {sweep_us(opt, d):7.1f} us")
opt2 = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    print(f"  slow set, another handle / workspace / stream: {sweep_us(opt2, sets[slow]):7.1f} us;  fast set: {sweep_us(opt2, sets[fast]):7.1f} us")
# mixing: up field of the fast set with the other planes of the slow one (values differ per set? no: same seed -> same values)
mix = dict(sets[slow]); mix["up_field"] = sets[fast]["up_field"]
print(f"  slow set with the FAST set's up_field: {sweep_us(opt, mix):7.1f} us")
mix = dict(sets[fast]); mix["up_field"] = sets[slow]["up_field"]
print(f"  fast set with the SLOW set's up_field: {sweep_us(opt, mix):7.1f} us")


# Is it THIS access pattern, or the region?  A plain read of every tensor (torch's sum kernel), per set.
def read_gbs(t, n=5):
    t.sum(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        t.sum()
    e1.record(); torch.cuda.synchronize()
    return t.numel() * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9


for i, d in enumerate(sets):
    print(f"set {i}: plain read GB/s  " + "  ".join(f"{k[:6]} {read_gbs(v):6.0f}" for k, v in d.items()), flush=True)
