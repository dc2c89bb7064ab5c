"""GPU box: where the microseconds of a one-launch-per-step solve of ONE image go -- device timestamps (100 MHz wall clock)
of workgroup 0's stages in every launch, from a -DGCLM_TRACE=1 build (scripts/build_variants.sh trace="-DGCLM_TRACE=1").
    GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/trace.so python scripts/probes/trace_probe.py [model]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer, _lib
from geocalib_amd.synth import synth_fields
model = sys.argv[1] if len(sys.argv) > 1 else "pinhole"
dev = torch.device("cuda:0")
lib = C.CDLL(_lib.LIB_PATH)
d, _, _ = synth_fields(model, 1, 480, 640, dev, seed=1)
opt = LMOptimizer({"camera_model": model}).eval()
for _ in range(50): out = opt(d)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 16))()
names = ["entry", "reduced", "solved", "sync1", "phaseB", "pblock", "swept"]
for rep in range(3):
    out = opt(d); torch.cuda.synchronize()
    assert lib.gclm_debug_trace(buf) == 0
    t = [[buf[s * 16 + k] for k in range(7)] for s in range(32)]
    base = t[0][0]
    print(f"solve {rep}: stop_at {out['stop_at'][0].item():.0f}   (us since the first launch's entry; stage durations)")
    for s in range(12):
        row = t[s]
        if row[0] < base: continue
        abs_us = (row[0] - base) / 100.0
        segs = " ".join(f"{names[k]}+{(row[k] - row[k-1]) / 100.0:5.2f}" if row[k] >= row[k - 1] and row[k] > base else f"{names[k]}  -- " for k in range(1, 7))
        print(f"  launch step {s:2d}: entry at {abs_us:7.2f} us | {segs} | total {(max(row[:7]) - row[0]) / 100.0:5.2f}")
