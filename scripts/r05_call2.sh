#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
L=geocalib_amd/lib/libgeocalib_hip.so
timeout 600 python bench.py --camera-model simple_radial --steps 10 --warmup 3 --cpu-sample 0 --placement-tries 1 --no-overlap > $O/bench_sr.json 2> $O/bench_sr.err; echo "rc $?" >> $O/bench_sr.err
timeout 900 python scripts/variant_probe.py --models simple_radial --reps 3 auto=$L@0@-1 off=$L@0@0 r04=geocalib_amd/lib/variants/r04.so > $O/variant_slat2.log 2>&1
python - <<'PY' > $O/ws_probe.log 2>&1
import torch, time
from geocalib_amd import LMOptimizer, _lib
from geocalib_amd.synth import synth_fields
lib=_lib.load()
dev=torch.device("cuda:0")
data,_,_=synth_fields("simple_radial",1024,480,640,dev,seed=2024)
for mode in (-1,0,-1,0):
    opt=LMOptimizer({"camera_model":"simple_radial","num_steps":20,"early_stop":False}).eval(); opt.overlap_streams=1
    h=opt._handle(dev); lib.gclm_set_slat_plane(h.ptr,mode)
    for _ in range(2): opt(data)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): opt(data)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(mode, lib.gclm_workspace_bytes(h.ptr), f"{dt*1e3:.3f} ms", f"{1024/dt:.0f} img/s")
PY
cat $O/bench_sr.json; tail -2 $O/bench_sr.err; cat $O/variant_slat2.log $O/ws_probe.log
