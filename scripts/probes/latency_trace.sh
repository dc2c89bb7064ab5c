#!/bin/bash
# GPU box: kernel trace of single-image solves (where do the microseconds of a B=1 solve go?).  usage: latency_trace.sh [model]
MODEL=${1:-pinhole}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/lat_$MODEL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/lat1.py <<PY
import sys; sys.path.insert(0, "$REPO")
import torch
from geocalib_amd import LMOptimizer
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
d, _, _ = synth_fields("$MODEL", 1, 480, 640, dev, seed=1)
opt = LMOptimizer({"camera_model": "$MODEL"}).eval()
for _ in range(5):
    out = opt(d); torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python /tmp/lat1.py > $OUT/kt.log 2>&1
F=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$F")))
rows = [r for r in rows if "gclm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 5
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"]); t1 = int(last[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print(f"$MODEL: launches/solve {n}, span {(t1-t0)/1e3:.1f} us, busy {busy/1e3:.1f} us")
prev = None
for r in last[:14] + last[-6:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{r['Kernel_Name'][:48]:48s} dur {(e-s)/1e3:7.2f} us  gap {gap:6.2f} us  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}")
    prev = e
PY
