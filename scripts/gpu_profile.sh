#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel trace + HBM PMC passes.
# Usage: scripts/gpu_profile.sh <tag> [camera_model]      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-r01}
MODEL=${2:-pinhole}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench" 
python $REPO/bench.py --steps 10 --warmup 2 --camera-model $MODEL > $OUT/bench_$MODEL.json 2> $OUT/bench_$MODEL.err
tail -c 3000 $OUT/bench_$MODEL.json
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$MODEL -o kt -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 --camera-model $MODEL > $OUT/kt_$MODEL.log 2>&1
find $OUT/kt_$MODEL -name "*kernel_stats.csv" | head -1 | xargs -r head -12
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$MODEL -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-timing --camera-model $MODEL > $OUT/pmc_fetch_$MODEL.log 2>&1
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$MODEL -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-timing --camera-model $MODEL > $OUT/pmc_write_$MODEL.log 2>&1
python - <<PY
import csv, glob, json, collections
out = {}
for name in ("fetch", "write"):
    files = glob.glob("$OUT/pmc_%s_$MODEL/**/*counter_collection.csv" % name, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[(row.get("Kernel_Name", "")[:60], row.get("Counter_Name"))].append(float(row.get("Counter_Value", 0)))
    for (k, c), v in sorted(agg.items()):
        if "sweep" in k:
            out["%s:%s" % (c, k)] = {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/pmc_summary_$MODEL.json", "w"), indent=1)
PY
# keep only the small summaries (gpurun_out is size-limited)
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
ls -R $OUT | head -40
