#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel trace (+ optionally the two HBM PMC passes).
# Usage: scripts/gpu_profile.sh <tag> <camera_model> [name] [pmc: 0|1] [extra bench.py args ...]
#   outputs under gpurun_out/<tag>/: bench_<name>.json, kt_<name>/.../kernel_stats.csv, pmc_summary_<name>.json
set -u
TAG=${1:-r02}
MODEL=${2:-pinhole}
NAME=${3:-$MODEL}
PMC=${4:-1}
shift 4 2>/dev/null || shift $#
EXTRA="$*"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench $NAME ($EXTRA)"
python $REPO/bench.py --steps 10 --warmup 2 --camera-model $MODEL $EXTRA > $OUT/bench_$NAME.json 2> $OUT/bench_$NAME.err
tail -c 2500 $OUT/bench_$NAME.json
echo "== kernel trace $NAME"
# the SAME command as the bench line (steps, warm-up, repeats), so that the two sweep averages are comparable: a short run
# sweeps 2-3 % faster than a sustained one (power / clocks); without the line's extra passes (`secondary`, `overlap`), whose
# launches would be averaged into the same kernel names
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$NAME -o kt -- python $REPO/bench.py --steps 10 --warmup 2 --camera-model $MODEL $EXTRA --cpu-sample 0 --no-secondary --no-overlap > $OUT/kt_$NAME.log 2>&1
# the traced process prints its own bench line: its HIP-event sweep average and rocprofv3's come from the SAME launches
grep -h "^{" $OUT/kt_$NAME.log | tail -1 > $OUT/bench_traced_$NAME.json
find $OUT/kt_$NAME -name "*kernel_stats.csv" | head -1 | xargs -r head -8
if [ "$PMC" = "1" ]; then
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$NAME -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --repeats 1 --cpu-sample 0 --no-timing --no-secondary --no-overlap --camera-model $MODEL $EXTRA > $OUT/pmc_fetch_$NAME.log 2>&1
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$NAME -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --repeats 1 --cpu-sample 0 --no-timing --no-secondary --no-overlap --camera-model $MODEL $EXTRA > $OUT/pmc_write_$NAME.log 2>&1
python - <<PY
import csv, glob, json, collections, re
out = {}
for name in ("fetch", "write"):
    files = glob.glob("$OUT/pmc_%s_$NAME/**/*counter_collection.csv" % name, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            # keyed by the kernel WITH its full template argument list: sweep_kernel<1, ..., 4, 1> (the first sweep of a
            # simple_radial solve, which also writes the scratch plane) and <..., 2> (every later sweep) are different rows
            kn = row.get("Kernel_Name", "")
            m = re.search(r"(\w+_kernel<[^>]*>)", kn)
            agg[(m.group(1) if m else kn[:120], row.get("Counter_Name"))].append(float(row.get("Counter_Value", 0)))
    for (k, c), v in sorted(agg.items()):
        if "sweep" in k:
            out["%s:%s" % (c, k)] = {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/pmc_summary_$NAME.json", "w"), indent=1)
PY
fi
# keep only the small summaries (gpurun_out is size-limited)
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*agent_info.csv" -delete
