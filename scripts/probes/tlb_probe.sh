#!/bin/bash
# GPU box: UTCL1 (per-CU address-translation cache) counters of the pinhole sweep in two physical placements of the inputs
# (fresh process / a process that first allocated and freed 40 GiB).  usage: scripts/probes/tlb_probe.sh [tag]
TAG=${1:-tlb}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in 0 40; do
  for SET in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum GRBM_UTCL2_BUSY"; do
    N=$(echo $SET | cut -c11-20)
    rocprofv3 --pmc $SET --output-format csv -d $OUT/p${G}_$N -o pmc -- python $REPO/scripts/probes/placement_probe.py $G pinhole 1024 > $OUT/p${G}_$N.log 2>&1
    grep "sweep" $OUT/p${G}_$N.log | head -1
  done
done
python - <<PY
import csv, glob, collections, json
out = {}
for g in (0, 40):
    agg = collections.defaultdict(list); dur = []
    for f in glob.glob("$OUT/p%d_*/**/*counter_collection.csv" % g, recursive=True):
        for row in csv.DictReader(open(f)):
            if "sweep_kernel<0, true, true, true, true" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"])); dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out["prealloc_%d_GiB" % g] = {k: sum(v) / len(v) for k, v in agg.items()} | {"mean_sweep_ns_under_pmc": sum(dur) / max(len(dur), 1)}
print(json.dumps(out, indent=1)); json.dump(out, open("$OUT/tlb_summary.json", "w"), indent=1)
PY
find $OUT -name "*counter_collection.csv" -size +1M -delete; find $OUT -name "*agent_info.csv" -delete
