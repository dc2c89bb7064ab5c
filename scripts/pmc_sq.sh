#!/bin/bash
# SQ / GRBM counters of the sweep kernel (VALU utilisation, wait breakdown).  usage: pmc_sq.sh <tag> <model> [name] [extra bench args]
TAG=${1:-r02}; MODEL=${2:-pinhole}; NAME=${3:-$MODEL}
shift 3 2>/dev/null || shift $#
EXTRA="$*"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 1 --warmup 1 --repeats 1 --cpu-sample 0 --no-timing --no-secondary --no-overlap --camera-model $MODEL $EXTRA"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_$NAME -o pmc -- $B > $OUT/pmc_sq_$NAME.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/pmc_sq2_$NAME -o pmc -- $B > $OUT/pmc_sq2_$NAME.log 2>&1
python - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(list); dur = []
for f in glob.glob("$OUT/pmc_sq*_$NAME/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sweep" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
out = {k: sum(v) / len(v) for k, v in agg.items()}
out["mean_duration_ns_under_pmc"] = sum(dur) / max(len(dur), 1)
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/pmc_sq_summary_$NAME.json", "w"), indent=1)
PY
tail -3 $OUT/pmc_sq_$NAME.log
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*agent_info.csv" -delete
