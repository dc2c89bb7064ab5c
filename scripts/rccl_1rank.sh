#!/bin/bash
# GPU box: the N>1 code paths of bench.py with ONE rank through RCCL (GCLM_FORCE_COLLECTIVES=1 under torchrun), with the
# collectives routed through torch.distributed AND directly through the C ABI (gclm_comm_*), at the per-rank shapes of the
# 8-GPU runs: configs[2] (1024 images per rank, ONE all-gather) and configs[4] (--virtual-world 8: 512 groups x 2 local
# frames, ONE all-reduce of 64 KB per LM step).  usage: rccl_1rank.sh <tag>  ->  gpurun_out/<tag>/bench_1rank_rccl_*.json
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd $REPO
run() {   # name, extra args...
  local name=$1; shift
  GCLM_FORCE_COLLECTIVES=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 "$@" \
    > $OUT/bench_1rank_rccl_$name.json 2> $OUT/bench_1rank_rccl_$name.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_1rank_rccl_$name.json") if l.startswith("{")][-1])
    mg = d["multi_gpu"]
    print("$name: %.0f img/s, %.3f ms/step, collective %.4f ms/step (%d x %d B, %s), sweep %.4f ms" % (
        d["value"], d["ms_per_step"], mg["collective_ms"], mg["collectives_per_step"], mg["collective_bytes"], mg["comm"], d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$OUT/bench_1rank_rccl_$name.err").read()[-1500:])
PY
}
run independent_torch
run independent_rccl --comm rccl
run split512x2_torch --shared-group 16 --virtual-world 8
run split512x2 --shared-group 16 --virtual-world 8 --comm rccl
run split512x2_simple_radial --shared-group 16 --virtual-world 8 --comm rccl --camera-model simple_radial
run sharedbygroup_rccl --shared-group 16 --shared-by-group --comm rccl
