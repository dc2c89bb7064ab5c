#!/bin/bash
# GPU box: the previous round's tree (a git worktree built in the container: `git worktree add _r03tree <commit>` + make) against
# the current tree, alternating, in ONE gpurun call -- bench lines of the same command and the latency probe.
#   gpurun -- 'scripts/ab_rounds.sh _r03tree'   ->  gpurun_out/ab_rounds.log
OLD=${1:-_r03tree}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
line() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
print("  %8.0f img/s  %7.3f ms/solve-of-batch  sweep %.4f ms (%.4f of 8 TB/s)  whole job %.4f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["whole_job_frac"]))
PY
}
for rep in 1 2; do
  for m in pinhole simple_radial radial simple_divisional; do
    for tree in $OLD .; do
      ( cd $tree && python bench.py --steps 10 --warmup 2 --camera-model $m --cpu-sample 0 --placement-tries 1 $( [ "$tree" = "." ] && echo "--no-secondary --no-overlap" ) > /tmp/ab_line.json 2>/dev/null )
      echo "rep $rep $m $( [ "$tree" = "." ] && echo 'this round ' || echo 'last round ')"; line /tmp/ab_line.json
    done
  done
done
