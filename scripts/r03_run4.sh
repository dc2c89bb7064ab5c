#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for c in "19 198" "13 82" "18 97" "13 76"; do
  echo "=== fuzz_trace $c"; timeout 300 python scripts/fuzz_trace.py $c 2>&1 | grep -v "amdgpu.ids\|Warning" | cut -c1-420
done | tee gpurun_out/r03/fuzz_traces.log
echo "=== fused bit test"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "one_launch_per_step or small or full_size or trace or step_by_step" --timeout 300 2>&1 | tail -15
