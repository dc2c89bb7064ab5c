#!/bin/bash
# round-3 GPU call 5: the full -m gpu suite with the final gates + the 12-seed soak + latency
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export GCLM_PARITY_LOG=$PWD/gpurun_out/r03/parity_measured.json
rm -f $GCLM_PARITY_LOG
timeout 1200 python -m pytest tests -m gpu -q -s --timeout 300 > gpurun_out/r03/pytest_gpu_full.log 2>&1
grep -h "^fuzz seed" gpurun_out/r03/pytest_gpu_full.log | cut -c1-900
tail -8 gpurun_out/r03/pytest_gpu_full.log
tail -150 gpurun_out/r03/pytest_gpu_full.log > gpurun_out/r03/pytest_gpu.log; rm -f gpurun_out/r03/pytest_gpu_full.log
unset GCLM_PARITY_LOG
echo "=== soak"
rm -f gpurun_out/r03_fuzz_soak.txt
timeout 1800 scripts/fuzz_soak.sh 11 22 300 2>&1 | grep "^seed" | cut -c1-1000
echo "=== latency"
timeout 600 python scripts/latency_probe.py --json gpurun_out/r03/latency.json 2>&1 | grep "B=  1"
