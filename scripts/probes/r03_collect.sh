#!/bin/bash
# round-3 GPU call 6 (final binary): full -m gpu suite, soak, power probes (incl. the no-math build), the round's profile set
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export GCLM_PARITY_LOG=$PWD/gpurun_out/r03/parity_measured.json
rm -f $GCLM_PARITY_LOG
timeout 1200 python -m pytest tests -m gpu -q -s --timeout 300 > gpurun_out/r03/pytest_gpu_full.log 2>&1
grep -h "^fuzz seed" gpurun_out/r03/pytest_gpu_full.log | cut -c1-900
tail -6 gpurun_out/r03/pytest_gpu_full.log
tail -150 gpurun_out/r03/pytest_gpu_full.log > gpurun_out/r03/pytest_gpu.log; rm -f gpurun_out/r03/pytest_gpu_full.log
unset GCLM_PARITY_LOG
echo "=== soak"
rm -f gpurun_out/r03_fuzz_soak.txt
timeout 1800 scripts/fuzz_soak.sh 11 22 300 2>&1 | grep "^seed" | cut -c1-200
echo "=== power (steady state, 150 steps)"
for m in pinhole simple_radial radial simple_divisional; do
  timeout 300 python scripts/power_probe.py gpurun_out/r03/power_$m.json --tag $m -- --camera-model $m --steps 150 --warmup 2 --repeats 1 2>&1 | tail -1 | cut -c1-700
done
touch geocalib_amd/csrc/gclm_pass.hip
make -C geocalib_amd/csrc PASS_FLAGS="-DGCLM_NOMATH=1" 2>&1 | grep -E "error|warning"
GCLM_BENCH_NO_CHECK=1 timeout 300 python scripts/power_probe.py gpurun_out/r03/power_nomath.json --tag nomath -- --camera-model pinhole --steps 150 --warmup 2 --repeats 1 2>&1 | tail -1 | cut -c1-700
touch geocalib_amd/csrc/gclm_pass.hip; make -C geocalib_amd/csrc 2>&1 | grep -E "error|warning"
echo "=== profiles"
timeout 1500 scripts/gpu_profile_all.sh r03 2>&1 | grep -v amdgpu.ids | grep -v "^E2026\|^W2026" | tail -60
