#!/bin/bash
# round-3 GPU call 2: full -m gpu suite, small-batch latency (one launch per step), the 12-seed soak, steady-state power
# incl. the no-math build, A/B of the log-focal final sweep
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
export GCLM_PARITY_LOG=$PWD/gpurun_out/r03/parity_measured.json
rm -f $GCLM_PARITY_LOG
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | tail -150 > gpurun_out/r03/pytest_gpu.log
tail -15 gpurun_out/r03/pytest_gpu.log
unset GCLM_PARITY_LOG
echo "=== latency"
timeout 600 python scripts/latency_probe.py --json gpurun_out/r03/latency.json 2>&1 | tail -70
echo "=== soak"
rm -f gpurun_out/r03_fuzz_soak.txt
timeout 1800 scripts/fuzz_soak.sh 11 22 300 2>&1 | tail -30
echo "=== power (steady state, 150 steps)"
for m in pinhole simple_radial radial simple_divisional; do
  timeout 300 python scripts/power_probe.py gpurun_out/r03/power_$m.json --tag $m -- --camera-model $m --steps 150 --warmup 2 --repeats 1 2>&1 | tail -1 | cut -c1-900
done
echo "=== power, no-math build (memory ceiling of the access pattern)"
touch geocalib_amd/csrc/gclm_pass.hip
make -C geocalib_amd/csrc PASS_FLAGS="-fno-slp-vectorize -DGCLM_NOMATH=1" 2>&1 | grep -E "error|warning"
GCLM_BENCH_NO_CHECK=1 timeout 300 python scripts/power_probe.py gpurun_out/r03/power_nomath.json --tag nomath -- --camera-model pinhole --steps 150 --warmup 2 --repeats 1 2>&1 | tail -1 | cut -c1-900
touch geocalib_amd/csrc/gclm_pass.hip; make -C geocalib_amd/csrc 2>&1 | grep -E "error|warning"
echo "=== A/B log-focal final sweep (ISO_FINAL)"
for rep in 1 2; do
  for F in "-DGCLM_ISO_FINAL=0" ""; do
    touch geocalib_amd/csrc/gclm_api.hip
    make -C geocalib_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -Wall -Wno-unused-function $F" 2>&1 | grep -E "error|warning"
    echo "== [$F] rep $rep"; python scripts/sweep_probe.py pinhole,simple_radial,radial 1024 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r03/ab_iso_final.log
