#!/bin/bash
# round-3 GPU call 3: full -m gpu suite, latency, soak, A/B (divisional occupancy, log-focal final sweep), then the round's
# profile set (scripts/gpu_profile_all.sh r03)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
export GCLM_PARITY_LOG=$PWD/gpurun_out/r03/parity_measured.json
rm -f $GCLM_PARITY_LOG
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1200 python -m pytest tests -m gpu -q -s --timeout 300 2>&1 | tail -150 > gpurun_out/r03/pytest_gpu.log
tail -12 gpurun_out/r03/pytest_gpu.log
unset GCLM_PARITY_LOG
echo "=== soak"
rm -f gpurun_out/r03_fuzz_soak.txt
timeout 1800 scripts/fuzz_soak.sh 11 22 300 2>&1 | grep "^seed" | cut -c1-700
echo "=== A/B divisional occupancy"
timeout 600 scripts/ab.sh "-DGCLM_DIV_WAVES=2" "" simple_divisional 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/ab_div_waves.log
echo "=== A/B log-focal final sweep (ISO_FINAL)"
for rep in 1 2; do
  for F in "-DGCLM_ISO_FINAL=0" ""; do
    touch geocalib_amd/csrc/gclm_api.hip
    make -C geocalib_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -Wall -Wno-unused-function $F" 2>&1 | grep -E "error|warning"
    echo "== [$F] rep $rep"; python scripts/sweep_probe.py pinhole,simple_radial,radial,simple_divisional 1024 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r03/ab_iso_final.log
touch geocalib_amd/csrc/gclm_pass.hip geocalib_amd/csrc/gclm_api.hip; make -C geocalib_amd/csrc 2>&1 | grep -E "error|warning"
echo "=== profiles"
timeout 1500 scripts/gpu_profile_all.sh r03 2>&1 | grep -v amdgpu.ids | tail -120
