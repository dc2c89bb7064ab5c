#!/bin/bash
# round-3 GPU call 1: parity with the tightened gates (no -x: collect everything), bit A/B of the guard-free divisional body,
# timing A/B, power probes, first soak seeds
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
export GCLM_PARITY_LOG=$PWD/gpurun_out/r03/parity_measured.json
timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | tail -120 > gpurun_out/r03/pytest_gpu.log
tail -40 gpurun_out/r03/pytest_gpu.log
unset GCLM_PARITY_LOG
echo "=== sweep probe (default build)"
timeout 300 python scripts/sweep_probe.py pinhole,simple_radial,radial,simple_divisional 1024 2>&1 | tee gpurun_out/r03/sweep_probe.log
echo "=== power"
for m in pinhole simple_radial radial simple_divisional; do
  timeout 200 python scripts/power_probe.py gpurun_out/r03/power_$m.json --tag $m -- --camera-model $m --steps 28 --warmup 2 --repeats 1 2>&1 | tail -2
done
echo "=== bits A/B divisional guard"
timeout 600 scripts/ab_bits.sh "-DGCLM_DIV_GUARD_ALWAYS=1" "" simple_divisional 2>&1 | tail -8 | tee gpurun_out/r03/ab_bits_div.log
echo "=== timing A/B divisional guard"
timeout 600 scripts/ab.sh "-DGCLM_DIV_GUARD_ALWAYS=1" "" simple_divisional 1024 2>&1 | tee gpurun_out/r03/ab_div.log
touch geocalib_amd/csrc/gclm_pass.hip; make -C geocalib_amd/csrc 2>&1 | grep -E "error|warning"
echo "=== soak (2 seeds)"
timeout 900 scripts/fuzz_soak.sh 11 12 300 2>&1 | tail -6
