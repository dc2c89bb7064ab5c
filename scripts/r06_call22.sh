# round 6, call 22: GCLM_LAT_PAIRS = 1 (the latitude sums of a row pair taken together) -- agreement with the one-row walk, the
# -m gpu suite on a build that forces row pairs (rp1) and on the default build, 20 new fuzz seeds on the forced build
O=gpurun_out/r06; mkdir -p $O
timeout 600 python scripts/probes/row_pairs_probe.py 2>&1 | grep -v amdgpu.ids > $O/row_pairs_probe_latp.log; cut -c1-250 $O/row_pairs_probe_latp.log | tail -30
export GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/rp1.so
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu_rp1_latp.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu_rp1_latp.log | cut -c1-300 | tail -20
rm -f gpurun_out/r06j_fuzz_soak.txt; SOAK_TAG=r06j scripts/fuzz_soak.sh 293 312 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06j_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
unset GCLM_LIB_PATH
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu_call22.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest_gpu_call22.log | tail -5
