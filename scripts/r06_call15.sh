# round 6, call 15: the shipped build's row-pair walkers (radial at 3 waves per SIMD) under the seeded fuzz on NEW seeds, forced
# on wherever the sweep can (a build whose handles start with gclm_set_row_pairs = 1), and the -m gpu suite on that build
export GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/rp1.so
rm -f gpurun_out/r06g_fuzz_soak.txt; SOAK_TAG=r06g scripts/fuzz_soak.sh 173 192 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06g_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r06/pytest_gpu_rp1b.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r06/pytest_gpu_rp1b.log | cut -c1-200 | tail -12
