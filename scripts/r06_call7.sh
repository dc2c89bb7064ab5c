# round 6, call 7: the -m gpu suite on a build whose handles start with gclm_set_row_pairs = 1 (every radial / simple_divisional
# solve with five planes and an even height walks row pairs, also where the step would have been one launch), then on the default build
O=gpurun_out/r06; mkdir -p $O
GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/rp1.so timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu_rp1.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu_rp1.log | cut -c1-300 | tail -40
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu_call7.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest_gpu_call7.log | tail -5
