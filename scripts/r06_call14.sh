# round 6, call 14: radial's row pairs as the built-in choice too (3 waves per SIMD) -- the -m gpu suite, the same-allocation A/B of the shipped build
O=gpurun_out/r06; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu_call14.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu_call14.log | tail -5
grep -E "^E " $O/pytest_gpu_call14.log | head -10 | cut -c1-500
timeout 900 python scripts/variant_probe.py --models radial,simple_divisional --reps 3 pairs=geocalib_amd/lib/libgeocalib_hip.so onerow=geocalib_amd/lib/variants/norp.so 2>&1 | grep -v amdgpu > $O/variant_row_pairs_shipped.log; cat $O/variant_row_pairs_shipped.log
