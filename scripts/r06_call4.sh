# round 6, call 4: fuzz 115/45 (radial, beyond its gate in r06_fuzz_soak_93_142_before_tie_fix.txt) step by step; the -m gpu
# suite on the tie-fix build; seeds 113-142 again on that build
O=gpurun_out/r06; mkdir -p $O
timeout 600 python scripts/fuzz_trace.py 115 45 > $O/fuzz_trace_115_45.log 2>&1; cut -c1-400 $O/fuzz_trace_115_45.log | tail -30
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu_call4.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_call4.log | tail -3
rm -f gpurun_out/r06d_fuzz_soak.txt; SOAK_TAG=r06d scripts/fuzz_soak.sh 113 142 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06d_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
