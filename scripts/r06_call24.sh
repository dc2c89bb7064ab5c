# round 6, call 24: fuzz seeds 307 / 309 on the SHIPPED default build, then radial draws 307/137 (image 2) and 309/232 (image 1) step by step
O=gpurun_out/r06; mkdir -p $O
{
rm -f gpurun_out/r06l_fuzz_soak.txt; SOAK_TAG=r06l scripts/fuzz_soak.sh 307 307 300 > /dev/null 2>&1; SOAK_TAG=r06l scripts/fuzz_soak.sh 309 309 300 > /dev/null 2>&1
echo "== shipped default build"; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06l_fuzz_soak.txt; grep "AssertionError" gpurun_out/r06l_fuzz_soak.txt | cut -c1-400
echo "== fuzz_trace 307 137 image 2"; timeout 600 python scripts/fuzz_trace.py 307 137 4 2 2>&1 | grep -v amdgpu | cut -c1-400
echo "== fuzz_trace 309 232 image 1"; timeout 600 python scripts/fuzz_trace.py 309 232 4 1 2>&1 | grep -v amdgpu | cut -c1-400
} > $O/fuzz_307_309_trace.log 2>&1
cat $O/fuzz_307_309_trace.log
