# round 6, call 23: fuzz seeds 307 / 309 (radial draws 137 / 232 beyond their gate on the forced row-pair build with GCLM_LAT_PAIRS = 1):
# the same seeds on the forced build WITHOUT the joint latitude sums, on the shipped default, and the two draws printed
O=gpurun_out/r06; mkdir -p $O
for v in rp1base rp1; do
  export GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/$v.so
  rm -f gpurun_out/r06k_${v}_fuzz_soak.txt; SOAK_TAG=r06k_$v scripts/fuzz_soak.sh 307 307 300 > /dev/null 2>&1; SOAK_TAG=r06k_$v scripts/fuzz_soak.sh 309 309 300 > /dev/null 2>&1
  echo "== $v"; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06k_${v}_fuzz_soak.txt; grep "AssertionError" gpurun_out/r06k_${v}_fuzz_soak.txt | cut -c1-400
  for sc in "307 137" "309 232"; do echo "-- $v draw $sc"; timeout 300 python scripts/fuzz_case.py $sc 2>&1 | grep -v amdgpu | cut -c1-300; done
done > $O/fuzz_307_309.log 2>&1
cat $O/fuzz_307_309.log
