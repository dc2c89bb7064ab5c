# round 6, call 36: 30 new fuzz seeds on the build that forces row pairs (call 35's second half found no library: build() had cleaned the variants)
export GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/rp1.so
rm -f gpurun_out/r06v_fuzz_soak.txt; SOAK_TAG=r06v scripts/fuzz_soak.sh 443 472 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06v_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
grep -h "AssertionError" gpurun_out/r06v_fuzz_soak.txt | cut -c1-500
