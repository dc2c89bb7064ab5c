# round 6, call 40: 90 more new fuzz seeds on the FINAL build (60 on the shipped library, 30 on the build that forces row pairs)
rm -f gpurun_out/r06w_fuzz_soak.txt; SOAK_TAG=r06w scripts/fuzz_soak.sh 473 532 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06w_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
export GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/rp1.so
rm -f gpurun_out/r06x_fuzz_soak.txt; SOAK_TAG=r06x scripts/fuzz_soak.sh 533 562 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06x_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
grep -h "AssertionError" gpurun_out/r06w_fuzz_soak.txt gpurun_out/r06x_fuzz_soak.txt | cut -c1-600
