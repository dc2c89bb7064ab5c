# round 6, call 25: fuzz seeds 307 / 309 with the early-stop drift diagnosis in the test (shipped build and the forced row-pair build),
# then 10 new seeds on the forced row-pair build (GCLM_LAT_PAIRS = 1)
rm -f gpurun_out/r06m_fuzz_soak.txt; SOAK_TAG=r06m scripts/fuzz_soak.sh 307 307 300 > /dev/null 2>&1; SOAK_TAG=r06m scripts/fuzz_soak.sh 309 309 300 > /dev/null 2>&1
cut -c1-1200 gpurun_out/r06m_fuzz_soak.txt
export GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/rp1.so
rm -f gpurun_out/r06n_fuzz_soak.txt; SOAK_TAG=r06n scripts/fuzz_soak.sh 307 307 300 > /dev/null 2>&1; SOAK_TAG=r06n scripts/fuzz_soak.sh 309 309 300 > /dev/null 2>&1
SOAK_TAG=r06n scripts/fuzz_soak.sh 313 322 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06n_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
grep -h "AssertionError" gpurun_out/r06n_fuzz_soak.txt | cut -c1-600
