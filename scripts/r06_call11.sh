# round 6, call 11: the seeded fuzz on NEW seeds with every radial / simple_divisional draw that can on the row-pair walk
# (a build whose handles start with gclm_set_row_pairs = 1), then new seeds on the shipped build
export GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/rp1.so
rm -f gpurun_out/r06e_fuzz_soak.txt; SOAK_TAG=r06e scripts/fuzz_soak.sh 143 162 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06e_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
unset GCLM_LIB_PATH
rm -f gpurun_out/r06f_fuzz_soak.txt; SOAK_TAG=r06f scripts/fuzz_soak.sh 163 172 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06f_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
