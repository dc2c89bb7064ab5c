# round 6, call 30: the round's collection on the FINAL build (pinhole at 3 waves per SIMD, joint latitude sums), then 20 new fuzz seeds on
# the shipped build and 20 on the build that forces row pairs
mkdir -p gpurun_out/r06; scripts/r06_collect.sh > gpurun_out/r06/collect_final.out 2>&1; grep -A12 "the driver's command" gpurun_out/r06/collect_final.out | cut -c1-330
rm -f gpurun_out/r06o_fuzz_soak.txt; SOAK_TAG=r06o scripts/fuzz_soak.sh 323 342 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06o_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
export GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/rp1.so
rm -f gpurun_out/r06p_fuzz_soak.txt; SOAK_TAG=r06p scripts/fuzz_soak.sh 343 362 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06p_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
grep -h "AssertionError" gpurun_out/r06o_fuzz_soak.txt gpurun_out/r06p_fuzz_soak.txt | cut -c1-500
