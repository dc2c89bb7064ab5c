# round 6, call 34: the round's collection on the FINAL build (t (n.uv) reuse in), then 20 more fuzz seeds on it
mkdir -p gpurun_out/r06
scripts/r06_collect.sh > gpurun_out/r06/collect_final.out 2>&1; grep -A8 "the driver.s command" gpurun_out/r06/collect_final.out | cut -c1-330
rm -f gpurun_out/r06t_fuzz_soak.txt; SOAK_TAG=r06t scripts/fuzz_soak.sh 393 412 300 > /dev/null 2>&1; grep -o "^seed [0-9]* cases 300 rc [0-9]*" gpurun_out/r06t_fuzz_soak.txt | awk '{print $2":"$6}' | paste -sd' '
grep -h "AssertionError" gpurun_out/r06t_fuzz_soak.txt | cut -c1-500
