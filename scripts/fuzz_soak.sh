#!/bin/bash
# GPU box: the fuzz of tests/test_gpu_parity.py over many seeds, one summary line per seed.
# usage: fuzz_soak.sh [first_seed] [last_seed] [cases]   ->  gpurun_out/${SOAK_TAG:-r05}_fuzz_soak.txt (copy to profiles/)
A=${1:-11}; B=${2:-22}; N=${3:-300}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
OUT=gpurun_out/${SOAK_TAG:-r05}_fuzz_soak.txt
for s in $(seq $A $B); do
  GCLM_PARITY_LOG=$PWD/gpurun_out/fuzz_measured_$s.json GCLM_FUZZ_SEED=$s GCLM_FUZZ_CASES=$N \
    timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -s -k randomised_configurations > gpurun_out/fuzz_$s.log 2>&1
  rc=$?
  line=$(grep -h "^fuzz seed" gpurun_out/fuzz_$s.log | tail -1)
  frac=$(grep -h "^fuzz seed .* fraction of the draws" gpurun_out/fuzz_$s.log | tail -1 | sed "s/^fuzz seed [0-9]*: //")
  echo "seed $s cases $N rc $rc | ${line:-NO SUMMARY (see fuzz_$s.log)} | ${frac}" >> $OUT
  [ $rc -ne 0 ] && grep -h "AssertionError\|assert " gpurun_out/fuzz_$s.log | head -5 >> $OUT
done
cat $OUT
