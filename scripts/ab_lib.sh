#!/bin/bash
# Same-box, interleaved A/B of two BUILT libraries (box-to-box variance is +-4 %: never compare across gpurun calls).
# usage: ab_lib.sh <libA.so> <libB.so> [models] [sizes] [reps]      (paths relative to the repo root)
A="$1"; B="$2"; MODELS=${3:-pinhole,simple_radial}; SIZES=${4:-1024}; REPS=${5:-2}
cd ${GRAFT_REPO_ROOT:-.}
for rep in $(seq 1 $REPS); do
  for V in A B; do
    L="$A"; [ $V = B ] && L="$B"
    echo "== $V ($L) rep $rep"
    GCLM_LIB_PATH=$PWD/$L python scripts/probes/sweep_probe.py $MODELS $SIZES
  done
done
