#!/bin/bash
# A/B two builds of the sweep kernel on the SAME box, interleaved.  usage: ab.sh "<flagsA>" "<flagsB>" [models] [sizes]
A="$1"; B="$2"; MODELS=${3:-pinhole,simple_radial}; SIZES=${4:-1024}
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for V in A B; do
    F="$A"; [ $V = B ] && F="$B"
    touch geocalib_amd/csrc/gclm_pass.hip
    make -C geocalib_amd/csrc PASS_FLAGS="$F" 2>&1 | grep -E "error|warning"
    echo "== $V ($F) rep $rep"; python scripts/probes/sweep_probe.py $MODELS $SIZES | tail -n +1
  done
done
