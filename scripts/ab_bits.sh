#!/bin/bash
# GPU box: are two builds of the sweep bit-identical in their RESULTS?  usage: ab_bits.sh "<flagsA>" "<flagsB>" [models]
# Builds the library twice (PASS_FLAGS of geocalib_amd/csrc/Makefile), solves the same device-generated batches with each
# and compares every output tensor bit for bit (scripts/dump_results.py).  Leaves the DEFAULT build in place.
A="$1"; B="$2"; MODELS=${3:-simple_divisional}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for V in A B; do
  F="$A"; [ $V = B ] && F="$B"
  BASE="${PASS_BASE_A:-}"; [ $V = B ] && BASE="${PASS_BASE_B:-}"        # optional: another PASS_BASE (Makefile) per side
  touch geocalib_amd/csrc/gclm_pass.hip geocalib_amd/csrc/gclm_api.hip
  make -C geocalib_amd/csrc ${BASE:+PASS_BASE="$BASE"} PASS_FLAGS="$F" CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -Wall -Wno-unused-function $F" 2>&1 | grep -E "error|warning"
  python scripts/dump_results.py gpurun_out/bits_$V.npz $MODELS
done
touch geocalib_amd/csrc/gclm_pass.hip geocalib_amd/csrc/gclm_api.hip
make -C geocalib_amd/csrc 2>&1 | grep -E "error|warning"
python - <<PY
import numpy as np
a, b = np.load("gpurun_out/bits_A.npz"), np.load("gpurun_out/bits_B.npz")
bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
print(f"A = [$A]  B = [$B]: {len(a.files)} tensors compared, {len(bad)} differ", bad[:10])
for k in bad[:10]:
    d = np.abs(a[k].astype(np.float64) - b[k]); print("   ", k, "max abs diff", np.nanmax(d), "rel", np.nanmax(d / np.maximum(np.abs(a[k]), 1e-30)))
PY
