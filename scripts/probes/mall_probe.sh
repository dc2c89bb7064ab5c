#!/bin/bash
# GPU box: does a batch that fits the 256 MiB Infinity Cache sweep faster than HBM allows?  Builds the sweep with and
# without non-temporal loads (and the no-math variant of each = the memory system alone) and times it at batch sizes
# either side of the cache's capacity (6.1 MB per 640x480 image: 32 images = 197 MB).  Leaves the DEFAULT build in place.
# CLOSED (round 3, profiles/archive/r03_mall_probe.log): the -DGCLM_NT_LOADS=0 switch left gclm_pass.hip in round 6 -- the sweep's loads are
# always non-temporal; check out a round-5 tree to re-run the plain-load half of this probe.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
SIZES=${1:-8,16,24,32,40,64,256}
for F in "" "-DGCLM_NT_LOADS=0" "-DGCLM_NOMATH=1" "-DGCLM_NOMATH=1 -DGCLM_NT_LOADS=0"; do
  touch geocalib_amd/csrc/gclm_pass.hip geocalib_amd/csrc/gclm_api.hip
  make -C geocalib_amd/csrc PASS_FLAGS="$F" CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -Wall -Wno-unused-function $F" 2>&1 | grep -E "error|warning"
  echo "== flags [$F]"
  GCLM_FUSED=0 python scripts/probes/sweep_probe.py pinhole $SIZES
done
touch geocalib_amd/csrc/gclm_pass.hip geocalib_amd/csrc/gclm_api.hip
make -C geocalib_amd/csrc 2>&1 | grep -E "error|warning"
