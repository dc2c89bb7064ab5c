#!/bin/bash
# One gpurun call that collects the round's evidence (bench lines, rocprofv3 kernel stats, HBM and SQ PMC passes, latency):
#   gpurun --timeout 1500 -- 'scripts/gpu_profile_all.sh r03'        then copy the summaries: scripts/keep_profiles.py r03
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
S=scripts/gpu_profile.sh
$S $TAG pinhole pinhole_B1024 1
$S $TAG simple_radial simple_radial_B1024 1
$S $TAG radial radial_B1024 0 --cpu-sample 0
$S $TAG simple_divisional simple_divisional_B1024 1 --cpu-sample 0      # (PMC: the row-pair walk reads every byte once, too)
$S $TAG pinhole shared16_pinhole 0 --shared-group 16 --cpu-sample 0
$S $TAG simple_radial shared16_simple_radial 0 --shared-group 16 --cpu-sample 0
$S $TAG pinhole pinhole_B8192 0 --batch 8192 --steps 3 --cpu-sample 0
scripts/pmc_sq.sh $TAG pinhole pinhole_B1024
scripts/pmc_sq.sh $TAG simple_radial simple_radial_B1024
scripts/pmc_sq.sh $TAG radial radial_B1024
scripts/pmc_sq.sh $TAG simple_divisional simple_divisional_B1024
scripts/rccl_1rank.sh $TAG
python scripts/latency_probe.py --json gpurun_out/$TAG/latency.json
python scripts/paced_probe.py --json gpurun_out/$TAG/paced.json
du -sh gpurun_out/$TAG
