cd ${GRAFT_REPO_ROOT:-.}
scripts/gpu_profile.sh r03 pinhole pinhole_B1024 1 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026" | tail -30
scripts/gpu_profile.sh r03 simple_radial simple_radial_B1024 1 2>&1 | grep -v "amdgpu.ids\|^E2026\|^W2026" | tail -12
