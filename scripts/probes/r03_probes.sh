#!/bin/bash
# round-3 GPU call (final tree): the full -m gpu suite once more, then the side probes (batch scaling, extremes, the kernels
# either side of the path) -> gpurun_out/r03/probes.log, fields_kernels.json
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -6
{
echo "=== batch scaling (sweep_probe)"; timeout 600 python scripts/probes/sweep_probe.py pinhole,simple_radial 1,4,16,32,64,128,256,512,2048
echo "=== extremes (stress_probe)"; timeout 600 python scripts/probes/stress_probe.py
echo "=== fields kernels"; timeout 300 python scripts/probes/fields_probe.py --json gpurun_out/r03/fields_kernels.json
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/probes.log
