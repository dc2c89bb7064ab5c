#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -f gpurun_out/r03_fuzz_soak.txt
timeout 3000 scripts/fuzz_soak.sh 11 62 300 2>&1 | grep "^seed" | cut -c1-60
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k randomised_configurations 2>&1 | tail -2
