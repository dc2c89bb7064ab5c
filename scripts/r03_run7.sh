#!/bin/bash
# extra soak: seeds without reference goldens for simple_divisional (oracle-loose gate there)
cd ${GRAFT_REPO_ROOT:-.}
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cp gpurun_out/r03_fuzz_soak.txt /tmp/keep_soak.txt 2>/dev/null
rm -f gpurun_out/r03_fuzz_soak.txt
timeout 2400 scripts/fuzz_soak.sh 23 62 300 2>&1 | grep "^seed" | cut -c1-1400
mv gpurun_out/r03_fuzz_soak.txt gpurun_out/r03_fuzz_soak_extra.txt
