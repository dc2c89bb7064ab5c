# round 6, call 5: fuzz 115/45 -- accuracy of the HIP sweep's evaluation at the states the draw passes through
O=gpurun_out/r06; mkdir -p $O
for st in 1 2 3; do timeout 300 python scripts/probes/system_probe.py 115 45 $st 0 2>&1 | grep -v amdgpu.ids; done > $O/system_probe_115_45.log 2>&1
cut -c1-330 $O/system_probe_115_45.log
