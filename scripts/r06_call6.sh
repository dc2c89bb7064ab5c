# round 6, call 6: the row-pair walk (gclm_set_row_pairs) -- agreement with the one-row walk, then same-allocation A/B
O=gpurun_out/r06; mkdir -p $O
timeout 600 python scripts/probes/row_pairs_probe.py 2>&1 | grep -v amdgpu.ids > $O/row_pairs_probe.log; cat $O/row_pairs_probe.log | cut -c1-300


