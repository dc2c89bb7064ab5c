# round 6, call 41: pinhole's iterations per workgroup (built-in 20) under the three-wave cap: gclm_set_sweep_iters 10 / 15 / 30 / 40, same allocation
O=gpurun_out/r06; mkdir -p $O
L=geocalib_amd/lib/libgeocalib_hip.so
timeout 900 python scripts/variant_probe.py --models pinhole --reps 3 --allocations 2 it20=$L it10=$L@10 it15=$L@15 it30=$L@30 it40=$L@40 2>&1 | grep -v amdgpu | cut -c1-150 > $O/variant_pinhole_iters.log; cat $O/variant_pinhole_iters.log
