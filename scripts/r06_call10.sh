# round 6, call 10: iterations per workgroup of the row-pair walk (gclm_set_sweep_iters), same allocation; both models on row pairs
O=gpurun_out/r06; mkdir -p $O
V=geocalib_amd/lib/variants
timeout 900 python scripts/variant_probe.py --models simple_divisional,radial --reps 3 it15=$V/both.so it10=$V/both.so@10 it20=$V/both.so@20 it30=$V/both.so@30 it8=$V/both.so@8 onerow=$V/norp.so 2>&1 | grep -v amdgpu > $O/variant_row_pairs_iters.log; cat $O/variant_row_pairs_iters.log
