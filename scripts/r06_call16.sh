# round 6, call 16: simple_radial with its up-field sums in the z basis (-DGCLM_ZBASIS=1: 249 -> 232 VALU per 4 px) against the shipped build, same allocation
O=gpurun_out/r06; mkdir -p $O
timeout 900 python scripts/variant_probe.py --models simple_radial --reps 4 --allocations 2 zbasis=geocalib_amd/lib/variants/zb.so shipped=geocalib_amd/lib/libgeocalib_hip.so 2>&1 | grep -v amdgpu > $O/variant_zbasis.log; cat $O/variant_zbasis.log
