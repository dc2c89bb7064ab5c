# round 6, call 27: the SHIPPED kernels (with their arithmetic) at fewer waves per SIMD than their registers allow (GCLM_DYN_LDS caps the
# workgroups per CU): pinhole 6 -> 5 / 4 / 3, simple_radial 4 -> 3.  (the no-math build streams 1-3 % faster at 2-3 waves than at 6-8: call 26)
O=gpurun_out/r06; mkdir -p $O
V=geocalib_amd/lib/variants
timeout 900 python scripts/variant_probe.py --models pinhole,simple_radial --reps 3 --allocations 2 shipped=geocalib_amd/lib/libgeocalib_hip.so w5=$V/m5.so w4=$V/m4.so w3=$V/m3.so 2>&1 | grep -v amdgpu | cut -c1-150 > $O/variant_occupancy_math.log; cat $O/variant_occupancy_math.log
