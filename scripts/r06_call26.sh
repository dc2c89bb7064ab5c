# round 6, call 26: what the sweep's access pattern streams as a function of the waves per SIMD -- the no-math build (GCLM_NOMATH: the
# loop, the loads and the reduction, no per-pixel arithmetic) with its workgroups per CU capped by dynamic LDS (GCLM_DYN_LDS): 8 / 6 / 5 /
# 4 / 3 / 2 waves per SIMD = pinhole's 6, simple_radial's 4, radial's 3, simple_divisional's 2.  Same allocation, two allocations.
O=gpurun_out/r06; mkdir -p $O
V=geocalib_amd/lib/variants
timeout 900 python scripts/variant_probe.py --models pinhole --reps 3 --allocations 2 w8=$V/nm8.so w6=$V/nm6.so w5=$V/nm5.so w4=$V/nm4.so w3=$V/nm3.so w2=$V/nm2.so shipped=geocalib_amd/lib/libgeocalib_hip.so 2>&1 | grep -v amdgpu | cut -c1-150 > $O/variant_occupancy.log; cat $O/variant_occupancy.log
