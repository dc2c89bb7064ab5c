# round 6, call 13: simple_divisional on row pairs at 3 waves per SIMD (168 VGPRs, 276 B of scratch) against the shipped 2 waves (220 VGPRs)
O=gpurun_out/r06; mkdir -p $O
V=geocalib_amd/lib/variants
timeout 900 python scripts/variant_probe.py --models simple_divisional --reps 3 pairs2w=geocalib_amd/lib/libgeocalib_hip.so pairs3w=$V/div3.so 2>&1 | grep -v amdgpu > $O/variant_row_pairs_div_waves.log; cat $O/variant_row_pairs_div_waves.log
