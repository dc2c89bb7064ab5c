# round 6, call 12: radial on row pairs at 3 waves per SIMD (168 VGPRs, 16 B of scratch) against 2 waves (172 VGPRs) and the one-row walk
O=gpurun_out/r06; mkdir -p $O
V=geocalib_amd/lib/variants
timeout 900 python scripts/variant_probe.py --models radial --reps 4 pairs3w=$V/both3.so pairs2w=$V/both2.so onerow=$V/norp.so 2>&1 | grep -v amdgpu > $O/variant_row_pairs_radial_waves.log; cat $O/variant_row_pairs_radial_waves.log
