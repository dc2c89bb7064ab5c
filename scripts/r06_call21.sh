# round 6, call 21: row pairs with the latitude sums of the two rows taken together (-DGCLM_LAT_PAIRS=1: radial 575 -> 553 VALU
# per 8 px, simple_divisional 650 -> 642) against the same tree without it, same allocation, two allocations
O=gpurun_out/r06; mkdir -p $O
timeout 900 python scripts/variant_probe.py --models radial,simple_divisional --reps 4 --allocations 2 latp=geocalib_amd/lib/variants/latp.so base=geocalib_amd/lib/variants/base.so 2>&1 | grep -v amdgpu | cut -c1-200 > $O/variant_lat_pairs.log; cat $O/variant_lat_pairs.log
