# round 6, call 39: the first sweep of a distortion-model solve (1.45 ms whatever the model: 6.29 GB read + 1.26 GB plane written) with the
# plane stored through the caches instead of non-temporally (-DGCLM_SLAT_STORE_PLAIN=1), same allocation; mean over the 21 sweeps of a solve
O=gpurun_out/r06; mkdir -p $O
timeout 900 python scripts/variant_probe.py --models simple_radial,radial --reps 4 --allocations 2 nt=geocalib_amd/lib/libgeocalib_hip.so plain=geocalib_amd/lib/variants/plainst.so 2>&1 | grep -v amdgpu | cut -c1-200 > $O/variant_slat_store.log; cat $O/variant_slat_store.log
