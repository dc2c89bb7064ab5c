# round 6, call 28: pinhole's shipped kernel at 3 / 2 / 1 waves per SIMD (GCLM_DYN_LDS), B = 1024 and B = 256
O=gpurun_out/r06; mkdir -p $O
V=geocalib_amd/lib/variants
for B in 1024 256; do
echo "== B = $B"
timeout 900 python scripts/variant_probe.py --models pinhole --batch $B --reps 3 --allocations 2 shipped=geocalib_amd/lib/libgeocalib_hip.so w3=$V/m3.so w2=$V/m2.so w1=$V/m1.so 2>&1 | grep -v amdgpu | cut -c1-150
done > $O/variant_occupancy_pinhole.log 2>&1; cat $O/variant_occupancy_pinhole.log
