# round 6, call 20: where does the row-pair walk start to pay?  same-allocation A/B at small and medium batches (the built-in
# choice pairs rows above 768 workgroups per launch = 77 images of 640x480)
O=gpurun_out/r06; mkdir -p $O; rm -f $O/variant_row_pairs_batches.log
for B in 80 96 128 192 256 512; do
  echo "== B = $B" >> $O/variant_row_pairs_batches.log
  timeout 600 python scripts/variant_probe.py --models simple_divisional,radial --batch $B --reps 3 pairs=geocalib_amd/lib/libgeocalib_hip.so onerow=geocalib_amd/lib/variants/norp.so 2>&1 | grep -v amdgpu | cut -c1-150 >> $O/variant_row_pairs_batches.log
done
cat $O/variant_row_pairs_batches.log
