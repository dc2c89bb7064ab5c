# round 6, call 19: socket power and shader clock of the simple_divisional / radial sweeps with row pairs (shipped) and without
# (a build with -DGCLM_MIRROR_MODELS=0), scripts/power_probe.py around bench.py --steps 28
O=gpurun_out/r06; mkdir -p $O
for m in simple_divisional radial; do
  timeout 300 python scripts/power_probe.py $O/power_${m}_row_pairs.json --tag ${m}_row_pairs -- --camera-model $m --steps 28 --cpu-sample 0 --no-secondary --no-overlap --placement-tries 1 2>&1 | tail -2 | cut -c1-400
  GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/norp.so timeout 300 python scripts/power_probe.py $O/power_${m}_one_row.json --tag ${m}_one_row -- --camera-model $m --steps 28 --cpu-sample 0 --no-secondary --no-overlap --placement-tries 1 2>&1 | tail -2 | cut -c1-400
done
