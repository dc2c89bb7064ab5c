# round 6, call 37: the four bench lines of one more box on the FINAL build, then socket power and shader clock of pinhole's sweep at 3 waves
# per SIMD (shipped) and at 6 (-DGCLM_PINHOLE_LDS=0), and of the three distortion models (the energy line of DESIGN 3.1)
O=gpurun_out/r06; mkdir -p $O
scripts/box_lines.sh box10 | cut -c1-900
timeout 300 python scripts/power_probe.py $O/power_pinhole_3waves.json --tag pinhole_3waves -- --camera-model pinhole --steps 28 --cpu-sample 0 --no-secondary --no-overlap --placement-tries 1 2>&1 | tail -2 | cut -c1-400
GCLM_LIB_PATH=$PWD/geocalib_amd/lib/variants/nocap.so timeout 300 python scripts/power_probe.py $O/power_pinhole_6waves.json --tag pinhole_6waves -- --camera-model pinhole --steps 28 --cpu-sample 0 --no-secondary --no-overlap --placement-tries 1 2>&1 | tail -2 | cut -c1-400
for m in simple_radial radial simple_divisional; do
  timeout 300 python scripts/power_probe.py $O/power_${m}_final.json --tag ${m}_final -- --camera-model $m --steps 28 --cpu-sample 0 --no-secondary --no-overlap --placement-tries 1 2>&1 | tail -2 | cut -c1-400
done
