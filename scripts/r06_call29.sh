# round 6, call 29: pinhole's sweep held to 3 waves per SIMD by an LDS reservation (GCLM_PINHOLE_LDS = 51200, shipped) against the same tree
# without it: same-allocation A/B at B = 1024 / 128 / 4096, the driver's command, the -m gpu suite
O=gpurun_out/r06; mkdir -p $O
V=geocalib_amd/lib/variants
for B in 1024 128 4096; do
echo "== B = $B"
timeout 900 python scripts/variant_probe.py --models pinhole --batch $B --reps 3 --allocations 2 cap3=geocalib_amd/lib/libgeocalib_hip.so nocap=$V/nocap.so 2>&1 | grep -v amdgpu | cut -c1-150
done > $O/variant_pinhole_cap.log 2>&1; cat $O/variant_pinhole_cap.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_cap3.json 2>/dev/null; python - <<PY
import json
d = json.loads([l for l in open("$O/bench_driver_cmd_cap3.json") if l.startswith("{")][-1])
r = d["roofline"]; s = d["secondary"]
print("driver cmd: %.0f img/s, sweep %.4f = %.3f of read ceiling %.4f, whole job %.4f, overlap %.0f (%s), shared16 %.0f (%.4f), simple_radial %.0f (%.4f)" % (d["value"], r["frac"], r["frac_of_read_ceiling"], r["read_ceiling_frac"], r["whole_job_frac"], d["overlap"]["value"], d["overlap"]["bit_identical"], s["shared16_pinhole"]["value"], s["shared16_pinhole"]["roofline"]["frac"], s["simple_radial_B1024"]["value"], s["simple_radial_B1024"]["roofline"]["frac"]))
PY
GCLM_LIB_PATH=$PWD/$V/nocap.so python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $O/bench_driver_cmd_nocap.json 2>/dev/null; python - <<PY
import json
d = json.loads([l for l in open("$O/bench_driver_cmd_nocap.json") if l.startswith("{")][-1])
r = d["roofline"]
print("same box, without the cap: %.0f img/s, sweep %.4f = %.3f of read ceiling %.4f, whole job %.4f, overlap %.0f" % (d["value"], r["frac"], r["frac_of_read_ceiling"], r["read_ceiling_frac"], r["whole_job_frac"], d["overlap"]["value"]))
PY
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu_call29.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest_gpu_call29.log | tail -5
