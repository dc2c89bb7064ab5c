# round 6, call 8: the row-pair walk as the built-in choice for simple_divisional -- the -m gpu suite, the same-allocation A/B
# against a build without it (radial through the knob: both libraries at gclm_set_row_pairs default; see variant_row_pairs.log
# of call 6 for radial), the driver's command
O=gpurun_out/r06; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu_call8.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu_call8.log | tail -5
grep -E "^E " $O/pytest_gpu_call8.log | head -20 | cut -c1-600
L=geocalib_amd/lib/libgeocalib_hip.so
timeout 900 python scripts/variant_probe.py --models simple_divisional,radial,simple_radial --reps 3 pairs=$L onerow=geocalib_amd/lib/variants/norp.so 2>&1 | grep -v amdgpu > $O/variant_row_pairs_final.log; cat $O/variant_row_pairs_final.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_call8.json 2> $O/bench_driver_cmd_call8.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_driver_cmd_call8.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "ceiling", d["roofline"].get("read_ceiling_frac"), "vs_oracle", d["check"]["vs_oracle"]["within_gate"])
for k, r in d["secondary"].items():
    print(k, r["value"], r["roofline"]["frac"], r["roofline"].get("read_ceiling_frac"), {a: b for a, b in r["check"]["vs_oracle"].items() if a != "against"})
    for c in ("slat_off", "row_pairs_off"):
        if c in r: print("   ", c, {a: b for a, b in r[c].items() if a != "what"})
PY
