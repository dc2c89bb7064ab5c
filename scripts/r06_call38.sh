# round 6, call 38: `secondary.radial_B1024` in the bench line (with its row_pairs_off control and vs_oracle): the bench test, then the driver's command
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bench_single_gpu_line" --timeout 600 2>&1 | tail -5 | cut -c1-400
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_radial_secondary.json 2>/dev/null ) 2>&1 | grep real
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_driver_cmd_radial_secondary.json") if l.startswith("{")][-1])
r = d["roofline"]; s = d["secondary"]
print("driver cmd: %.0f img/s, sweep %.4f = %.3f of read ceiling %.4f, whole job %.4f" % (d["value"], r["frac"], r["frac_of_read_ceiling"], r["read_ceiling_frac"], r["whole_job_frac"]))
for k, v in s.items():
    c = v["check"]["vs_oracle"]
    print("  %s: %.0f (%.4f, %.3f of its ceiling) vs oracle %d/%d within 1e-4, max %.1e/%.1e/%.1e %s" % (k, v["value"], v["roofline"]["frac"], v["roofline"]["frac_of_read_ceiling"], c["images_within_gate"], c["images"], c["max_focal_rel"], c["max_gravity_abs"], c["max_final_cost_rel"],
          ("| one-row %.0f / %.4f -> again %.4f" % (v["row_pairs_off"]["value"], v["row_pairs_off"]["frac"], v["row_pairs_off"]["on_again"]["frac"])) if "row_pairs_off" in v else ""))
PY
