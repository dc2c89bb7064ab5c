#!/bin/bash
# round-6 evidence in ONE gpurun call (final binary): smoke, the full -m gpu suite with its measured-parity log, the driver's own
# command twice, self-launched two-rank lines (gloo ranks sharing this GPU: the N > 1 code path with its parity block, not a
# scaling number), the read + write ceilings next to the field kernels, and the round's profile set (bench lines, rocprofv3
# kernel stats, HBM PMC passes keyed per sweep instantiation, SQ PMC passes, one-rank RCCL lines, single-image latency).
#   gpurun --timeout 3000 -- 'scripts/r06_collect.sh'   then   python scripts/keep_profiles.py r06
cd ${GRAFT_REPO_ROOT:-.}
T=r06
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export GCLM_PARITY_LOG=$PWD/gpurun_out/$T/parity_measured.json
rm -f $GCLM_PARITY_LOG
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 600 > gpurun_out/$T/pytest_gpu_full.log 2>&1
grep -h "^fuzz seed\|^stop_at" gpurun_out/$T/pytest_gpu_full.log | cut -c1-900
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/$T/pytest_gpu_full.log | tail -4
tail -150 gpurun_out/$T/pytest_gpu_full.log > gpurun_out/$T/pytest_gpu.log; rm -f gpurun_out/$T/pytest_gpu_full.log
unset GCLM_PARITY_LOG
echo "=== the driver's command (N = 1), twice"
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench_driver_cmd_run$i.json 2> gpurun_out/$T/bench_driver_cmd_run$i.err; python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/$T/bench_driver_cmd_run$i.json") if l.startswith("{")][-1])
r, s = d["roofline"], d["secondary"]
sr, sh = s["simple_radial_B1024"], s["shared16_pinhole"]
vo = lambda c: "%.1e/%.1e/%.1e %s" % (c["max_focal_rel"], c["max_gravity_abs"], c["max_final_cost_rel"], "ok" if c["within_gate"] else "BEYOND THE GATE")
print("run $i: %.0f img/s (%.3f ms), sweep %.4f = %.3f of this allocation's read ceiling %.4f, whole job %.4f | vs oracle %s | best of n %s | overlap %.0f (%s)" % (
    d["value"], d["ms_per_step"], r["frac"], r["frac_of_read_ceiling"], r["read_ceiling_frac"], r["whole_job_frac"], vo(d["check"]["vs_oracle"]), d["placement"].get("best_of_n"),
    d["overlap"]["value"], d["overlap"]["bit_identical"]))
print("       simple_radial %.0f (%.4f; plane off %.4f, on again %.4f, bits %s; %.3f of its ceiling; vs oracle %s) | shared16 %.0f (%.4f; %.3f of its ceiling; vs oracle %s) | cpu %s %.1f" % (
    sr["value"], sr["roofline"]["frac"], sr["slat_off"]["frac"], sr["slat_off"]["on_again"]["frac"], sr["slat_off"]["bit_identical"], sr["roofline"]["frac_of_read_ceiling"], vo(sr["check"]["vs_oracle"]),
    sh["value"], sh["roofline"]["frac"], sh["roofline"]["frac_of_read_ceiling"], vo(sh["check"]["vs_oracle"]), d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"]))
ra = s.get("radial_B1024")
if ra:
    print("       radial %.0f (%.4f; one-row walk %.0f / %.4f, row pairs again %.4f; %.3f of its ceiling; vs oracle %s)" % (
        ra["value"], ra["roofline"]["frac"], ra["row_pairs_off"]["value"], ra["row_pairs_off"]["frac"], ra["row_pairs_off"]["on_again"]["frac"], ra["roofline"]["frac_of_read_ceiling"], vo(ra["check"]["vs_oracle"])))
sd = s["simple_divisional_B1024"]; rp = sd["row_pairs_off"]; c = sd["check"]["vs_oracle"]
ys = c.get("yardstick", {})
print("       simple_divisional %.0f (%.4f; one-row walk %.0f / %.4f, row pairs again %.4f; vs oracle: %d of %d images within 1e-4, medians %.1e/%.1e/%.1e, worst %.1e/%.1e/%.1e; the oracle's own float32 vs float64 exceeds 1e-4 on %s images, %s of %d within 1e-4 + 10 x that)" % (
    sd["value"], sd["roofline"]["frac"], rp["value"], rp["frac"], rp["on_again"]["frac"], c["images_within_gate"], c["images"],
    c["median_focal_rel"], c["median_gravity_abs"], c["median_final_cost_rel"], c["max_focal_rel"], c["max_gravity_abs"], c["max_final_cost_rel"],
    ys.get("images_where_it_exceeds_gate"), ys.get("images_within_gate_plus_10x_own"), c["images"]))
PY
done
echo "=== bench.py --gpus 2 without a launcher (two gloo ranks sharing this GPU), image sharding and the frame split"
python bench.py --gpus 2 --backend gloo --batch 512 --steps 5 --warmup 2 --cpu-sample 16 > gpurun_out/$T/bench_selflaunch_2ranks_gloo.json 2> gpurun_out/$T/bench_selflaunch_2ranks_gloo.err
python bench.py --gpus 2 --backend gloo --batch 512 --steps 5 --warmup 2 --cpu-sample 16 --shared-group 16 > gpurun_out/$T/bench_selflaunch_2ranks_gloo_split.json 2> gpurun_out/$T/bench_selflaunch_2ranks_gloo_split.err
python - <<PY
import json
for n in ("", "_split"):
    d = json.loads([l for l in open("gpurun_out/$T/bench_selflaunch_2ranks_gloo%s.json" % n) if l.startswith("{")][-1])
    print("2 gloo ranks%s: %.0f img/s, per-rank frac %s, parity %s, vs_oracle %s" % (n, d["value"], d["roofline"]["per_rank_frac"], d["multi_gpu"]["parity"], d["check"].get("vs_oracle")))
PY
python bench.py --gpus 2 --batch 512 > gpurun_out/$T/bench_selflaunch_2ranks_nccl_on_1gpu.json 2>/dev/null; cat gpurun_out/$T/bench_selflaunch_2ranks_nccl_on_1gpu.json
echo "=== ceilings: five planes read (the sweep's pattern), five planes read + written in place (pack_fields' pattern); the field kernels"
[ -x scripts/probes/_build/read_bench ] && timeout 300 scripts/probes/_build/read_bench > gpurun_out/$T/read_bench.log 2>&1; grep "4 units" gpurun_out/$T/read_bench.log
[ -x scripts/probes/_build/pack_bench ] && timeout 300 scripts/probes/_build/pack_bench > gpurun_out/$T/pack_bench.log 2>&1; grep "copy\|grid 300" gpurun_out/$T/pack_bench.log
timeout 300 python scripts/probes/fields_probe.py --json gpurun_out/$T/fields_kernels.json 2>&1 | grep "upsample_fields'\|pack\|ceiling" | cut -c1-220
echo "=== profiles"
timeout 1800 scripts/gpu_profile_all.sh $T 2>&1 | grep -v amdgpu.ids | grep -v "^E2026\|^W2026" | tail -80
