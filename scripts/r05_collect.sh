#!/bin/bash
# round-5 evidence in one gpurun call (final binary): smoke, full -m gpu suite with the measurement log, the driver's own
# command twice (plain / self-launched two ranks over gloo), the kernels either side of the path, a 20-seed soak, the
# round's profile set (bench lines, rocprofv3 kernel stats, HBM + SQ PMC passes, one-rank RCCL lines, latency).
#   gpurun --timeout 2700 -- 'scripts/r05_collect.sh'   then   python scripts/keep_profiles.py r05
cd ${GRAFT_REPO_ROOT:-.}
T=r05
mkdir -p gpurun_out/$T
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export GCLM_PARITY_LOG=$PWD/gpurun_out/$T/parity_measured.json
rm -f $GCLM_PARITY_LOG
timeout 1200 python -m pytest tests -m gpu -q -s --timeout 300 > gpurun_out/$T/pytest_gpu_full.log 2>&1
grep -h "^fuzz seed\|^stop_at" gpurun_out/$T/pytest_gpu_full.log | cut -c1-900
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/$T/pytest_gpu_full.log | tail -4
tail -150 gpurun_out/$T/pytest_gpu_full.log > gpurun_out/$T/pytest_gpu.log; rm -f gpurun_out/$T/pytest_gpu_full.log
unset GCLM_PARITY_LOG
echo "=== the driver's command (N = 1), twice"
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench_driver_cmd_run$i.json 2> gpurun_out/$T/bench_driver_cmd_run$i.err; python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/$T/bench_driver_cmd_run$i.json") if l.startswith("{")][-1])
p = d["placement"]
print("run $i: %.0f img/s (%.3f ms) on the FIRST allocation, sweep frac %.4f, whole-job %.4f | best of n %s | overlap %.0f img/s (%s) | simple_radial %.0f (frac %.4f) | shared16 %.0f (frac %.4f) | cpu %s %.1f" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_job_frac"], p.get("best_of_n"), d["overlap"]["value"], d["overlap"]["bit_identical"],
    d["secondary"]["simple_radial_B1024"]["value"], d["secondary"]["simple_radial_B1024"]["roofline"]["frac"], d["secondary"]["shared16_pinhole"]["value"],
    d["secondary"]["shared16_pinhole"]["roofline"]["frac"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"]))
PY
done
echo "=== bench.py --gpus 2 without a launcher (two gloo ranks sharing this GPU: the N > 1 code path, not a scaling number)"
python bench.py --gpus 2 --backend gloo --batch 512 --steps 5 --warmup 2 --cpu-sample 16 > gpurun_out/$T/bench_selflaunch_2ranks_gloo.json 2> gpurun_out/$T/bench_selflaunch_2ranks_gloo.err; tail -c 900 gpurun_out/$T/bench_selflaunch_2ranks_gloo.json
python bench.py --gpus 2 --batch 512 > gpurun_out/$T/bench_selflaunch_2ranks_nccl_on_1gpu.json 2>/dev/null; cat gpurun_out/$T/bench_selflaunch_2ranks_nccl_on_1gpu.json
echo "=== the kernels either side of the path"
timeout 300 python scripts/probes/fields_probe.py --json gpurun_out/$T/fields_kernels.json 2>&1 | grep "upsample\|pack" | cut -c1-200
[ -x scripts/probes/_build/upsample_bench ] && timeout 300 scripts/probes/_build/upsample_bench 20 > gpurun_out/$T/upsample_bench.log 2>&1
timeout 300 python scripts/probes/calibrate_probe.py > gpurun_out/$T/calibrate_probe.log 2>&1; tail -6 gpurun_out/$T/calibrate_probe.log
echo "=== soak (20 seeds x 300 draws)"
rm -f gpurun_out/${T}_fuzz_soak.txt
SOAK_TAG=$T timeout 2000 scripts/fuzz_soak.sh 11 30 300 2>&1 | grep "^seed" | cut -c1-200
echo "=== profiles"
timeout 1500 scripts/gpu_profile_all.sh $T 2>&1 | grep -v amdgpu.ids | grep -v "^E2026\|^W2026" | tail -70
