O=gpurun_out/r06; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -k "row_pairs or bench_single_gpu" > $O/pytest_gpu_call9.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu_call9.log | tail -5
grep -E "^E " $O/pytest_gpu_call9.log | head -20 | cut -c1-600
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_measured.json")) if __import__("os").path.exists("gpurun_out/parity_measured.json") else {}
for k, v in d.items():
    if k.startswith("row_pairs"): print(k, v)
PY
