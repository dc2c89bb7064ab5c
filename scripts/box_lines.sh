#!/bin/bash
# GPU box: the four B = 1024 bench lines of ONE box in one compact JSON line (appended to gpurun_out/box_lines.jsonl) -- each gpurun
# call lands on another physical MI355X, and the same binary's sweep moves by several per cent from box to box; this
# is the sample of that spread the round's numbers are read against.   usage: gpurun -- 'scripts/box_lines.sh <tag>'
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=${1:-box}
ID=$(rocm-smi --showuniqueid 2>/dev/null | grep -o "0x[0-9a-f]*" | head -1)
for m in pinhole simple_radial radial simple_divisional; do
  python bench.py --camera-model $m --steps 10 --warmup 2 --cpu-sample 0 --no-secondary --no-overlap --placement-tries 1 > gpurun_out/_box_$m.json 2>/dev/null
done
python - <<PY
import json
out = {"tag": "$TAG", "gpu_unique_id": "$ID"}
for m in ("pinhole", "simple_radial", "radial", "simple_divisional"):
    d = json.loads([l for l in open("gpurun_out/_box_%s.json" % m) if l.startswith("{")][-1])
    r = d["roofline"]
    out[m] = {"images_per_s": d["value"], "sweep_frac": r["frac"], "read_ceiling_frac": r.get("read_ceiling_frac"), "avg_launch_ms": r["avg_launch_ms"]}
line = json.dumps(out)
open("gpurun_out/box_lines.jsonl", "a").write(line + "\n")
print(line)
PY
rm -f gpurun_out/_box_*.json
