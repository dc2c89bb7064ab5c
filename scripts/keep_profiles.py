"""Copy the summaries of a scripts/gpu_profile_all.sh run from gpurun_out/<tag>/ (scratch) into profiles/ (tracked).
usage: python scripts/keep_profiles.py <tag>"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
kept = []
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    line = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if not line:
        continue
    name = os.path.basename(f)[len("bench_"):-len(".json")]      # "traced_<name>": the line printed by the rocprofv3-traced process
    with open(os.path.join(dst, f"{tag}_bench_{name}.json"), "w") as fh:
        json.dump(json.loads(line[-1]), fh, indent=1)
    kept.append(f"{tag}_bench_{name}.json")
for d in sorted(glob.glob(os.path.join(src, "kt_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[3:]
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats_{name}.csv"))
        kept.append(f"{tag}_kernel_stats_{name}.csv")
for f in sorted(glob.glob(os.path.join(src, "pmc_summary_*.json"))):
    name = os.path.basename(f)[len("pmc_summary_"):-5]
    shutil.copy(f, os.path.join(dst, f"{tag}_pmc_hbm_{name}.json")); kept.append(f"{tag}_pmc_hbm_{name}.json")
for f in sorted(glob.glob(os.path.join(src, "pmc_sq_summary_*.json"))):
    name = os.path.basename(f)[len("pmc_sq_summary_"):-5]
    shutil.copy(f, os.path.join(dst, f"{tag}_pmc_sq_{name}.json")); kept.append(f"{tag}_pmc_sq_{name}.json")
for extra in glob.glob(os.path.join(src, "power_*.json")) + glob.glob(os.path.join(src, "ab_*.log")):
    shutil.copy(extra, os.path.join(dst, f"{tag}_{os.path.basename(extra)}")); kept.append(f"{tag}_{os.path.basename(extra)}")
if os.path.exists(os.path.join(src, "paced.json")):
    shutil.copy(os.path.join(src, "paced.json"), os.path.join(dst, f"{tag}_paced.json")); kept.append(f"{tag}_paced.json")
if os.path.exists(os.path.join(src, "latency.json")):
    shutil.copy(os.path.join(src, "latency.json"), os.path.join(dst, f"{tag}_latency.json")); kept.append(f"{tag}_latency.json")
# HBM traffic table read by bench.py (roofline.traffic): FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, KB -> bytes
tpath = os.path.join(dst, "pmc_traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for f in sorted(glob.glob(os.path.join(src, "pmc_summary_*.json"))):
    name = os.path.basename(f)[len("pmc_summary_"):-5]
    s = json.load(open(f))
    if any(k.startswith("FETCH_SIZE") for k in s) and any(k.startswith("WRITE_SIZE") for k in s) and "_B" in name:
        model, b = name.rsplit("_B", 1)
        # per instantiation (FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes), and per SOLVE: the launch-weighted mean over all sweep
        # instantiations of the run (20 loop sweeps + 1 final sweep; simple_radial: 1 plane-writing + 20 plane-reading)
        inst = {}
        for k, v in s.items():
            counter, kern = k.split(":", 1)
            inst.setdefault(kern, {})[counter] = v
        per = {}
        for kern, c in inst.items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                per[kern] = {"hbm_bytes_per_launch": int(round((2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024)),
                             "fetch_size_kb": c["FETCH_SIZE"]["mean"], "write_size_kb": c["WRITE_SIZE"]["mean"],
                             "launches_fetch_pass": c["FETCH_SIZE"]["n"], "launches_write_pass": c["WRITE_SIZE"]["n"]}
        fs = [(v["mean"], v["n"]) for k, v in s.items() if k.startswith("FETCH_SIZE")]
        ws = [(v["mean"], v["n"]) for k, v in s.items() if k.startswith("WRITE_SIZE")]
        fm = sum(a * n for a, n in fs) / sum(n for _, n in fs)
        wm = sum(a * n for a, n in ws) / sum(n for _, n in ws)
        traffic[f"{model}_B{b}_640x480"] = {"hbm_bytes_per_launch": int(round((2 * fm + wm) * 1024)), "fetch_size_kb": fm, "write_size_kb": wm,
                                              "what": "mean over ALL sweep launches of the solves under the counters (per-solve figure); "
                                                      "`per_instantiation` splits it by template arguments",
                                              "per_instantiation": per,
                                              "correction": "FETCH_SIZE x 2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM)", "source": f"profiles/{tag}_pmc_hbm_{name}.json"}
json.dump(traffic, open(tpath, "w"), indent=1)
print("\n".join(kept))
