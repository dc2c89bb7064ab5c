"""Per-model worst-case table of what the GPU parity suite MEASURED (build container or GPU box).

    GCLM_PARITY_LOG=gpurun_out/r03/parity_measured.json python -m pytest tests -m gpu      # on the GPU box
    python scripts/parity_report.py gpurun_out/r03/parity_measured.json [gpurun_out/fuzz_measured_*.json ...] --json profiles/archive/r03_parity.json

tests/conftest.compare_result (and the fuzz) record every distance they gate: focal (relative), distortion / gravity
(absolute), costs / covariance / uncertainties (relative to the largest entry).  This script groups the records by camera
model and yardstick (reference golden / oracle) and keeps the worst value per quantity next to the gate it was held to,
plus the fuzz statistics per seed."""
import collections
import json
import sys

MODELS = ("pinhole", "simple_radial", "radial", "simple_divisional")
QUANT = ("focal", "dist", "gravity", "cost", "cov", "unc")
REFERENCE_LABELS = ("/default", "/bench", "/euclid", "/linfocal", "/fixlambda", "/loss_scale", "/noconf", "/lat_only", "/scales",
                    "/prior_focal", "/prior_gravity", "/squared_loss", "/heuristic", "/shared", "full/", "cnn/", "shared16/", "shared16-split8/", "split/")


def model_of(label):
    for m in sorted(MODELS, key=len, reverse=True):
        if m in label:
            return m
    return "pinhole" if label.startswith(("cnn/", "B8192", "unaligned")) else None


def main():
    args = sys.argv[1:]
    out_json = None
    if "--json" in args:
        i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]
    rec = {}
    for path in args:
        rec.update(json.load(open(path)))
    table = {m: {"vs reference goldens": {}, "vs oracle": {}} for m in MODELS}
    counts = collections.Counter()
    for label, v in rec.items():
        if label.startswith(("fuzz/", "system/")) or "focal" not in v:
            continue
        m = model_of(label)
        if m is None:
            continue
        kind = "vs reference goldens" if any(t in label for t in REFERENCE_LABELS) and not label.startswith(("shared16x8", "prior_dist", "B1024", "B8192")) else "vs oracle"
        counts[(m, kind)] += 1
        row = table[m][kind]
        for q in QUANT:
            if q in v and (q not in row or v[q] > row[q]["worst"]):
                row[q] = {"worst": v[q], "case": label, "gate": v.get("tol", {}).get(q)}
    for m in MODELS:
        for kind in list(table[m]):
            table[m][kind]["cases"] = counts[(m, kind)]
    systems = {k: v for k, v in rec.items() if k.startswith("system/")}
    fuzz = collections.defaultdict(lambda: {"draws": 0, "undetermined": 0, "gated": 0, "by_yardstick": collections.Counter(),
                                            "worst": [0, 0, 0, 0], "worst_over_gate": 0.0, "beyond_gate_on_unstable_yardstick": []})
    for label, v in rec.items():
        if not label.startswith("fuzz/"):      # ("fuzz_summary/<seed>" is the test's own tally; recomputed here)
            continue
        f = fuzz[(label.split("/")[1], v["model"])]
        f["draws"] += 1
        if v.get("undetermined"):
            f["undetermined"] += 1
            if "first_step_spread" in v:          # compared after ONE LM step instead (round 5)
                f["undetermined_compared_at_first_step"] = f.get("undetermined_compared_at_first_step", 0) + 1
                f["first_step_worst_over_gate"] = max(f.get("first_step_worst_over_gate", 0.0),
                                                      max(s / t for s, t in zip(v["first_step_spread"], v["first_step_tol"])))
            continue
        if v.get("excused"):             # beyond its gate, and the oracle's own float32 evaluation shown unstable on the draw
            f["beyond_gate_on_unstable_yardstick"].append({"case": label, "spread": v["spread"], "gate": v["tol"], "diagnosis": v["excused"]})
            continue
        f["gated"] += 1
        f["by_yardstick"][v["yardstick"]] += 1
        f["worst"] = [max(a, b) for a, b in zip(f["worst"], v["spread"])]
        f["worst_over_gate"] = max(f["worst_over_gate"], max(s / t for s, t in zip(v["spread"], v["tol"])))
    fz = {}
    for (seed, m), f in sorted(fuzz.items()):
        fz.setdefault(seed, {})[m] = {**f, "by_yardstick": dict(f["by_yardstick"]),
                                      "fraction_compared_at_the_end": round(1.0 - f["undetermined"] / max(f["draws"], 1), 3),
                                      "fraction_compared_at_all": round(1.0 - (f["undetermined"] - f.get("undetermined_compared_at_first_step", 0)) / max(f["draws"], 1), 3)}
    out = {"what": "worst HIP-vs-yardstick distance per camera model, as recorded by the -m gpu suite on an MI355X "
                   "(focal: relative; dist, gravity: absolute; cost, cov, unc: relative to the largest entry); gate = the "
                   "tolerance the worst case was held to", "goldens_and_oracle": table,
           "single_sweep_system": systems, "fuzz [focal rel, gravity abs, dist abs, final-cost rel]": fz}
    for m in MODELS:
        for kind, row in table[m].items():
            if row.get("cases"):
                print(f"{m:18s} {kind:22s} ({row['cases']:3d} cases): " + "  ".join(
                    f"{q} {row[q]['worst']:.1e}" for q in QUANT if q in row))
    for seed, ms in fz.items():
        tot = {k: sum(v[k] for v in ms.values()) for k in ("draws", "undetermined", "gated")}
        exc = [e["case"] for v in ms.values() for e in v["beyond_gate_on_unstable_yardstick"]]
        print(f"fuzz seed {seed}: {tot}  worst/gate {max(v['worst_over_gate'] for v in ms.values()):.2f}  beyond the gate on an unstable yardstick: {exc}")
    if out_json:
        with open(out_json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
