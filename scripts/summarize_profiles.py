"""Print the numbers DESIGN.md section 6 quotes, straight out of profiles/<tag>_* (so that the tables are copied, not typed).
usage: python scripts/summarize_profiles.py [tag]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(ROOT, "profiles")


def load(name):
    path = os.path.join(P, f"{tag}_{name}")
    return json.load(open(path)) if os.path.exists(path) else None


def rocprof_avg_ms(name, model_id):
    path = os.path.join(P, f"{tag}_kernel_stats_{name}.csv")
    if not os.path.exists(path):
        return None
    for row in csv.DictReader(open(path)):
        if f"sweep_kernel<{model_id}," in row["Name"]:
            return float(row["AverageNs"]) / 1e6
    return None


print("| line | images/s | ms per solve of the batch | mean sweep (rocprofv3) | sweep / 8 TB/s | whole job / 8 TB/s | two streams |")
for name, mid in (("pinhole_B1024", 0), ("simple_radial_B1024", 1), ("radial_B1024", 2), ("simple_divisional_B1024", 3),
                  ("shared16_pinhole", 0), ("shared16_simple_radial", 1), ("pinhole_B8192", 0)):
    d = load(f"bench_{name}.json")
    if d is None:
        continue
    r = d["roofline"]
    ov = d.get("overlap")
    rp = rocprof_avg_ms(name, mid)
    print(f"| {name} | {d['value']:.0f} | {d['ms_per_step']:.2f} | {r['avg_launch_ms']:.4f} ({rp:.4f}) | {r['frac']:.3f} | {r['whole_job_frac']:.3f} | "
          + (f"{ov['value']:.0f} ({ov['whole_job_frac']:.3f}, bit-identical {ov['bit_identical']})" if ov else "–") + " |")
for i in (1, 2):
    d = load(f"bench_driver_cmd_run{i}.json")
    if d is None:
        continue
    s = d["secondary"]
    sr = next(v for k, v in s.items() if k.startswith("simple_radial"))
    print(f"driver cmd run {i}: {d['value']:.0f} img/s (sweep {d['roofline']['frac']:.4f}, whole {d['roofline']['whole_job_frac']:.4f}), first allocation "
          f"{d['placement'].get('first_allocation')}, overlap {d['overlap']['value']:.0f} ({d['overlap']['bit_identical']}), simple_radial {sr['value']:.0f} "
          f"({sr['roofline']['frac']:.4f}), shared16 {s['shared16_pinhole']['value']:.0f} ({s['shared16_pinhole']['roofline']['frac']:.4f}), "
          f"cpu {d['cpu_baseline']['kind']} {d['cpu_baseline']['value']:.1f} img/s on {d['cpu_baseline']['cores']} cores")
    sd = next((v for k, v in s.items() if k.startswith("simple_divisional")), None)
    if sd:
        rp, c = sd["row_pairs_off"], sd["check"]["vs_oracle"]
        print(f"   simple_divisional {sd['value']:.0f} ({sd['roofline']['frac']:.4f}, {sd['roofline'].get('frac_of_read_ceiling')} of its read ceiling); one-row walk "
              f"{rp['value']:.0f} ({rp['frac']:.4f}); row pairs again {rp['on_again']['frac']:.4f}; vs oracle {c['images_within_gate']} of {c['images']} within 1e-4, "
              f"medians {c['median_focal_rel']:.1e} / {c['median_gravity_abs']:.1e} / {c['median_final_cost_rel']:.1e}")
for m in ("pinhole", "simple_radial", "radial", "simple_divisional"):
    d = load(f"power_{m}.json")
    if d:
        s = d["summary"]
        print(f"power {m}: {s['busy_power_W_mean']:.0f} W, {s['busy_gfxclk_MHz_mean'] / 1e3:.2f} GHz")
d = load("paced.json")
if d:
    for r in d["rows"]:
        if r["height"] == 480 and r["paced_depth"] in (0, 3):
            print(f"latency {r['camera_model']} depth {r['paced_depth']}: {r['median_us_per_solve']} us (p10 {r['p10_us']})")
t = json.load(open(os.path.join(P, "pmc_traffic.json")))
for k, v in t.items():
    print(f"traffic {k}: {v['hbm_bytes_per_launch'] / 1e6:.1f} MB = {v['hbm_bytes_per_launch'] / 6291456000:.4f} x algorithmic ({v['source']})")
for n in ("independent_torch", "independent_rccl", "split512x2_torch", "split512x2", "split512x2_simple_radial", "sharedbygroup_rccl"):
    d = load(f"bench_1rank_rccl_{n}.json")
    if d:
        mg = d["multi_gpu"]
        print(f"1-rank RCCL {n}: {d['value']:.0f} img/s, collective {mg['collective_ms'] * 1e3:.1f} us per solve ({mg['collectives_per_step']} x {mg['collective_bytes']} B)")
for m in ("pinhole", "simple_radial"):
    d = load(f"pmc_sq_{m}_B1024.json")
    if d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        print(f"SQ {m}: {d['SQ_INSTS_VALU'] / 1e6:.1f} M VALU insts, VALU busy {d['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc:.3f} at {cyc / d['mean_duration_ns_under_pmc']:.2f} GHz, "
              f"{d['mean_duration_ns_under_pmc'] / 1e3:.1f} us under PMC")
