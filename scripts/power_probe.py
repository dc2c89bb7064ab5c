"""GPU box: socket power, shader clock and throttle / violation status sampled at >= 10 Hz while bench.py runs.

usage: python scripts/power_probe.py OUT.json [--tag NAME] -- <bench.py arguments>
   e.g. python scripts/power_probe.py gpurun_out/r03_power_simple_radial.json -- --camera-model simple_radial --steps 28

Replaces the round-2 INFERENCE "the distortion models are power-limited" (shader clock from GRBM_GUI_ACTIVE only) by a
measurement: amdsmi's gpu_metrics (current_socket_power, current_gfxclks per XCD, throttle status, the PPT / thermal
residency accumulators), amdsmi_get_violation_status (which limiter is active, for what share of the time),
amdsmi_get_power_cap_info (the cap itself); sysfs hwmon as a fallback.  One sampling thread, ~25 Hz; the summary covers
the samples taken while the GPU was busy (power above idle + 30 % of the swing)."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain(v):
    """amdsmi returns nested dicts / lists / enums / 'N/A': make it JSON."""
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    return str(v)


class Sampler:
    def __init__(self):
        self.smi, self.h, self.err = None, None, []
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.smi = amdsmi
            self.h = amdsmi.amdsmi_get_processor_handles()[0]
        except Exception as e:  # noqa: BLE001
            self.err.append(f"amdsmi: {e!r}")
        self.hwmon = None
        for card in sorted(os.listdir("/sys/class/drm")) if os.path.isdir("/sys/class/drm") else []:
            base = f"/sys/class/drm/{card}/device/hwmon"
            if os.path.isdir(base):
                for hm in os.listdir(base):
                    if os.path.exists(f"{base}/{hm}/power1_average") or os.path.exists(f"{base}/{hm}/power1_input"):
                        self.hwmon = f"{base}/{hm}"
                        break
            if self.hwmon:
                break

    def static(self):
        out = {"errors": self.err, "hwmon": self.hwmon}
        if self.smi:
            for name, fn in (("power_cap", "amdsmi_get_power_cap_info"), ("asic", "amdsmi_get_gpu_asic_info"),
                             ("power_management", "amdsmi_is_gpu_power_management_enabled")):
                try:
                    out[name] = _plain(getattr(self.smi, fn)(self.h))
                except Exception as e:  # noqa: BLE001
                    out[name] = f"unavailable: {e!r}"
        return out

    def sample(self):
        s = {"t": time.perf_counter()}
        if self.smi:
            try:
                m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
                for k in ("current_socket_power", "average_socket_power", "current_gfxclk", "average_gfxclk_frequency",
                          "current_uclk", "average_uclk_frequency", "temperature_hotspot", "temperature_mem",
                          "throttle_status", "indep_throttle_status", "average_gfx_activity", "average_umc_activity",
                          "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc",
                          "vr_thm_residency_acc", "hbm_thm_residency_acc", "gfxclk_lock_status"):
                    if k in m:
                        s[k] = _plain(m[k])
                if "current_gfxclks" in m:
                    v = [x for x in _plain(m["current_gfxclks"]) if isinstance(x, (int, float)) and 0 < x < 60000]
                    if v:
                        s["gfxclk_xcd_mean"], s["gfxclk_xcd_min"], s["gfxclk_xcd_max"] = sum(v) / len(v), min(v), max(v)
            except Exception as e:  # noqa: BLE001
                s["metrics_error"] = repr(e)
            try:
                s["power_info"] = _plain(self.smi.amdsmi_get_power_info(self.h))
            except Exception:  # noqa: BLE001
                pass
        if self.hwmon:
            for k, f in (("hwmon_power_uW", "power1_average"), ("hwmon_power_uW", "power1_input"), ("hwmon_sclk_Hz", "freq1_input")):
                try:
                    s[k] = int(open(f"{self.hwmon}/{f}").read())
                except Exception:  # noqa: BLE001
                    pass
        return s

    def violations(self):
        if not self.smi:
            return None
        try:
            return _plain(self.smi.amdsmi_get_violation_status(self.h))
        except Exception as e:  # noqa: BLE001
            return f"unavailable: {e!r}"


def power_of(s):
    for k in ("current_socket_power", "average_socket_power"):
        if isinstance(s.get(k), (int, float)) and 0 < s[k] < 5000:
            return float(s[k])
    p = s.get("power_info")
    if isinstance(p, dict):
        for k in ("current_socket_power", "average_socket_power", "socket_power"):
            if isinstance(p.get(k), (int, float)) and 0 < p[k] < 5000:
                return float(p[k])
    if "hwmon_power_uW" in s:
        return s["hwmon_power_uW"] / 1e6
    return None


def clock_of(s):
    for k in ("gfxclk_xcd_mean", "current_gfxclk", "average_gfxclk_frequency"):
        if isinstance(s.get(k), (int, float)) and 0 < s[k] < 60000:
            return float(s[k])
    if "hwmon_sclk_Hz" in s:
        return s["hwmon_sclk_Hz"] / 1e6
    return None


def main():
    args = sys.argv[1:]
    out_path = args.pop(0)
    tag = None
    if args and args[0] == "--tag":
        tag = args[1]
        args = args[2:]
    if args and args[0] == "--":
        args = args[1:]
    sm = Sampler()
    samples, stop = [], threading.Event()

    def loop():
        while not stop.is_set():
            samples.append(sm.sample())
            time.sleep(0.03)

    v0 = sm.violations()
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    time.sleep(1.0)                                     # idle baseline
    t0 = time.perf_counter()
    env = dict(os.environ)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-sample", "0", *args], capture_output=True,
                         text=True, cwd=ROOT, env=env)
    t1 = time.perf_counter()
    time.sleep(0.5)
    stop.set()
    th.join()
    v1 = sm.violations()
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    bench = json.loads(line[-1]) if line else {"error": res.stderr[-1500:]}
    pw = [(s["t"], power_of(s), clock_of(s)) for s in samples]
    pw = [(t, p, c) for t, p, c in pw if p is not None]
    summary = {"samples": len(samples), "rate_hz": len(samples) / max(samples[-1]["t"] - samples[0]["t"], 1e-9) if len(samples) > 1 else 0}
    if pw:
        idle = sorted(p for _, p, _ in pw)[max(0, len(pw) // 20)]
        peak = max(p for _, p, _ in pw)
        busy = [(t, p, c) for t, p, c in pw if p > idle + 0.3 * (peak - idle) and t0 <= t <= t1]
        summary |= {"idle_power_W": idle, "peak_power_W": peak, "busy_samples": len(busy)}
        if busy:
            ps = sorted(p for _, p, _ in busy)
            cs = sorted(c for _, _, c in busy if c is not None)
            summary |= {"busy_power_W_mean": sum(ps) / len(ps), "busy_power_W_p10": ps[len(ps) // 10], "busy_power_W_p90": ps[len(ps) * 9 // 10]}
            if cs:
                summary |= {"busy_gfxclk_MHz_mean": sum(cs) / len(cs), "busy_gfxclk_MHz_p10": cs[len(cs) // 10], "busy_gfxclk_MHz_p90": cs[len(cs) * 9 // 10]}
    for k in ("ppt_residency_acc", "socket_thm_residency_acc", "prochot_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "accumulation_counter"):
        vals = [s[k] for s in samples if isinstance(s.get(k), (int, float))]
        if vals:
            summary[f"{k}_delta"] = vals[-1] - vals[0]
    ts = sorted({str(s.get("throttle_status")) for s in samples} | {str(s.get("indep_throttle_status")) for s in samples})
    summary["throttle_status_values_seen"] = ts
    out = {"tag": tag, "command": "bench.py --cpu-sample 0 " + " ".join(args), "static": sm.static(), "summary": summary,
           "violation_status_before": v0, "violation_status_after": v1,
           "bench": {k: bench.get(k) for k in ("value", "ms_per_step", "roofline", "config", "error")},
           "series": [[round(t - t0, 3), p, c] for t, p, c in pw]}
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(out, fh)
    print(json.dumps({"tag": tag, "summary": summary, "bench_value": bench.get("value"),
                      "avg_launch_ms": (bench.get("roofline") or {}).get("avg_launch_ms"),
                      "violations_after": v1 if not isinstance(v1, dict) else {k: v for k, v in v1.items() if "ppt" in k or "thm" in k or "active" in k}}))


if __name__ == "__main__":
    main()
