"""CPU: the host-side logic of bench.py that does not need a device.

(1) without a HIP device `bench.py` (and `bench.py --gpus N`, which would start its own ranks) answers with ONE JSON line
    carrying `error` and a non-zero exit code -- never a traceback, never a CPU fallback of the product path;
(2) `cpu_baseline()` has two branches: the reference's own PyTorch path timed on THIS box when the checkout is there
    ($GEOCALIB_REFERENCE / /root/reference: the build container), otherwise the C port alone with
    `reference_on_this_box: false`.  Both keep the keys the driver parses (value, unit, cores, kind)."""
import json
import os
import subprocess
import sys
import types

import pytest
import torch

from conftest import ROOT


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


@pytest.mark.parametrize("extra", [[], ["--gpus", "2"], ["--gpus", "8", "--shared-group", "16"]])
def test_without_a_device_bench_prints_one_error_line(extra):
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=env)
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert res.returncode == 2 and len(lines) == 1, (res.returncode, res.stdout, res.stderr[-1000:])
    out = json.loads(lines[0])
    assert out["value"] is None and "no HIP device" in out["error"]
    assert out["n_gpus"] == (int(extra[1]) if extra else 1) and out["unit"] == "images/sec"
    assert "Traceback" not in res.stderr


def _sample(n, H=48, W=64, model="pinhole"):
    from oracle import synth
    data, _, _ = synth.make_fields(11, range(n), model, H, W)
    return data


def _args(model="pinhole", H=48, W=64, steps=5):
    return types.SimpleNamespace(camera_model=model, lm_steps=steps, height=H, width=W)


def test_cpu_baseline_without_the_reference_is_the_port(monkeypatch):
    bench = _bench()
    from oracle import ref_import
    monkeypatch.setattr(ref_import, "REFERENCE_ROOT", "/nonexistent/reference")
    assert not ref_import.available()
    out = bench.cpu_baseline(_args(), 4, _sample(4))
    assert out["kind"] == "port" and out["reference_on_this_box"] is False
    assert out["value"] > 0 and out["unit"] == "images/sec" and 1 <= out["cores"] <= 4
    assert "port" not in out and "oracle/lm_oracle.c" in out["sample"]
    # the committed build-container timing of the reference is attached only for the shape it was measured at
    assert "reference_torch" not in out
    full = bench.reference_torch(types.SimpleNamespace(camera_model="pinhole", lm_steps=20, height=480, width=640))
    assert full is None or (full["kind"] == "reference" and "build container" in full["measured_on"])


def test_cpu_baseline_times_the_reference_where_it_exists():
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("no reference checkout on this box")
    bench = _bench()
    data = _sample(3)
    out = bench.cpu_baseline(_args(), 3, data)
    assert out["kind"] == "reference" and out["measured_on"] == "this box" and out["reference_on_this_box"] is True
    assert out["value"] > 0 and out["unit"] == "images/sec" and out["cores"] >= 1
    assert "lm_optimizer.py:141" in out["code"] and "the first 3 images" in out["sample"]
    assert out["port"]["kind"] == "port" and out["port"]["value"] > 0
    # the thread count of the process is restored
    assert torch.get_num_threads() >= 1


def test_cpu_baseline_survives_a_broken_reference_checkout(monkeypatch, tmp_path):
    """A directory that looks like a checkout but does not import: the baseline falls back to the port and says why."""
    (tmp_path / "geocalib").mkdir()
    bench = _bench()
    from oracle import ref_import
    real_root = ref_import.REFERENCE_ROOT
    monkeypatch.setattr(ref_import, "REFERENCE_ROOT", str(tmp_path))
    monkeypatch.setattr(sys, "path", [q for q in sys.path if q != real_root])     # (an earlier test may have imported the real one)
    assert ref_import.available()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "geocalib" or k.startswith("geocalib.")}
    try:
        out = bench.cpu_baseline(_args(), 2, _sample(2))
    finally:
        for k in [k for k in sys.modules if k == "geocalib" or k.startswith("geocalib.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    assert out["kind"] == "port" and out["reference_on_this_box"] is False and "reference_error" in out


def test_vs_oracle_record_and_the_kept_oracle_result(monkeypatch):
    """`check.vs_oracle`: cpu_baseline hands the oracle's solve of the sample back (`keep`), vs_oracle turns
    it and the HIP rows into the record of the bench line -- distances next to the 1e-4 gate; a NaN is never inside a gate;
    shared-intrinsics samples are solved group by group like the reference does."""
    import numpy as np
    bench = _bench()
    from oracle import ref_import
    monkeypatch.setattr(ref_import, "REFERENCE_ROOT", "/nonexistent/reference")
    data = _sample(6)
    kept = {}
    out = bench.cpu_baseline(_args(), 6, data, keep=kept)
    assert out["kind"] == "port" and set(kept) == {"camera", "gravity", "final_cost"}
    assert kept["camera"].shape == (6, 8) and kept["gravity"].shape == (6, 3) and kept["final_cost"].shape == (6,)
    same = bench.vs_oracle({k: v.copy() for k, v in kept.items()}, kept, "itself")
    assert same["within_gate"] is True and same["images"] == 6 and same["gate"] == 1e-4
    assert same["max_focal_rel"] == same["max_gravity_abs"] == same["max_final_cost_rel"] == 0.0
    off = {k: v.copy() for k, v in kept.items()}
    off["camera"][2, 3] *= 1 + 3e-4
    rec = bench.vs_oracle(off, kept, "a focal 3e-4 off")
    assert rec["within_gate"] is False and 2.9e-4 < rec["max_focal_rel"] < 3.1e-4 and "a focal 3e-4 off" in rec["against"]
    nan = {k: v.copy() for k, v in kept.items()}
    nan["gravity"][0, 0] = np.nan
    assert bench.vs_oracle(nan, kept, "nan")["within_gate"] is False
    # more HIP rows than oracle rows: only the sample is compared
    more = {k: np.concatenate([v, v]) for k, v in kept.items()}
    assert bench.vs_oracle(more, kept, "prefix")["images"] == 6
    # shared intrinsics: one oracle call per group of `group` frames (lm_optimizer.py:350-383), concatenated in order
    args = _args()
    args.shared_group = 3
    kept_g = {}
    out_g = bench.cpu_baseline(args, 6, data, keep=kept_g)
    assert out_g["unit"] == "frames/sec" and "2 shared-intrinsics groups of 3" in out_g["sample"]
    assert kept_g["camera"].shape == (6, 8)
    for lo in (0, 3):
        assert np.ptp(kept_g["camera"][lo:lo + 3, 3]) == 0            # one focal per group ...
    assert kept_g["camera"][0, 3] != kept_g["camera"][3, 3]          # ... and not the same for both
    direct = bench.oracle_solve("pinhole", 5, {k: v[:3] for k, v in data.items()}, 3, 2)
    assert np.array_equal(direct["camera"], kept_g["camera"][:3])
