import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
BENCH = {"num_steps": 20, "early_stop": False}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # host-side thread pools sized to what the container may use (a GPU box shows 256 CPUs and grants 16)
    try:
        import torch
        from oracle.lm_oracle import effective_cpus
        torch.set_num_threads(effective_cpus())
    except Exception:
        pass


def pytest_sessionstart(session):
    """Build the native pieces if a checkout has none yet (hipcc cross-compiles without a GPU)."""
    lib = os.path.join(ROOT, "geocalib_amd", "lib", "libgeocalib_hip.so")
    orc = os.path.join(ROOT, "oracle", "_build", "liblm_oracle_f32.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must never run (and silently pass) without a device."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import lm_oracle
    lm_oracle.build()
    return lm_oracle


def conf_for(setname: str, variant: str) -> dict:
    """The LMOptimizer conf each golden variant was generated with (tests/golden/make_golden.py)."""
    model = setname.replace("shared_", "")
    c = {"camera_model": model, **BENCH}
    if setname.startswith("shared_"):
        c["shared_intrinsics"] = True
    if variant == "default":
        c = {"camera_model": model}
    elif variant == "euclid":
        c["use_spherical_manifold"] = False
    elif variant == "linfocal":
        c["use_log_focal"] = False
    elif variant == "fixlambda":
        c["fix_lambda"] = True
    elif variant == "loss_scale":
        c.update(up_loss_fn_scale=5e-2, lat_loss_fn_scale=2e-2)
    return c


def data_for(setname: str, variant: str) -> dict:
    inp = np.load(os.path.join(GOLDEN, f"inputs_{setname}.npz"))
    d = {k: inp[k] for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    if variant == "noconf":
        d = {k: v for k, v in d.items() if "confidence" not in k}
    elif variant == "lat_only":
        d = {k: d[k] for k in ("latitude_field", "latitude_confidence")}
    elif variant == "scales":
        d["scales"] = np.array([0.5, 0.6], np.float32)
    elif variant == "prior_focal":
        d["prior_focal"] = inp["gt_camera"][:, 3].copy()
    elif variant == "prior_gravity":
        d["prior_gravity"] = inp["gt_gravity"].copy()
    return d


def golden_cases(models=None):
    small = np.load(os.path.join(GOLDEN, "golden_small.npz"))
    cases = sorted({tuple(k.split("/")[:2]) for k in small.files})
    cases = [c for c in cases if c[1] != "training"]
    if models is not None:
        cases = [c for c in cases if c[0].replace("shared_", "") in models]
    return cases


def golden_outputs(setname: str, variant: str) -> dict:
    small = np.load(os.path.join(GOLDEN, "golden_small.npz"))
    pre = f"{setname}/{variant}/"
    return {k[len(pre):]: small[k] for k in small.files if k.startswith(pre)}


MEASURED = {}      # label -> what compare_result / the fuzz actually measured (written to gpurun_out/ at session end)


def measure_result(out: dict, ref: dict) -> dict:
    """The distances compare_result gates, as numbers (scripts/parity_report.py, profiles/archive/r03_parity.json)."""
    def rel(a, b):
        return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
    cam, rcam = np.asarray(out["camera"]), ref["camera"]
    m = {"focal": float(np.abs(cam[:, 2:4] / rcam[:, 2:4] - 1).max()), "dist": float(np.abs(cam[:, 6:] - rcam[:, 6:]).max()),
         "gravity": float(np.abs(np.asarray(out["gravity"]) - ref["gravity"]).max()),
         "cost": max(rel(out[k], ref[k]) for k in ("initial_cost", "final_cost", "initial_latitude_cost", "final_latitude_cost"))}
    if "covariance" in ref and "covariance" in out:
        m["cov"] = rel(out["covariance"], ref["covariance"])
        m["unc"] = max([rel(out[k], ref[k]) for k in ("roll_uncertainty", "pitch_uncertainty", "gravity_uncertainty",
                                                      "focal_uncertainty", "vfov_uncertainty") if np.abs(ref[k]).max() > 0] or [0.0])
    return m


def pytest_sessionfinish(session, exitstatus):
    if MEASURED and os.path.isdir(os.path.join(ROOT, "gpurun_out")) or (MEASURED and os.environ.get("GCLM_PARITY_LOG")):
        import json
        path = os.environ.get("GCLM_PARITY_LOG") or os.path.join(ROOT, "gpurun_out", "parity_measured.json")
        try:
            old = json.load(open(path)) if os.path.exists(path) else {}
            old.update(MEASURED)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            json.dump(old, open(path, "w"), indent=0, sort_keys=True)
        except Exception:
            pass


def compare_result(out: dict, ref: dict, tol: dict, label: str = ""):
    """Shared parity assertion.  tol: focal (rel), dist/gravity (abs), cost/unc/cov (rel to max)."""
    def rel(a, b):
        return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)
    if label:
        MEASURED[label] = {**measure_result(out, ref), "tol": {k: float(v) for k, v in tol.items()}}
    cam, rcam = np.asarray(out["camera"]), ref["camera"]
    assert np.array_equal(cam[:, [0, 1, 4, 5]], rcam[:, [0, 1, 4, 5]]), label
    f = np.abs(cam[:, 2:4] / rcam[:, 2:4] - 1).max()
    assert f < tol["focal"], f"{label}: focal rel err {f:.2e}"
    k = np.abs(cam[:, 6:] - rcam[:, 6:]).max()
    assert k < tol["dist"], f"{label}: distortion abs err {k:.2e}"
    g = np.abs(np.asarray(out["gravity"]) - ref["gravity"]).max()
    assert g < tol["gravity"], f"{label}: gravity abs err {g:.2e}"
    for key in ("initial_cost", "final_cost", "initial_latitude_cost", "final_latitude_cost"):
        assert rel(out[key], ref[key]) < tol["cost"], f"{label}: {key} {rel(out[key], ref[key]):.2e}"
    if "covariance" in ref and "covariance" in out:
        assert rel(out["covariance"], ref["covariance"]) < tol["cov"], f"{label}: covariance {rel(out['covariance'], ref['covariance']):.2e}"
        for key in ("roll_uncertainty", "pitch_uncertainty", "gravity_uncertainty", "focal_uncertainty", "vfov_uncertainty"):
            if np.abs(ref[key]).max() > 0:
                assert rel(out[key], ref[key]) < tol["unc"], f"{label}: {key} {rel(out[key], ref[key]):.2e}"


ALL_MODELS = ("pinhole", "simple_radial", "radial", "simple_divisional")


def fuzz_draws(seed: int, n_cases: int, n_models: int = 4):
    """The seeded random configurations of the fuzz tests: yields (case, model, (H, W), B, data, conf, cams, gravs).
    Shared by tests/test_gpu_parity.py::test_randomised_configurations_against_oracle, scripts/fuzz_case.py and
    tests/golden/make_golden_div.py (which runs the REFERENCE on the simple_divisional draws), so that a
    (seed, case) pair names the same inputs everywhere."""
    from oracle import synth
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        model = ALL_MODELS[rng.integers(0, n_models)]
        H, W = int(rng.integers(24, 90)), int(rng.integers(24, 120))
        if rng.random() < 0.1:                                      # many chunk records per image: striped reduction
            H, W = int(rng.integers(200, 300)), int(rng.integers(260, 340))
        if rng.random() < 0.5:
            W = W // 4 * 4
        B = int(rng.integers(1, 6))
        data, cams, gravs = synth.make_fields(int(rng.integers(0, 1 << 30)), range(B), model, H, W,
                                              noise=float(rng.choice([0.0, 0.01, 0.03])))
        conf = {"camera_model": model, "num_steps": int(rng.integers(1, 25)), "early_stop": bool(rng.random() < 0.5),
                "use_spherical_manifold": bool(rng.random() < 0.7), "use_log_focal": bool(rng.random() < 0.7),
                "fix_lambda": bool(rng.random() < 0.2), "lambda_": float(rng.choice([0.1, 0.01, 1.0])),
                "up_loss_fn_scale": float(rng.choice([1e-2, 5e-2])), "lat_loss_fn_scale": float(rng.choice([1e-2, 3e-2]))}
        if rng.random() < 0.15:
            conf["loss_fn"] = "squared_loss"
        if rng.random() < 0.15:
            conf["init_conf"] = {"name": "heuristic"}
        mode = rng.random()
        if mode < 0.15:
            data = {k: v for k, v in data.items() if "confidence" not in k}
        elif mode < 0.25:
            data = {k: data[k] for k in ("latitude_field", "latitude_confidence")}
            conf.pop("init_conf", None)
        elif mode < 0.35:
            data["prior_gravity"] = gravs
        elif mode < 0.45 and model == "pinhole":
            data["prior_focal"] = cams[:, 3].copy()
        if rng.random() < 0.2:
            data["scales"] = np.array([rng.uniform(0.4, 1.0), rng.uniform(0.4, 1.0)], np.float32)
        shared = rng.random() < 0.15 and model != "radial" and "prior_gravity" not in data and "prior_focal" not in data
        if shared:
            conf |= {"shared_intrinsics": True, "early_stop": False}
        yield case, model, (H, W), B, data, conf, cams, gravs


PER_PIXEL = ("up_field", "latitude_field", "up_confidence", "latitude_confidence")


def perturbed(data: dict, rng) -> dict:
    """Every per-pixel input scaled by (1 +- 2^-23), sign drawn per pixel: one unit in the last place.  How far a
    float32 solver moves under this is the sharpest any OTHER float32 evaluation of the same algorithm can be held to
    (the LM loop is discontinuous in rounding noise: the x10 / x0.1 damping rule and the batch-global stop compare costs
    that agree to the last bit or two, lm_optimizer.py:95-106, 90-92)."""
    out = dict(data)
    for k in PER_PIXEL:
        if k in data:
            sign = rng.integers(0, 2, data[k].shape).astype(np.float32) * 2 - 1
            out[k] = (data[k] * (np.float32(1) + sign * np.float32(2.0 ** -23))).astype(np.float32)
    return out


def result_spread(a: dict, b: dict) -> np.ndarray:
    """[focal rel, gravity abs, distortion abs, final-cost rel] distance of two result dicts (fuzz gates)."""
    rel_f = np.abs(a["camera"][:, 2:4] / b["camera"][:, 2:4] - 1).max()
    dg = np.abs(a["gravity"] - b["gravity"]).max()
    dk = np.abs(a["camera"][:, 6:] - b["camera"][:, 6:]).max()
    # noise-free draws end at costs ~1e-9 that are pure rounding: measure against the problem's own scale
    floor = max(1e-7, 1e-3 * np.abs(b["initial_cost"]).max())
    dc = np.abs(a["final_cost"] - b["final_cost"]).max() / max(np.abs(b["final_cost"]).max(), floor)
    return np.array([rel_f, dg, dk, dc], np.float64)
