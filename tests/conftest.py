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


def compare_result(out: dict, ref: dict, tol: dict, label: str = ""):
    """Shared parity assertion.  tol: focal (rel), dist/gravity (abs), cost/unc/cov (rel to max)."""
    def rel(a, b):
        return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)
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
