"""CPU: the oracle (oracle/lm_oracle.c) against golden vectors produced by the REFERENCE itself
(tests/golden/make_golden.py ran /root/reference's LMOptimizer).  This pins the oracle; the -m gpu
tests then compare the HIP path with the same goldens and with the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, compare_result, conf_for, data_for, golden_cases, golden_outputs

# float32 oracle vs float32 reference: summation order is the only difference
TIGHT = {"focal": 5e-6, "dist": 5e-6, "gravity": 5e-6, "cost": 5e-5, "cov": 1e-4, "unc": 5e-4}
# simple_divisional: the reference's own formulas cancel catastrophically (flagged unstable at camera.py:913)
LOOSE = {"focal": 2e-3, "dist": 3e-3, "gravity": 5e-4, "cost": 5e-4, "cov": 1e-2, "unc": 1e-2}


@pytest.mark.parametrize("setname,variant", golden_cases())
def test_oracle_matches_reference_small(oracle, setname, variant):
    ref = golden_outputs(setname, variant)
    out = oracle.solve(data_for(setname, variant), conf_for(setname, variant), precision="f32")
    div = "divisional" in setname
    compare_result(out, ref, LOOSE if div else TIGHT, f"{setname}/{variant}")
    if not div:
        assert np.array_equal(out["stop_at"], ref["stop_at"])


@pytest.mark.parametrize("setname,variant", [c for c in golden_cases(("pinhole", "simple_radial")) if c[1] in ("bench", "default")])
def test_oracle_f64_agrees_with_reference(oracle, setname, variant):
    """The float64 build is the 'exact' answer: the float32 reference must sit within fp32 noise of it."""
    ref = golden_outputs(setname, variant)
    out = oracle.solve(data_for(setname, variant), conf_for(setname, variant), precision="f64")
    compare_result(out, ref, {"focal": 2e-5, "dist": 2e-5, "gravity": 2e-5, "cost": 1e-5, "cov": 1e-3, "unc": 1e-3},
                   f"{setname}/{variant}/f64")


def test_oracle_training_mode_has_no_uncertainty(oracle):
    small = np.load(os.path.join(GOLDEN, "golden_small.npz"))
    out = oracle.solve(data_for("pinhole", "bench"), conf_for("pinhole", "bench"), training=True)
    assert "covariance" not in out
    assert not any("uncertainty" in k or k == "covariance" for k in small["pinhole/training/keys"])
    assert np.abs(out["camera"][:, 2:4] / small["pinhole/training/camera"][:, 2:4] - 1).max() < 5e-6


@pytest.mark.parametrize("setname", ["pinhole", "simple_radial", "shared_pinhole", "shared_simple_radial"])
def test_oracle_step_trace(oracle, setname):
    """Per-step Grad / Hess / delta / lambda / costs / parameters against the reference's own loop body."""
    tr = np.load(os.path.join(GOLDEN, "golden_trace.npz"))
    ref = {k.split("/", 1)[1]: tr[k] for k in tr.files if k.startswith(setname + "/")}
    steps = ref["camera"].shape[0]
    conf = {**conf_for(setname, "bench"), "num_steps": steps}
    out = oracle.solve(data_for(setname, "bench"), conf, trace=True)["trace"]
    B = ref["camera"].shape[1]
    P = 3 if "pinhole" in setname else 4
    shared = setname.startswith("shared_")
    for i in range(steps):
        scale = np.abs(ref["H"][i]).max()
        if not shared:
            assert np.abs(out["H"][i][:, :P, :P] - ref["H"][i]).max() / scale < 3e-5, i
            gs = np.sqrt(np.abs(np.einsum("bii->bi", ref["H"][i])))      # gradient scale per parameter
            assert (np.abs(out["G"][i][:, :P] - ref["G"][i]) / (gs + 1e-12)).max() < 1e-3, i
            assert np.abs(out["delta"][i][:, :P] - ref["delta"][i]).max() < 1e-4, i
            assert np.allclose(out["lambda"][i], ref["lambda"][i], rtol=1e-6), i
        else:
            ni = P - 2                                                     # arrow-head layout (:350-383)
            Hd = ref["H"][i][0]
            for b in range(B):
                assert np.abs(out["H"][i][b, :2, :2] - Hd[2 * b:2 * b + 2, 2 * b:2 * b + 2]).max() / scale < 3e-5
                assert np.abs(out["H"][i][b, :2, 2:P] - Hd[2 * b:2 * b + 2, 2 * B:]).max() / scale < 3e-5
            assert np.abs(out["H"][i][:, 2:P, 2:P].sum(0) - Hd[2 * B:, 2 * B:]).max() / scale < 3e-5
            d = ref["delta"][i][0]
            assert np.abs(out["delta"][i][:, :2].reshape(-1) - d[:2 * B]).max() < 2e-5
            assert np.abs(out["delta"][i][0, 2:2 + ni] - d[2 * B:]).max() < 2e-5
        assert np.allclose(out["cost_up"][i], ref["cost_up"][i], rtol=3e-5)
        assert np.allclose(out["cost_lat"][i], ref["cost_lat"][i], rtol=3e-5)
        assert np.abs(out["cam"][i][:, :2] / ref["camera"][i][:, 2:4] - 1).max() < 2e-5
        assert np.abs(out["gravity"][i] - ref["gravity"][i]).max() < 2e-5


@pytest.mark.parametrize("model", ["pinhole", "simple_radial", "radial", "simple_divisional"])
@pytest.mark.parametrize("mode", ["loop", "rpf"])
def test_oracle_single_pass_system(oracle, model, mode):
    """costs, J^T W r, J^T W J at fixed (non-converged) parameters, both parametrisations."""
    s = np.load(os.path.join(GOLDEN, "golden_system.npz"))
    inp = np.load(os.path.join(GOLDEN, f"inputs_{model}.npz"))
    data = {k: inp[k] for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    out = oracle.system(data, s[f"{model}/camera"], s[f"{model}/gravity"], {"camera_model": model},
                        as_rpf=(mode == "rpf"), precision="f64")
    tol = 3e-3 if model == "simple_divisional" else 3e-5   # divisional: fp32 cancellation in the reference itself
    Hr, Gr = s[f"{model}/{mode}/H"], s[f"{model}/{mode}/G"]
    dscale = np.sqrt(np.abs(np.einsum("bii->bi", Hr)))
    assert (np.abs(out["H"] - Hr) / (dscale[:, :, None] * dscale[:, None, :])).max() < tol
    # gradient: compare relative to the per-parameter scale sqrt(H_kk * total cost)
    cost = (s[f"{model}/{mode}/cost_up"] + s[f"{model}/{mode}/cost_lat"]) * data["latitude_field"][0].size
    assert (np.abs(out["G"] - Gr) / (dscale * np.sqrt(cost)[:, None])).max() < tol
    assert np.allclose(out["cost_up"], s[f"{model}/{mode}/cost_up"], rtol=1e-5)
    assert np.allclose(out["cost_lat"], s[f"{model}/{mode}/cost_lat"], rtol=1e-5)


@pytest.mark.parametrize("model", ["pinhole", "simple_radial"])
def test_oracle_matches_reference_full_size(oracle, model):
    """BASELINE configs[1]/[3] shape (640x480, 20 iters), 4 images: inputs are regenerated from the seed."""
    from oracle import synth
    full = np.load(os.path.join(GOLDEN, "golden_full.npz"))
    data, cams, gravs = synth.make_fields(1234, range(4), model, 480, 640)
    chk = np.array([np.float64(np.asarray(v, np.float64).sum()) for _, v in sorted(data.items())])
    assert np.allclose(chk, full[f"{model}/input_checksum"], rtol=1e-9, atol=1e-3), "regenerated inputs drifted"
    out = oracle.solve(data, {"camera_model": model, "num_steps": 20, "early_stop": False}, precision="f32")
    ref = {k.split("/", 1)[1]: full[k] for k in full.files if k.startswith(model + "/")}
    compare_result(out, ref, TIGHT, f"full/{model}")
    assert np.array_equal(out["stop_at"], ref["stop_at"])
    # and the answer is the ground truth up to the noise level
    assert np.abs(out["camera"][:, 3] / cams[:, 3] - 1).max() < 5e-3


@pytest.mark.parametrize("variant", ["default", "bench"])
def test_oracle_matches_reference_cnn_fields(oracle, variant):
    """BASELINE configs[0] restated: fields of the (seeded, randomly initialised) reference CNN on
    assets/pinhole-church.jpg.  Ill-conditioned (focal uncertainty ~25 %), hence the looser focal gate."""
    g = np.load(os.path.join(GOLDEN, "golden_cnn.npz"))
    data = {k: g[k] for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    conf = {} if variant == "default" else {"num_steps": 20, "early_stop": False}
    out = oracle.solve(data, conf, precision="f32")
    ref = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(variant + "/")}
    compare_result(out, ref, {"focal": 2e-3, "dist": 1e-6, "gravity": 1e-4, "cost": 1e-4, "cov": 5e-2, "unc": 2e-2},
                   f"cnn/{variant}")


def test_oracle_render_matches_reference_fields(oracle):
    """The generator's renderer restates get_perspective_field (perspective_fields.py:278): the
    noise-free field of the GT camera must be a zero-cost fixed point of the reference-pinned solver."""
    from oracle import synth
    data, cams, gravs = synth.make_fields(5, range(2), "simple_radial", 48, 64, noise=0.0, confidences=False)
    s = oracle.system(data, cams, gravs, {"camera_model": "simple_radial"}, precision="f64")
    assert s["cost_up"].max() < 1e-12 and s["cost_lat"].max() < 1e-12


@pytest.mark.parametrize("model", ["pinhole", "simple_radial", "radial", "simple_divisional"])
@pytest.mark.parametrize("tag", ["loop", "rpf"])
def test_oracle_jacobian_fields_match_reference(oracle, model, tag):
    """Per-pixel Jacobians against the reference's J_perspective_field (perspective_fields.py:323-365), both
    parametrisations (spherical + log focal of the loop, roll/pitch + focal of the uncertainty pass)."""
    g = np.load(os.path.join(GOLDEN, "golden_jac.npz"))
    sph = tag == "loop"
    for prec, tol in (("f64", 1e-6), ("f32", 1e-3 if model == "simple_divisional" else 5e-6)):
        J_up, J_lat = oracle.jacobian_fields(model, 12, 16, g[f"{model}/camera"], g[f"{model}/gravity"], sph, sph,
                                             precision=prec)
        assert J_up.shape == g[f"{model}/{tag}/J_up"].shape and J_lat.shape == g[f"{model}/{tag}/J_lat"].shape
        assert np.abs(J_up - g[f"{model}/{tag}/J_up"]).max() < tol, (model, tag, prec)
        assert np.abs(J_lat - g[f"{model}/{tag}/J_lat"]).max() < tol, (model, tag, prec)


@pytest.mark.parametrize("model", ["pinhole", "simple_radial", "radial", "simple_divisional"])
def test_oracle_residuals_and_costs_match_reference(oracle, model):
    """Per-pixel calculate_residuals / calculate_costs (lm_optimizer.py:248-315) on noisy fields at a perturbed
    estimate (78 % of the pixels beyond the Huber threshold)."""
    g = np.load(os.path.join(GOLDEN, "golden_jac.npz"))
    data = {k: g[f"{model}/res/{k}"] for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    res = oracle.residual_fields(model, data, g[f"{model}/res/camera"], g[f"{model}/res/gravity"], precision="f32")
    for k in ("up_residual", "latitude_residual"):
        assert res[k].shape == g[f"{model}/res/{k}"].shape
        assert np.abs(res[k] - g[f"{model}/res/{k}"]).max() < 1e-6, (model, k)
    for key, conf, ck, wk in (("up_residual", "up_confidence", "up_cost", "up_weights"),
                              ("latitude_residual", "latitude_confidence", "latitude_cost", "latitude_weights")):
        cost, weight = oracle.huber_costs(g[f"{model}/res/{key}"], 1e-2, data[conf], precision="f32")
        assert np.abs(cost - g[f"{model}/res/{ck}"]).max() < 1e-6 * np.abs(g[f"{model}/res/{ck}"]).max()
        assert np.abs(weight - g[f"{model}/res/{wk}"]).max() < 1e-6


@pytest.mark.parametrize("model", ["pinhole", "simple_radial"])
@pytest.mark.parametrize("knob", ["heuristic", "squared_loss"])
def test_oracle_matches_reference_siclib_knobs(oracle, model, knob):
    """Training-time knobs (siclib): heuristic initialisation (utils.py:27-82) and squared loss (losses.py:26);
    goldens from tests/golden/make_golden_extra.py."""
    g = np.load(os.path.join(GOLDEN, "golden_extra.npz"))
    ref = {k.split("/", 2)[2]: g[k] for k in g.files if k.startswith(f"{model}/{knob}/")}
    conf = {"camera_model": model, "num_steps": 20, "early_stop": False}
    conf |= {"init_conf": {"name": "heuristic"}} if knob == "heuristic" else {"loss_fn": "squared_loss"}
    out = oracle.solve(data_for(model, "bench"), conf, precision="f32")
    compare_result(out, ref, TIGHT, f"{model}/{knob}")
    if knob == "heuristic":     # and the initial estimate itself (0 LM steps)
        init = oracle.solve(data_for(model, "bench"), {**conf, "num_steps": 0}, precision="f32")
        assert np.allclose(init["camera"], ref["init_camera"], rtol=2e-6)
        assert np.allclose(init["gravity"], ref["init_gravity"], atol=2e-6)


def test_oracle_matches_reference_shared_radial(oracle):
    """Shared intrinsics with three shared parameters (f, k1, k2): the dense arrow-head path of the reference."""
    g = np.load(os.path.join(GOLDEN, "golden_extra.npz"))
    ref = {k.split("/", 2)[2]: g[k] for k in g.files if k.startswith("radial/shared/")}
    conf = {"camera_model": "radial", "shared_intrinsics": True, "num_steps": 20, "early_stop": False}
    out = oracle.solve(data_for("radial", "bench"), conf, precision="f32")
    compare_result(out, ref, {**TIGHT, "cost": 2e-4, "cov": 1e-3, "unc": 2e-3}, "radial/shared")


@pytest.mark.parametrize("model", ["pinhole", "simple_radial"])
def test_oracle_matches_reference_shared16_at_shape(oracle, model):
    """BASELINE configs[4] at its stated shape: ONE shared-intrinsics group of 16 frames at 640x480, 20 iterations,
    against the reference's own output (tests/golden/make_golden_shared16.py; lm_optimizer.py:350-383, 597-603)."""
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, "golden_shared16.npz"))
    data, cams, gravs = synth.make_shared_group(1234, 0, model, 480, 640, frames=16)
    chk = np.array([np.float64(np.asarray(v, np.float64).sum()) for _, v in sorted(data.items())])
    assert np.allclose(chk, g[f"{model}/g0/input_checksum"], rtol=1e-9, atol=1e-3), "regenerated inputs drifted"
    conf = {"camera_model": model, "shared_intrinsics": True, "num_steps": 20, "early_stop": False}
    out = oracle.solve(data, conf, precision="f32")
    ref = {k.split("/", 2)[2]: g[k] for k in g.files if k.startswith(f"{model}/g0/")}
    compare_result(out, ref, TIGHT, f"shared16/{model}")
    assert np.array_equal(out["stop_at"], ref["stop_at"])
    # one camera for the whole group, and it is the ground truth up to the noise level
    assert np.abs(out["camera"][:, 2:4] - out["camera"][0, 2:4]).max() == 0
    assert np.abs(out["camera"][0, 3] / cams[0, 3] - 1) < 2e-3


def test_oracle_matches_reference_divisional_fuzz_draws(oracle):
    """simple_divisional on the fuzz generator's draws (seed 2024: all 40 cases; seed 11: the first 80), oracle vs the
    REFERENCE's own float32 result (tests/golden/make_golden_div.py).  The reference's k-column cancels in float32 for
    small |k| (camera.py:913), so the gate is the reference's own reproducibility: where two 1-ulp perturbations of
    its input move it by `spread`, the oracle may sit 10 x spread (+ the usual gate) away, and where spread > 1e-3
    (27 % of the draws) parity is undefined and only finiteness is asked."""
    from conftest import fuzz_draws, result_spread
    g = np.load(os.path.join(GOLDEN, "golden_div_fuzz.npz"))
    checked = undefined = 0
    for seed, cases in ((2024, 40), (11, 80)):
        for case, model, (H, W), B, data, conf, cams, gravs in fuzz_draws(seed, cases, 4):
            if f"{seed}/{case}/camera" not in g.files:
                continue
            assert model == "simple_divisional"
            out = oracle.solve(data, conf, precision="f32")
            assert all(np.isfinite(out[k]).all() for k in ("camera", "gravity", "final_cost"))
            spread = g[f"{seed}/{case}/spread"]
            if spread.max() > 1e-3:
                undefined += 1
                continue
            ref = {k: g[f"{seed}/{case}/{k}"] for k in ("camera", "gravity", "final_cost", "initial_cost")}
            d = result_spread(out, ref)
            tol = np.array([2e-3, 2e-3, 5e-3, 2e-3]) + 10.0 * spread
            assert (d < tol).all(), (seed, case, (H, W), B, conf, d, tol)
            checked += 1
    assert checked >= 10 and undefined >= 3, (checked, undefined)


@pytest.mark.parametrize("model,idx", [("radial", (0, 1)), ("simple_divisional", (2, 5))])
def test_oracle_matches_reference_full_size_other_models(oracle, model, idx):
    """The two non-BASELINE camera models at the BASELINE image size (640x480, 20 iterations, two images each) against
    the reference's own result (tests/golden/make_golden_full_rd.py); gate = 1e-4 + 10 x the reference's own 1-ulp
    input sensitivity on these images."""
    from conftest import result_spread
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, "golden_full_rd.npz"))
    data, cams, gravs = synth.make_fields(1234, idx, model, 480, 640)
    chk = np.array([np.float64(np.asarray(v, np.float64).sum()) for _, v in sorted(data.items())])
    assert np.allclose(chk, g[f"{model}/input_checksum"], rtol=1e-9, atol=1e-3), "regenerated inputs drifted"
    out = oracle.solve(data, {"camera_model": model, "num_steps": 20, "early_stop": False}, precision="f32")
    ref = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(model + "/")}
    d = result_spread(out, ref)
    assert (d < 1e-4 + 10.0 * ref["spread"]).all(), (model, d, ref["spread"])
    assert np.abs(out["camera"][:, 3] / cams[:, 3] - 1).max() < 5e-3          # and it is the ground truth
