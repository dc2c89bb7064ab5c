"""-m gpu: parity of the HIP path (through the C ABI of include/gclm.h, driven by geocalib_amd.LMOptimizer)
against (1) golden vectors produced by the REFERENCE itself and (2) the CPU oracle on the same seeded
inputs, plus size-independent properties at the full BASELINE size.

Tolerances (float32 path; the reference's own run-to-run / fp32-vs-fp64 spread is ~1e-6, see
tests/test_oracle.py): focal < 1e-4 relative, gravity < 1e-4 absolute (BASELINE.json north_star),
costs < 1e-4 relative, covariance / uncertainties < 1e-3 relative to their largest entry."""
import ctypes as C
import os
import time

import numpy as np
import pytest
import torch

from conftest import GOLDEN, compare_result, conf_for, data_for, golden_cases, golden_outputs

pytestmark = pytest.mark.gpu

TOL = {"focal": 1e-4, "dist": 1e-4, "gravity": 1e-4, "cost": 1e-4, "cov": 1e-3, "unc": 1e-3}
HIP_MODELS = ("pinhole", "simple_radial")          # the two BASELINE models: full test matrix
ALL_MODELS = ("pinhole", "simple_radial", "radial", "simple_divisional")
# simple_divisional is held to the same gates.  Its reference formulas cancel in float32 (the k-column of the Jacobian,
# flagged unstable at camera.py:913), so every simple_divisional comparison ADDS what its yardstick itself moves on
# that input -- the reference under 1-ulp input perturbations (committed with the goldens), the oracle between its
# float32 and float64 builds -- and nothing else: there is no blanket tolerance for the model any more.


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from geocalib_amd import _lib
    _lib.load()                       # fail loudly if the extension is not built
    return torch.device("cuda:0")


def to_dev(data, dev):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in data.items()}


def to_np(out):
    res = {}
    for k, v in out.items():
        res[k] = (v._data if hasattr(v, "_data") else v).detach().cpu().numpy()
    return res


def run(conf, data, dev, training=False, row_pairs=None):
    from geocalib_amd import LMOptimizer
    opt = LMOptimizer(dict(conf))
    opt.row_pairs = row_pairs            # None: the library's choice; True: the row-pair walk wherever the sweep has it
    opt = opt.train() if training else opt.eval()
    out = opt(to_dev(data, dev))
    torch.cuda.synchronize()
    return to_np(out)


# ------------------------------------------------------------------ against the reference's goldens

@pytest.mark.parametrize("row_pairs", [None, True])
@pytest.mark.parametrize("setname,variant", golden_cases(ALL_MODELS))
def test_hip_matches_reference_small(dev, setname, variant, row_pairs):
    """`row_pairs`: the same goldens, same gates, with the sweep walking row pairs (gclm_set_row_pairs(h, 1): radial /
    simple_divisional sets with both confidences; the library's own choice for inputs this small is the one-row walk)."""
    if row_pairs and not any(m in setname for m in ("radial", "divisional")) or (row_pairs and "simple_radial" in setname):
        pytest.skip("the row-pair walk exists for radial / simple_divisional")
    ref = golden_outputs(setname, variant)
    out = run(conf_for(setname, variant), data_for(setname, variant), dev, row_pairs=row_pairs)
    tol = dict(TOL)
    if "divisional" in setname:
        # north_star's 1e-4 plus 3 x what the REFERENCE itself moves on this case under 1-ulp input perturbations
        # (tests/golden/make_golden_div_small.py: its float32 k-column cancels, camera.py:913; measured there: focal up to
        # 3.3e-4 / k 6.9e-4 on `default`, 3e-5 / 4e-4 on `bench`) -- the HIP path's distance is that size, not more
        sp = np.load(os.path.join(GOLDEN, "golden_small_div_spread.npz"))[f"{setname}/{variant}"]
        tol = {k: base + 3.0 * sp[i] for i, (k, base) in enumerate(
            (("focal", 1e-4), ("gravity", 1e-4), ("dist", 1e-4), ("cost", 1e-4), ("cov", 1e-3), ("unc", 1e-3)))}
    if (setname, variant) == ("simple_radial", "prior_focal"):
        tol.update(focal=1e-4, dist=5e-4, gravity=5e-4, cost=5e-4)   # reference quirk 7: never converges
    compare_result(out, ref, tol, f"{setname}/{variant}")
    if "divisional" not in setname:
        assert np.array_equal(out["stop_at"], ref["stop_at"]), (out["stop_at"], ref["stop_at"])
    assert set(k for k in ref if k not in ("camera", "gravity")) <= set(out), set(ref) - set(out)
    assert out["step_failures"].max() == 0


@pytest.mark.parametrize("model", HIP_MODELS)
def test_hip_matches_reference_full_size(dev, model):
    """BASELINE configs[1] / [3] shape: 640x480, 20 iterations (4 images; inputs regenerated from the seed)."""
    from oracle import synth
    full = np.load(os.path.join(GOLDEN, "golden_full.npz"))
    data, cams, gravs = synth.make_fields(1234, range(4), model, 480, 640)
    chk = np.array([np.float64(np.asarray(v, np.float64).sum()) for _, v in sorted(data.items())])
    assert np.allclose(chk, full[f"{model}/input_checksum"], rtol=1e-9, atol=1e-3)
    out = run({"camera_model": model, "num_steps": 20, "early_stop": False}, data, dev)
    ref = {k.split("/", 1)[1]: full[k] for k in full.files if k.startswith(model + "/")}
    compare_result(out, ref, TOL, f"full/{model}")
    assert np.array_equal(out["stop_at"], ref["stop_at"])


@pytest.mark.parametrize("row_pairs", [None, True])
@pytest.mark.parametrize("model,idx", [("radial", (0, 1)), ("simple_divisional", (2, 5))])
def test_hip_matches_reference_full_size_other_models(dev, model, idx, row_pairs):
    """The two non-BASELINE camera models at the BASELINE image size (640x480, 20 iterations) against the REFERENCE's own
    result (tests/golden/make_golden_full_rd.py): north_star's 1e-4, plus 10 x the reference's own 1-ulp input
    sensitivity on these images (2e-4 at most: both are in the regime where the reference reproduces itself)."""
    from conftest import result_spread
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, "golden_full_rd.npz"))
    data, cams, gravs = synth.make_fields(1234, idx, model, 480, 640)
    chk = np.array([np.float64(np.asarray(v, np.float64).sum()) for _, v in sorted(data.items())])
    assert np.allclose(chk, g[f"{model}/input_checksum"], rtol=1e-9, atol=1e-3)
    out = run({"camera_model": model, "num_steps": 20, "early_stop": False}, data, dev, row_pairs=row_pairs)
    ref = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(model + "/")}
    d = result_spread(out, ref)
    assert (d < 1e-4 + 10.0 * ref["spread"]).all(), (model, d, ref["spread"])
    # stop_at: the step at which every cost has stopped moving by 1e-8, a threshold crossing.  Gated by what the
    # REFERENCE's own stop_at does under 1-ulp input perturbations (`stop_at_set`: the unperturbed run + 8 perturbed ones,
    # make_golden_full_rd.py): radial creeps along a flat distortion valley and the reference itself lands on 19 or 20;
    # simple_divisional's crossing is sharp, the reference always reports 9 -- and so must the HIP path.
    allowed = ref["stop_at_set"]
    assert np.isin(out["stop_at"], allowed).all(), (out["stop_at"], ref["stop_at"], allowed)
    print(f"stop_at {model}: HIP {out['stop_at']} reference {ref['stop_at']} reference under 1-ulp perturbations {allowed}")
    assert np.abs(out["covariance"] - ref["covariance"]).max() / np.abs(ref["covariance"]).max() < 1e-3


# BASELINE configs[0] restated on the fields of a RANDOM-INIT CNN: north_star's 1e-4 without exception (round 2 allowed
# 2e-3 on the focal for this ill-conditioned problem; measured: 6e-8, profiles/archive/r03_parity.json).  No distortion: exact.
TOL_CNN = {**TOL, "dist": 1e-6}


@pytest.mark.parametrize("variant", ["default", "bench"])
def test_hip_matches_reference_cnn_fields(dev, variant, oracle):
    """BASELINE configs[0] restated: fields of the reference CNN (seeded random init) on the church image.
    The problem is ill-conditioned (focal sigma ~25 %): gate = a few times the fp32-vs-fp64 oracle spread."""
    g = np.load(os.path.join(GOLDEN, "golden_cnn.npz"))
    data = {k: g[k] for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    conf = {} if variant == "default" else {"num_steps": 20, "early_stop": False}
    out = run(conf, data, dev)
    ref = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(variant + "/")}
    compare_result(out, ref, TOL_CNN, f"cnn/{variant}")
    assert np.array_equal(out["stop_at"], ref["stop_at"])


@pytest.mark.parametrize("setname", ["pinhole", "simple_radial", "shared_pinhole", "shared_simple_radial"])
def test_hip_follows_reference_step_by_step(dev, setname):
    """Parameters after k = 1..n LM steps against the reference's recorded trajectory."""
    tr = np.load(os.path.join(GOLDEN, "golden_trace.npz"))
    ref_cam, ref_grav = tr[f"{setname}/camera"], tr[f"{setname}/gravity"]
    for k in range(1, ref_cam.shape[0] + 1):
        out = run({**conf_for(setname, "bench"), "num_steps": k}, data_for(setname, "bench"), dev)
        assert np.abs(out["camera"][:, 2:4] / ref_cam[k - 1][:, 2:4] - 1).max() < 3e-5, k
        assert np.abs(out["camera"][:, 6] - ref_cam[k - 1][:, 6]).max() < 3e-5, k
        assert np.abs(out["gravity"] - ref_grav[k - 1]).max() < 3e-5, k


TOL_SYSTEM_DIV_K = 1e-3      # single-sweep system, simple_divisional, k row / column only (camera.py:913)


@pytest.mark.parametrize("model", ALL_MODELS)
@pytest.mark.parametrize("mode", ["loop", "rpf"])
@pytest.mark.parametrize("row_pairs", [None, True])
def test_hip_single_sweep_system(dev, model, mode, row_pairs):
    """gclm_system: costs, J^T W r, J^T W J of ONE fused sweep at fixed, non-converged parameters against the reference's
    setup_system (`row_pairs`: the same gates on the row-pair walk of the sweep, radial / simple_divisional)."""
    from geocalib_amd import Gravity, LMOptimizer, camera_models
    if row_pairs and model not in ("radial", "simple_divisional"):
        pytest.skip("the row-pair walk exists for radial / simple_divisional")
    s = np.load(os.path.join(GOLDEN, "golden_system.npz"))
    inp = np.load(os.path.join(GOLDEN, f"inputs_{model}.npz"))
    data = {k: inp[k] for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    assert data["latitude_field"].shape[-2] % 2 == 0 and data["latitude_field"].shape[-1] % 4 == 0      # (pairs can be walked)
    opt = LMOptimizer({"camera_model": model}).eval()
    opt.row_pairs = row_pairs
    cam = camera_models[model](torch.from_numpy(s[f"{model}/camera"]).to(dev))
    grav = Gravity(torch.from_numpy(s[f"{model}/gravity"]).to(dev))
    out = to_np(opt.system(to_dev(data, dev), cam, grav, as_rpf=(mode == "rpf")))
    Hr, Gr = s[f"{model}/{mode}/H"], s[f"{model}/{mode}/G"]
    d = np.sqrt(np.abs(np.einsum("bii->bi", Hr)))
    # 5e-5 of the natural scale of each entry.  Named exception: simple_divisional's k row / column (index 3), whose
    # reference formulas cancel in float32 (camera.py:913) -- the other rows of that model keep the common gate
    tol = np.full(Hr.shape[1], 5e-5)
    if model == "simple_divisional":
        tol[3] = TOL_SYSTEM_DIV_K
    from conftest import MEASURED
    eh = (np.abs(out["H"] - Hr) / (d[:, :, None] * d[:, None, :])).max(0)
    cost = (s[f"{model}/{mode}/cost_up"] + s[f"{model}/{mode}/cost_lat"]) * data["latitude_field"][0].size
    eg = (np.abs(out["G"] - Gr) / (d * np.sqrt(cost)[:, None])).max(0)
    MEASURED[f"system/{model}/{mode}" + ("/row_pairs" if row_pairs else "")] = {"H_rows": eh.max(1).tolist(), "G": eg.tolist()}
    assert (eh < np.maximum(tol[:, None], tol[None, :])).all(), eh
    assert (eg < tol).all(), eg
    assert np.allclose(out["cost_up"], s[f"{model}/{mode}/cost_up"], rtol=2e-5)
    assert np.allclose(out["cost_lat"], s[f"{model}/{mode}/cost_lat"], rtol=2e-5)


# ------------------------------------------------------------------ against the oracle on other shapes

@pytest.mark.parametrize("model", ALL_MODELS)
@pytest.mark.parametrize("shape", [(50, 70), (33, 47), (96, 128), (7, 5)])
def test_hip_matches_oracle_odd_shapes(dev, oracle, model, shape):
    """Widths that are not multiples of 4 take the scalar-load path; tiny images take one workgroup."""
    from oracle import synth
    H, W = shape
    data, _, _ = synth.make_fields(99, range(3), model, H, W)
    conf = {"camera_model": model, "num_steps": 20, "early_stop": False}
    ref = oracle.solve(data, conf, precision="f32")
    out = run(conf, data, dev)
    tol = dict(TOL)
    if model == "simple_divisional":
        # the oracle is ANOTHER float32 evaluation of the reference's cancelling formulas (camera.py:913): it is only as
        # sharp a yardstick as it agrees with its own float64 build on these inputs
        from conftest import result_spread
        own = result_spread(ref, oracle.solve(data, conf, precision="f64"))
        tol.update(focal=1e-4 + 10 * own[0], gravity=1e-4 + 10 * own[1], dist=1e-4 + 10 * own[2], cost=1e-4 + 10 * own[3])
    if H * W < 100:
        tol.update(focal=5e-3, dist=5e-3, gravity=5e-3, cost=5e-3, cov=1e-1, unc=1e-1)   # 35 pixels: ill-posed
    compare_result(out, ref, tol, f"{model}/{shape}")


@pytest.mark.parametrize("model", HIP_MODELS)
@pytest.mark.parametrize("shape", [(640, 2600), (500, 2051), (320, 1280), (200, 516), (30, 36)])
def test_hip_matches_oracle_tile_geometries(dev, oracle, model, shape):
    """The column-stationary tiling of the sweep (plan_geometry): rows wider than 512 units are cut into strips (2600 px:
    650 float4 units -> 3 strips; 2051 px, scalar path: 2051 units -> strips with idle lanes), 1280 px = 320 units is one
    row per tile, 516 px = 129 units leaves idle lanes in every tile, 36 px packs 7 rows into a tile.  Cameras with a
    narrow vertical field of view, so that the wide images stay well-posed (the horizontal field of view below ~100 deg)."""
    from oracle import synth
    H, W = shape
    idx = [i for i in range(400) if synth.gt_params(7, i, model, H, W)[2][2] < np.deg2rad(28)][:2]
    data, cams, _ = synth.make_fields(7, idx, model, H, W)
    conf = {"camera_model": model, "num_steps": 20, "early_stop": False}
    ref = oracle.solve(data, conf, precision="f32")
    out = run(conf, data, dev)
    compare_result(out, ref, TOL, f"{model}/{shape}")
    assert np.abs(out["camera"][:, 3] / cams[:, 3] - 1).max() < 2e-2          # and it is the ground truth
    assert np.abs(out["stop_at"] - ref["stop_at"]).max() <= 1


def test_hip_unaligned_views_take_scalar_path(dev, oracle):
    """Field tensors whose storage is not 16-byte aligned must still be handled (scalar loads)."""
    from geocalib_amd import LMOptimizer
    from oracle import synth
    data, _, _ = synth.make_fields(5, range(2), "pinhole", 48, 64)
    conf = {"camera_model": "pinhole", "num_steps": 20, "early_stop": False}
    td = {}
    for k, v in data.items():
        flat = torch.zeros(v.size + 1, device=dev)
        flat[1:] = torch.from_numpy(v).to(dev).reshape(-1)
        td[k] = flat[1:].view(*v.shape)
        assert td[k].data_ptr() % 16 != 0 and td[k].is_contiguous()
    out = to_np(LMOptimizer(conf).eval()(td))
    compare_result(out, oracle.solve(data, conf, precision="f32"), TOL, "unaligned")


def test_known_answer_noise_free(dev):
    """The reference's own end-to-end test (siclib/geometry/gradient_checker.py:584-641, atol 1e-3):
    noise-free fields of a random camera are recovered (here: 320x320 as there, seeded)."""
    from geocalib_amd import Gravity, camera_models, perspective_fields as pf
    for model, spherical in (("pinhole", True), ("pinhole", False), ("simple_radial", True), ("simple_radial", False)):
        g = torch.Generator().manual_seed(17)
        B = 6
        roll = (torch.rand(B, generator=g) - 0.5) * np.pi / 2
        pitch = (torch.rand(B, generator=g) - 0.5) * np.pi / 2
        vfov = np.deg2rad(5) + torch.rand(B, generator=g) * np.deg2rad(75)
        d = {"height": torch.full((B,), 320.0), "width": torch.full((B,), 320.0), "vfov": vfov}
        if model != "pinhole":
            d["k1"] = torch.full((B,), -0.1)
        cam = camera_models[model].from_dict(d)
        grav = Gravity.from_rp(roll, pitch)
        up, lat = pf.get_perspective_field(cam, grav)
        out = run({"camera_model": model, "use_spherical_manifold": spherical},
                  {"up_field": up.contiguous().numpy(), "latitude_field": lat.contiguous().numpy()}, dev)
        assert np.allclose(out["camera"][:, 3], cam.f[:, 1].numpy(), rtol=1e-3, atol=1e-3), model
        assert np.allclose(out["gravity"], grav.vec3d.numpy(), atol=1e-3), model
        if model != "pinhole":
            assert np.allclose(out["camera"][:, 6], -0.1, atol=1e-3)


def test_training_mode_skips_uncertainty(dev):
    small = np.load(os.path.join(GOLDEN, "golden_small.npz"))
    out = run(conf_for("pinhole", "bench"), data_for("pinhole", "bench"), dev, training=True)
    assert not any("uncertainty" in k or k == "covariance" for k in out)
    assert np.abs(out["camera"][:, 2:4] / small["pinhole/training/camera"][:, 2:4] - 1).max() < 1e-4


def test_verbose_conf_logs_the_reference_lines(dev, caplog):
    """conf.verbose (lm_optimizer.py:652-662): five log lines -- the time, initial / optimised vfov and roll-pitch in degrees --
    and the same result as the quiet solve, bit for bit."""
    import logging
    from geocalib_amd import LMOptimizer
    data = to_dev(data_for("pinhole", "default"), dev)
    quiet = to_np(LMOptimizer(conf_for("pinhole", "default")).eval()(dict(data)))
    with caplog.at_level(logging.INFO, logger="geocalib_amd.lm_optimizer"):
        loud = to_np(LMOptimizer({**conf_for("pinhole", "default"), "verbose": True}).eval()(dict(data)))
    msgs = [r.getMessage() for r in caplog.records]
    assert len(msgs) == 5 and msgs[0].startswith("Optimization took") and msgs[0].endswith("ms")
    assert [m.split(":")[0] for m in msgs[1:]] == ["Initial camera", "Optimized camera", "Initial gravity", "Optimized gravity"]
    assert all(np.array_equal(quiet[k], loud[k], equal_nan=True) for k in quiet)


def test_reference_error_behaviour(dev):
    from geocalib_amd import LMOptimizer, _lib
    lat = torch.zeros(2, 1, 16, 16, device=dev)
    with pytest.raises(KeyError):                       # lm_optimizer.py:31
        LMOptimizer({})({"up_field": torch.zeros(2, 2, 16, 16, device=dev)})
    with pytest.raises(RuntimeError, match="HIP device"):
        LMOptimizer({})({"latitude_field": lat.cpu()})
    # the C ABI itself: null latitude pointer is an error with a message, not a crash
    opt = LMOptimizer({}).eval()
    h = opt._handle(dev)
    lib = _lib.load()
    buf = torch.zeros(2, 48, device=dev)
    rc = lib.gclm_solve(h.ptr, None, None, None, None, 2, 16, 16, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), None)
    assert rc != 0 and "latitude_field" in _lib.last_error(h.ptr)
    assert lib.gclm_solve(h.ptr, None, lat.data_ptr(), None, None, 0, 16, 16, buf.data_ptr(), buf.data_ptr(),
                          buf.data_ptr(), None) == 0    # empty batch is a no-op
    assert lib.gclm_workspace_bytes(h.ptr) >= 0
    # the tuning hooks validate their argument and say what was wrong
    assert lib.gclm_set_fused_steps(h.ptr, 5) == -3 and "gclm_set_fused_steps" in _lib.last_error(h.ptr)
    assert lib.gclm_set_sweep_iters(h.ptr, -1) == -3 and "gclm_set_sweep_iters" in _lib.last_error(h.ptr)
    assert lib.gclm_set_fused_steps(h.ptr, -1) == 0 and lib.gclm_set_sweep_iters(h.ptr, 0) == 0 and lib.gclm_set_stop_comm(h.ptr, None) == 0
    # a handle cannot change its device, and a config of another ABI generation is refused by gclm_configure as well
    cfg = opt._config(0)
    cfg.device = 1
    assert lib.gclm_configure(h.ptr, C.byref(cfg)) == -2 and "cannot move" in _lib.last_error(h.ptr)
    cfg = opt._config(0)
    cfg.abi_version = 100
    assert lib.gclm_configure(h.ptr, C.byref(cfg)) == -5 and "ABI mismatch" in _lib.last_error(h.ptr)


# ------------------------------------------------------------------ shared intrinsics extensions

def _shared16(model, groups):
    """Inputs of `groups` BASELINE configs[4] groups (16 frames of one camera, 640x480) concatenated."""
    from oracle import synth
    parts = [synth.make_shared_group(1234, g, model, 480, 640, frames=16) for g in groups]
    data = {k: np.concatenate([p[0][k] for p in parts]) for k in parts[0][0]}
    return data, parts


@pytest.mark.parametrize("model", HIP_MODELS)
def test_hip_matches_reference_shared16_at_shape(dev, model):
    """BASELINE configs[4] AT ITS STATED SHAPE against the REFERENCE: two shared-intrinsics groups of 16 frames at
    640x480 solved in ONE call (`group_size: 16`), each compared with the reference's own call on that group
    (tests/golden/make_golden_shared16.py; lm_optimizer.py:350-383, 597-603).  Gate: north_star's 1e-4."""
    g = np.load(os.path.join(GOLDEN, "golden_shared16.npz"))
    data, parts = _shared16(model, (0, 1))
    for i, (d, _, _) in enumerate(parts):
        chk = np.array([np.float64(np.asarray(v, np.float64).sum()) for _, v in sorted(d.items())])
        assert np.allclose(chk, g[f"{model}/g{i}/input_checksum"], rtol=1e-9, atol=1e-3), "regenerated inputs drifted"
    conf = {"camera_model": model, "shared_intrinsics": True, "group_size": 16, "num_steps": 20, "early_stop": False}
    out = run(conf, data, dev)
    for i in range(2):
        ref = {k.split("/", 2)[2]: g[k] for k in g.files if k.startswith(f"{model}/g{i}/")}
        sub = {k: v[16 * i:16 * (i + 1)] for k, v in out.items()}
        compare_result(sub, ref, TOL, f"shared16/{model}/g{i}")
        # stop_at = first step after which ALL 16 costs moved by < 1e-8 (lm_optimizer.py:90-92, 619-620).  With the
        # fixed lambda of shared mode (:612) the cost creeps through that threshold over several steps, so the step
        # it is crossed at is rounding noise: the same step or its neighbour
        assert np.abs(sub["stop_at"] - ref["stop_at"]).max() <= 1, (sub["stop_at"], ref["stop_at"])
        assert np.abs(sub["camera"][:, 2:4] - sub["camera"][0, 2:4]).max() == 0      # one camera per group
    assert out["step_failures"].max() == 0


@pytest.mark.parametrize("model", HIP_MODELS)
def test_hip_matches_oracle_shared16_eight_groups(dev, oracle, model):
    """Eight 16-frame groups at 640x480 in one call against the oracle run group by group (the reference has no
    group dimension: 512 groups = 512 calls)."""
    groups = tuple(range(2, 10))
    data, parts = _shared16(model, groups)
    conf = {"camera_model": model, "shared_intrinsics": True, "num_steps": 20, "early_stop": False}
    out = run({**conf, "group_size": 16}, data, dev)
    from conftest import measure_result
    for i, (d, cams, _) in enumerate(parts):
        ref = oracle.solve(d, conf, precision="f32")
        sub = {k: v[16 * i:16 * (i + 1)] for k, v in out.items()}
        # the roll uncertainty of a frame that looks almost straight up or down is ill-conditioned (Gravity.J_rp at
        # |g_z| -> 1, gravity.py:69-101): the oracle's own float32 and float64 builds differ by 1e-4 there, so the
        # uncertainties get 10 x that on top of their gate
        own = measure_result(ref, oracle.solve(d, conf, precision="f64"))
        compare_result(sub, ref, {**TOL, "unc": TOL["unc"] + 10 * own["unc"], "cov": TOL["cov"] + 10 * own["cov"]},
                       f"shared16x8/{model}/g{groups[i]}")
        assert np.abs(sub["camera"][0, 3] / cams[0, 3] - 1) < 1e-2                    # and it is the ground truth


def test_group_size_equals_independent_shared_solves(dev):
    """group_size splits a batch into independent shared-intrinsics groups (512 x 16 in BASELINE config 5):
    identical to one reference-style call per group."""
    d = data_for("shared_pinhole", "bench")
    conf = conf_for("shared_pinhole", "bench")
    single = run(conf, d, dev)
    d2 = {k: np.concatenate([v, v[::-1].copy()]) for k, v in d.items()}
    both = run({**conf, "group_size": 4}, d2, dev)
    assert np.array_equal(both["camera"][:4], single["camera"]) and np.array_equal(both["gravity"][:4], single["gravity"])
    rev = run(conf, {k: v[::-1].copy() for k, v in d.items()}, dev)
    assert np.allclose(both["camera"][4:], rev["camera"], rtol=1e-6) and np.allclose(both["gravity"][4:], rev["gravity"], atol=1e-6)
    assert np.allclose(single["camera"][:, 3], single["camera"][0, 3], rtol=1e-6)   # one focal for the group


@pytest.mark.parametrize("setname", ["shared_pinhole", "shared_simple_radial"])
def test_split_api_single_rank_equals_solve(dev, setname):
    """gclm_shared_begin/reduce/apply/finish (the multi-GPU protocol) with one rank == gclm_solve."""
    from geocalib_amd import LMOptimizer
    from geocalib_amd.parallel import SharedIntrinsicsSplit
    conf, d = conf_for(setname, "bench"), data_for(setname, "bench")
    single = run(conf, d, dev)
    opt = LMOptimizer(conf).eval()
    B = d["latitude_field"].shape[0]
    out = to_np(SharedIntrinsicsSplit(opt, num_groups=1)(to_dev(d, dev), torch.zeros(B, dtype=torch.int32)))
    for k in ("camera", "gravity", "final_cost"):
        assert np.array_equal(out[k], single[k]), k
    # the uncertainty sweep of a plain solve runs in its log-focal form (fx == fy is known there, iso_final); the split
    # protocol starts from caller-provided cameras and takes the general focal column: same numbers up to rounding
    for k in ("covariance", "focal_uncertainty"):
        assert np.allclose(out[k], single[k], rtol=2e-5, atol=0), k
    compare_result(out, golden_outputs(setname, "bench"), TOL, f"split/{setname}")


# ------------------------------------------------------------------ properties at the BASELINE size

def synth_device(model, B, H, W, dev, seed=11, first=0, **kw):
    from geocalib_amd.synth import synth_fields
    out = synth_fields(model, B, H, W, dev, seed=seed, first_index=first, **kw)
    torch.cuda.synchronize()
    return out


def test_synth_generator_is_index_keyed(dev):
    a, ca, ga = synth_device("simple_radial", 6, 48, 64, dev, first=0)
    b, cb, gb = synth_device("simple_radial", 3, 48, 64, dev, first=3)
    for k in a:
        assert torch.equal(a[k][3:], b[k]), k
    assert torch.equal(ca[3:], cb) and torch.equal(ga[3:], gb)
    assert torch.allclose(a["up_field"].norm(dim=1), torch.ones(6, 48, 64, device=dev), atol=1e-5)
    assert a["latitude_field"].abs().max() <= np.pi / 2 and 0 <= a["up_confidence"].min() and a["up_confidence"].max() <= 1
    # grouped / strided variant (multi-GPU frame split): frames of a group share the intrinsics and a
    # rank's strided runs equal the corresponding images of the contiguous generation
    g, cg, gg = synth_device("simple_radial", 8, 48, 64, dev, group_size=4)
    assert torch.equal(cg[:4, 2], cg[0, 2].expand(4)) and torch.equal(cg[4:, 6], cg[4, 6].expand(4)) and cg[0, 2] != cg[4, 2]
    assert not torch.equal(gg[0], gg[1])
    r1, c1, g1 = synth_device("simple_radial", 4, 48, 64, dev, first=2, group_size=4, run=2, run_stride=4)
    sel = [2, 3, 6, 7]
    for k in g:
        assert torch.equal(g[k][sel], r1[k]), k
    assert torch.equal(cg[sel], c1) and torch.equal(gg[sel], g1)


@pytest.mark.parametrize("model", ALL_MODELS)
def test_synth_generator_renders_the_reference_field(dev, oracle, model):
    """The device generator behind bench.py's inputs (synth_kernel) at the FIELD level: with noise = 0 its up /
    latitude planes are the perspective field of its own ground truth as `oracle.render` evaluates it in float64
    (tests/test_oracle.py pins that to the reference's get_perspective_field, perspective_fields.py:278); with the bench's
    noise the deviation from that field is N(0, sigma) per component (latitude directly; up: the tangential part, the
    field is re-normalised), sigma = 0.02 +- 2 %, and the confidences are U(0, 1)."""
    from conftest import MEASURED
    B, H, W = 3, 96, 128
    clean, cam, grav = synth_device(model, B, H, W, dev, seed=5, noise=0.0)
    up_ref, lat_ref = oracle.render(model, H, W, cam.cpu().numpy(), grav.cpu().numpy(), precision="f64")
    assert np.array_equal(cam[:, :2].cpu().numpy(), np.tile(np.float32([W, H]), (B, 1)))
    up, lat = clean["up_field"].cpu().numpy(), clean["latitude_field"].cpu().numpy()
    lim = np.pi / 2 - 1e-3                                    # the generator clamps the latitude there (as oracle/synth.py)
    inside = np.abs(lat_ref) < lim - 1e-4
    assert inside.mean() > 0.99
    # latitude = asin(s): towards the poles one float32 ulp of s is 1 / cos(latitude) ulps of the latitude (x900 at the
    # generator's clamp), so the latitude itself is held to 2e-6 below 1.3 rad and its SINE -- what the residual uses
    # (lm_optimizer.py:262,270) -- everywhere
    mid = inside & (np.abs(lat_ref) < 1.3)
    d_up, d_lat, d_sin = np.abs(up - up_ref).max(), np.abs(lat - lat_ref)[mid].max(), np.abs(np.sin(lat.astype(np.float64)) - np.sin(lat_ref.astype(np.float64)))[inside].max()
    MEASURED[f"synth_field/{model}"] = {"up": float(d_up), "latitude_below_1.3rad": float(d_lat), "sin_latitude": float(d_sin),
                                        "share_below_1.3rad": float(mid.mean())}
    assert d_up <= 2e-6 and d_lat <= 2e-6 and d_sin <= 2e-6, (model, d_up, d_lat, d_sin)
    sigma = 0.02
    noisy, cam_n, grav_n = synth_device(model, B, H, W, dev, seed=5, noise=sigma)
    assert torch.equal(cam_n, cam) and torch.equal(grav_n, grav)
    un, ln = noisy["up_field"].cpu().numpy().astype(np.float64), noisy["latitude_field"].cpu().numpy().astype(np.float64)
    assert np.abs(np.sqrt((un ** 2).sum(1)) - 1).max() < 1e-6
    tangential = up_ref[:, 0] * un[:, 1] - up_ref[:, 1] * un[:, 0]          # sin(angle between the noisy and the clean up vector)
    unclamped = inside & (np.abs(ln) < lim - 1e-6)
    dl = (ln - lat_ref)[unclamped]
    n = tangential.size
    stats = {"up_tangential_std": float(tangential.std()), "up_tangential_mean": float(tangential.mean()),
             "latitude_std": float(dl.std()), "latitude_mean": float(dl.mean())}
    MEASURED[f"synth_noise/{model}"] = stats
    for key in ("up_tangential", "latitude"):
        assert abs(stats[key + "_std"] / sigma - 1) < 0.02, stats
        assert abs(stats[key + "_mean"]) < 5 * sigma / np.sqrt(n), stats
    for key in ("up_confidence", "latitude_confidence"):
        c = noisy[key].cpu().numpy().astype(np.float64)
        assert 0 < c.min() and c.max() < 1 and abs(c.mean() - 0.5) < 5 / np.sqrt(12 * n) and abs(c.var() * 12 - 1) < 0.03, key
    # independent draws per plane and pixel: no correlation between the two noise components
    assert abs(np.corrcoef(tangential[unclamped[:, 0]], dl)[0, 1]) < 5 / np.sqrt(dl.size)


@pytest.mark.parametrize("model", HIP_MODELS)
def test_full_size_batch_properties(dev, oracle, model):
    """BASELINE configs[1]/[3]: B=1024, 640x480, 20 iterations.  Size-independent properties:
    determinism, batch-permutation equivariance, shard invariance, weight-scale invariance, ground truth
    recovery; plus exact parity with the oracle on a sample of the same device-generated images."""
    from geocalib_amd import LMOptimizer
    B, H, W = 1024, 480, 640
    data, gtc, gtg = synth_device(model, B, H, W, dev)
    opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
    a = to_np(opt(data))
    b = to_np(opt(data))
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), f"non-deterministic {k}"
    assert a["step_failures"].max() == 0 and np.isfinite(a["camera"]).all() and np.isfinite(a["covariance"]).all()
    # ground truth within the noise level (sigma = 0.02 on 307k pixels)
    assert np.median(np.abs(a["camera"][:, 3] / gtc[:, 3].cpu().numpy() - 1)) < 1e-3
    assert np.median(np.abs(a["gravity"] - gtg.cpu().numpy()).max(1)) < 5e-4
    if model != "pinhole":
        assert np.median(np.abs(a["camera"][:, 6] - gtc[:, 6].cpu().numpy())) < 2e-3
    # permutation equivariance: images are independent, bit for bit
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(dev)
    p = to_np(opt({k: v[perm].contiguous() for k, v in data.items()}))
    pn = perm.cpu().numpy()
    assert np.array_equal(p["camera"], a["camera"][pn]) and np.array_equal(p["gravity"], a["gravity"][pn])
    # shard invariance (what the 8-GPU image sharding relies on): a sub-batch gives the same answer
    lo, hi = 640, 768
    s = to_np(opt({k: v[lo:hi].contiguous() for k, v in data.items()}))
    assert np.abs(s["camera"][:, 2:4] / a["camera"][lo:hi, 2:4] - 1).max() < 1e-5
    assert np.abs(s["gravity"] - a["gravity"][lo:hi]).max() < 1e-5
    # scaling every confidence by a constant leaves the optimum unchanged (costs scale linearly)
    half = dict(data)
    half["up_confidence"] = data["up_confidence"][:128] * 0.5
    half["latitude_confidence"] = data["latitude_confidence"][:128] * 0.5
    half["up_field"], half["latitude_field"] = data["up_field"][:128], data["latitude_field"][:128]
    hs = to_np(opt(half))
    assert np.abs(hs["camera"][:, 3] / a["camera"][:128, 3] - 1).max() < 2e-5
    assert np.allclose(hs["final_cost"], 0.5 * a["final_cost"][:128], rtol=1e-4)
    # oracle parity on a sample of exactly these images
    idx = [0, 1, 511, 1023]
    sample = {k: v[idx].cpu().numpy() for k, v in data.items()}
    ref = oracle.solve(sample, {"camera_model": model, "num_steps": 20, "early_stop": False}, precision="f32")
    compare_result({k: v[idx] for k, v in a.items()}, ref, TOL, f"B1024/{model}")


def run_virtual_ranks(dev, conf, data, selections, gofs, G, H, W):
    """The split protocol of BASELINE configs[4] on ONE device: rank r = its own LMOptimizer / gclm_handle holding the
    frames `selections[r]` (indices into `data`, sorted by group) with local group ids `gofs[r]`.  Per LM step every
    rank reduces its frames to Schur partials (gclm_shared_reduce), the buffers are summed (what the RCCL all-reduce
    does) and every rank applies (gclm_shared_apply).  Returns per rank (cam, grav, info) as numpy."""
    from geocalib_amd import LMOptimizer, _lib, get_trivial_estimation
    lib = _lib.load()
    ranks = []
    for sel, gof in zip(selections, gofs):
        local = {k: v[sel].contiguous() for k, v in data.items()}
        opt = LMOptimizer(conf).eval()
        cam0, grav0 = get_trivial_estimation(local, opt.camera_model)
        opt.setup_optimization_and_priors(local, shared_intrinsics=True)
        up, lat, upc, latc, (B, _, _) = opt._fields(local)
        h = opt._handle(dev)
        keep_alive = torch.zeros(16, device=dev)                     # an empty tensor has a null data_ptr
        st = dict(opt=opt, h=h, sel=sel, cam=cam0._data.clone(), grav=grav0._data.clone(), keep=(up, lat, upc, latc),
                  gof=gof.to(torch.int32).contiguous() if gof.numel() else torch.zeros(1, device=dev, dtype=torch.int32),
                  part=torch.zeros((G, _lib.SHARED_PARTIAL_STRIDE), device=dev),
                  info=torch.empty((B, _lib.INFO_STRIDE), device=dev), keep_alive=keep_alive)
        ptr = lambda t: t.data_ptr() if t.numel() else keep_alive.data_ptr()  # noqa: E731
        _lib.check(lib.gclm_shared_begin(h.ptr, ptr(up), ptr(lat), ptr(upc), ptr(latc), B, H, W,
                                         ptr(st["cam"]), ptr(st["grav"]), st["gof"].data_ptr(), G, None), h.ptr)
        ranks.append(st)
    for step in range(conf["num_steps"]):
        for st in ranks:
            _lib.check(lib.gclm_shared_reduce(st["h"].ptr, step, st["part"].data_ptr(), None), st["h"].ptr)
        total = torch.stack([st["part"] for st in ranks]).sum(0)      # the all-reduce
        for st in ranks:
            st["part"].copy_(total)
            _lib.check(lib.gclm_shared_apply(st["h"].ptr, step, st["part"].data_ptr(), None), st["h"].ptr)
    for st in ranks:
        _lib.check(lib.gclm_shared_finish(st["h"].ptr, st["info"].data_ptr() or st["keep_alive"].data_ptr(), None), st["h"].ptr)
    torch.cuda.synchronize()
    return [(st["cam"].cpu().numpy(), st["grav"].cpu().numpy(), st["info"].cpu().numpy(), st["opt"]) for st in ranks]


@pytest.mark.parametrize("split", ["interleaved", "empty_rank"])
@pytest.mark.parametrize("model", HIP_MODELS)
def test_split_protocol_two_virtual_ranks(dev, model, split):
    """BASELINE configs[4] protocol on ONE device: the frames of every group are dealt to two handles
    ("ranks"); per LM step both reduce their local Schur partials, the partial buffers are summed (what the
    RCCL all-reduce does), both apply.  Must equal the single-handle shared-intrinsics solve."""
    from geocalib_amd import LMOptimizer, _lib
    G, gs, H, W = 6, 8, 48, 64
    data, gtc, gtg = synth_device(model, G * gs, H, W, dev, seed=3, group_size=gs)
    conf = {"camera_model": model, "num_steps": 12, "early_stop": False, "shared_intrinsics": True, "group_size": gs}
    single = to_np(LMOptimizer(conf).eval()(data))
    frames = torch.arange(G * gs, device=dev)
    sels, gofs = [], []
    for r in range(2):
        if split == "interleaved":
            sel = frames[(frames % gs) // (gs // 2) == r]             # rank r: frames [r*gs/2, (r+1)*gs/2) of each group
            gof = torch.arange(sel.numel(), device=dev, dtype=torch.int32) // (gs // 2)
        else:                                                         # rank 1 holds NO frame: it only joins the all-reduce
            sel = frames if r == 0 else frames[:0]
            gof = (sel // gs).to(torch.int32)
        sels.append(sel); gofs.append(gof)
    res = run_virtual_ranks(dev, conf, data, sels, gofs, G, H, W)
    for (cam, grav, info, _), sel in zip(res, sels):
        sel = sel.cpu().numpy()
        if sel.size == 0:
            continue
        assert np.abs(cam[:, 2:4] / single["camera"][sel, 2:4] - 1).max() < 2e-6
        assert np.abs(cam[:, 6] - single["camera"][sel, 6]).max() < 2e-6
        assert np.abs(grav - single["gravity"][sel]).max() < 2e-6
        assert np.allclose(info[:, _lib.INFO["final_cost"]], single["final_cost"][sel], rtol=1e-5)
    if split == "empty_rank":       # rank 0 did everything: the protocol IS the single-device solve, bit for bit
        assert np.array_equal(res[0][0], single["camera"])
        assert np.array_equal(res[0][1], single["gravity"])
    # one focal per group
    f = single["camera"][:, 3].reshape(G, gs)
    assert np.abs(f / f[:, :1] - 1).max() < 1e-6


@pytest.mark.parametrize("model", HIP_MODELS)
def test_configs4_partition_on_eight_virtual_ranks_matches_reference(dev, model):
    """BASELINE configs[4]'s ACTUAL partition against the REFERENCE: 16-frame shared-intrinsics groups at 640x480 whose
    frames are split over 8 ranks (2 frames of every group per rank, `group_of_frame` = [0, 0, 1, 1]), one sum of the
    Schur partials per LM step.  Every rank's frames must match the reference's own solve of the whole group
    (tests/golden/golden_shared16.npz; lm_optimizer.py:350-383, 597-603) within north_star's 1e-4."""
    from geocalib_amd.parallel import frame_split_layout, global_frame_index
    g = np.load(os.path.join(GOLDEN, "golden_shared16.npz"))
    data_np, parts = _shared16(model, (0, 1))
    data = to_dev(data_np, dev)
    G, gs, world, H, W = 2, 16, 8, 480, 640
    conf = {"camera_model": model, "shared_intrinsics": True, "group_size": gs, "num_steps": 20, "early_stop": False}
    sels, gofs = [], []
    for r in range(world):
        lay = frame_split_layout(G * gs // world, gs, world, r)          # bench.py's own index arithmetic
        b = torch.arange(G * gs // world, device=dev)
        sels.append(global_frame_index(b, lay))
        gofs.append((b // lay["fpg"]).to(torch.int32))
    assert torch.equal(torch.cat(sels).sort().values, torch.arange(G * gs, device=dev))
    res = run_virtual_ranks(dev, conf, data, sels, gofs, G, H, W)
    cam, grav = np.zeros((G * gs, 8), np.float32), np.zeros((G * gs, 3), np.float32)
    info = {}
    for (c, gv, inf, opt), sel in zip(res, sels):
        sel = sel.cpu().numpy()
        cam[sel], grav[sel] = c, gv
        for k, v in opt._unpack_info(torch.from_numpy(inf), True).items():
            info.setdefault(k, np.zeros((G * gs,) + tuple(v.shape[1:]), np.float32))[sel] = v.numpy()
    out = {"camera": cam, "gravity": grav, **info}
    for i in range(G):
        ref = {k.split("/", 2)[2]: g[k] for k in g.files if k.startswith(f"{model}/g{i}/")}
        sub = {k: v[gs * i:gs * (i + 1)] for k, v in out.items()}
        # the covariance of the reference is the one of the per-frame (roll, pitch, focal) system: every rank computes it
        # for its own frames from the same final estimate
        compare_result(sub, ref, TOL, f"shared16-split8/{model}/g{i}")
        assert np.abs(sub["camera"][:, 2:4] - sub["camera"][0, 2:4]).max() == 0      # one camera per group, on every rank
    assert out["step_failures"].max() == 0


@pytest.mark.parametrize("model", HIP_MODELS)
def test_configs4_per_rank_shape_512_groups_x_2_frames(dev, model):
    """The per-rank SHAPE of the 8-GPU configs[4] run -- 512 groups, 2 local frames of each, `group_of_frame` with 512
    distinct values, shared_step_kernel<.., false> over 512 blocks of 2 frames -- on 8 virtual ranks (48x64 frames so
    that the 8192 frames of the whole job fit comfortably): equal to the single-handle solve of the 512 whole groups."""
    from geocalib_amd import LMOptimizer
    from geocalib_amd.parallel import frame_split_layout, global_frame_index
    G, gs, world, H, W = 512, 16, 8, 48, 64
    data, gtc, _ = synth_device(model, G * gs, H, W, dev, seed=5, group_size=gs)
    conf = {"camera_model": model, "num_steps": 10, "early_stop": False, "shared_intrinsics": True, "group_size": gs}
    single = to_np(LMOptimizer(conf).eval()(data))
    sels, gofs = [], []
    for r in range(world):
        lay = frame_split_layout(G * gs // world, gs, world, r)
        assert (lay["fpg"], lay["n_groups"]) == (2, 512)
        b = torch.arange(G * gs // world, device=dev)
        sels.append(global_frame_index(b, lay))
        gofs.append((b // lay["fpg"]).to(torch.int32))
        assert gofs[-1].unique().numel() == 512
    res = run_virtual_ranks(dev, conf, data, sels, gofs, G, H, W)
    for (cam, grav, info, _), sel in zip(res, sels):
        sel = sel.cpu().numpy()
        # HIP vs HIP, only the order of the sum over a group's frames differs (8 rank partials vs frame order): rounding,
        # carried through 10 unconverged steps on 48-px-high frames
        assert np.abs(cam[:, 2:4] / single["camera"][sel, 2:4] - 1).max() < 5e-5
        assert np.abs(cam[:, 6] - single["camera"][sel, 6]).max() < 5e-5
        assert np.abs(grav - single["gravity"][sel]).max() < 5e-5
    f = single["camera"][:, 3].reshape(G, gs)
    assert np.abs(f / f[:, :1] - 1).max() < 1e-6 and np.median(np.abs(f[:, 0] / gtc[::gs, 3].cpu().numpy() - 1)) < 2e-2


def test_latitudes_beyond_halfpi_follow_torch_sin(dev):
    """The reference takes torch.sin of whatever latitude field it is handed (lm_optimizer.py:262, 270); the sweep's
    polynomial covers [-pi/2, pi/2] and anything beyond is folded into that range first.  Per-pixel residuals against
    torch.sin in float64 for latitudes up to +-90 (a caller feeding degrees), the fused sweep against the mean of the
    per-pixel costs (vector path, scalar path), and in-range data is untouched by the fold (bit for bit)."""
    from geocalib_amd import LMOptimizer
    from geocalib_amd.lm_optimizer import get_trivial_estimation
    for W in (64, 50):                                                   # float4 path / scalar-load path
        data, _, _ = synth_device("simple_radial", 3, 48, W, dev, seed=13)
        opt = LMOptimizer({"camera_model": "simple_radial"}).eval()
        cam, grav = get_trivial_estimation(data, opt.camera_model)
        base = opt.calculate_residuals(cam, grav, data)["latitude_residual"].double()
        gen = torch.Generator().manual_seed(W)
        wild = dict(data)
        off = torch.zeros_like(data["latitude_field"])
        off[0] = (torch.rand(off[0].shape, generator=gen) * 180 - 90).to(dev)             # "degrees"
        off[1, :, ::7, ::5] = (torch.randint(-3, 4, off[1, :, ::7, ::5].shape, generator=gen) * np.pi).float().to(dev)
        off[2, :, 10, 11] = 1.6                                                            # one pixel of one wave
        wild["latitude_field"] = (data["latitude_field"] + off).contiguous()
        got = opt.calculate_residuals(cam, grav, wild)["latitude_residual"].double()
        want = base + (torch.sin(wild["latitude_field"].double()) - torch.sin(data["latitude_field"].double())).reshape(base.shape)
        assert (got - want).abs().max() < 5e-7, (got - want).abs().max()
        # the fused sweep (its own fold, behind the wave-uniform branch) agrees with the per-pixel path
        res = opt.calculate_residuals(cam, grav, wild)
        costs, _ = opt.calculate_costs(res, wild)
        sysm = opt.system(wild, cam, grav)
        assert torch.allclose(costs["latitude_cost"].mean(1), sysm["cost_lat"], rtol=2e-5)
        assert torch.allclose(costs["up_cost"].mean(1), sysm["cost_up"], rtol=2e-5)
        # a field shifted by 2 pi everywhere solves to the same calibration
        shifted = dict(data)
        shifted["latitude_field"] = data["latitude_field"] + 2 * np.pi
        conf = {"camera_model": "simple_radial", "num_steps": 20, "early_stop": False}
        a, b = run_dev(conf, data), run_dev(conf, shifted)
        assert np.abs(a["camera"][:, 3] / b["camera"][:, 3] - 1).max() < 1e-4 and np.abs(a["gravity"] - b["gravity"]).max() < 1e-4


@pytest.mark.parametrize("model", HIP_MODELS)
@pytest.mark.parametrize("knob", ["heuristic", "squared_loss"])
def test_hip_matches_reference_siclib_knobs(dev, model, knob):
    """siclib's training-time knobs: heuristic initialisation evaluated in the device init kernel, and the
    squared loss (= Huber with an unreachable threshold, exact by power-of-two scaling)."""
    g = np.load(os.path.join(GOLDEN, "golden_extra.npz"))
    ref = {k.split("/", 2)[2]: g[k] for k in g.files if k.startswith(f"{model}/{knob}/")}
    conf = {"camera_model": model, "num_steps": 20, "early_stop": False}
    conf |= {"init_conf": {"name": "heuristic"}} if knob == "heuristic" else {"loss_fn": "squared_loss"}
    out = run(conf, data_for(model, "bench"), dev)
    compare_result(out, ref, TOL, f"{model}/{knob}")
    if knob == "heuristic":
        init = run({**conf, "num_steps": 0}, data_for(model, "bench"), dev)
        assert np.allclose(init["camera"], ref["init_camera"], rtol=5e-6)
        assert np.allclose(init["gravity"], ref["init_gravity"], atol=5e-6)


@pytest.mark.parametrize("shape", [(3, 96, 128), (2, 33, 47)])
@pytest.mark.parametrize("conf", [True, False])
def test_pack_fields_matches_torch_head_epilogue(dev, shape, conf):
    """gclm_pack_fields against the plain-PyTorch fp32 expression of the reference's head epilogues
    (geocalib/geocalib.py:57,73-75), vectorised and scalar paths, with and without confidences, in place too."""
    from geocalib_amd.fields import pack_fields
    B, H, W = shape
    g = torch.Generator().manual_seed(0)
    up_raw = (torch.randn(B, 2, H, W, generator=g) * 3).to(dev)
    lat_raw = (torch.randn(B, 1, H, W, generator=g) * 2).to(dev)
    ulc = torch.randn(B, H, W, generator=g).to(dev) if conf else None
    llc = torch.randn(B, 1, H, W, generator=g).to(dev) if conf else None
    up_raw[0, :, 0, 0] = 0                                   # F.normalize eps path
    ref = {"up_field": torch.nn.functional.normalize(up_raw, dim=1),
           "latitude_field": torch.asin(torch.clamp(torch.tanh(lat_raw), -1 + 1e-5, 1 - 1e-5))}
    if conf:
        ref |= {"up_confidence": torch.sigmoid(ulc), "latitude_confidence": torch.sigmoid(llc[:, 0])}
    out = pack_fields(up_raw, lat_raw, ulc, llc)
    assert set(out) == set(ref)
    for k in ref:
        assert out[k].shape == ref[k].shape, k
        assert torch.allclose(out[k], ref[k], atol=2e-6, rtol=2e-6), (k, (out[k] - ref[k]).abs().max().item())
    inp = pack_fields(up_raw.clone(), lat_raw.clone(), None if ulc is None else ulc.clone(),
                      None if llc is None else llc.clone(), inplace=True)
    for k in ref:
        assert torch.equal(inp[k], out[k]), k
    # and the packed planes feed the optimiser directly
    from geocalib_amd import LMOptimizer
    res = LMOptimizer({"num_steps": 3, "early_stop": False}).eval()(out)
    assert torch.isfinite(res["camera"]._data).all()


def test_sharded_early_stop_through_the_stop_communicator(dev):
    """The reference's batch-global early stop for a batch sharded over ranks (gclm_set_stop_comm: the per-step counters of
    images whose cost still moved are summed over the ranks on the solve's stream).  One GPU: (1) with a one-rank
    communicator the all-reduce is the identity, so the sharded call equals the plain solve bit for bit, `stop_at`
    included; (2) the counter protocol itself -- a shard alone stops EARLIER than the batch it belongs to (its own
    slowest image decides), which is exactly what the summed counters prevent: the slice of the whole-batch solve is
    reproduced by that shard only when it is told the other shard's counters."""
    import os as _os
    from geocalib_amd import LMOptimizer, _lib
    from geocalib_amd.parallel import RcclComm, calibrate_sharded
    lib = _lib.load()
    data, _, _ = synth_device("simple_radial", 12, 96, 128, dev, seed=77)
    conf = {"camera_model": "simple_radial"}                           # the default conf: 30 steps, early stop
    whole = run_dev(conf, data)
    comm = RcclComm(RcclComm.unique_id(), 1, 0, 0)
    _os.environ["GCLM_FORCE_COLLECTIVES"] = "1"
    try:
        opt = LMOptimizer(conf).eval()
        out = to_np(calibrate_sharded(opt, data, 12, comm=comm))
    finally:
        _os.environ.pop("GCLM_FORCE_COLLECTIVES", None)
    for k in ("camera", "gravity", "final_cost", "stop_at"):
        assert np.array_equal(out[k], whole[k]), k
    assert 1 < whole["stop_at"][0] < 30
    h = opt._handle(dev)
    assert lib.gclm_set_stop_comm(h.ptr, None) == 0                     # and it was unset again after the call
    # (1b) an EMPTY local shard (n_total < world, or an uneven split) must still take part in the per-step collectives of
    # its peers instead of failing the shape checks (a zero-size tensor's pointer is NULL): through the Python
    # wrapper, with the stop communicator set, the call goes through and returns empty results
    _os.environ["GCLM_FORCE_COLLECTIVES"] = "1"
    try:
        none = calibrate_sharded(LMOptimizer(conf).eval(), {k: v[:0] for k, v in data.items()}, 0, comm=comm)
    finally:
        _os.environ.pop("GCLM_FORCE_COLLECTIVES", None)
    torch.cuda.synchronize()
    assert none["camera"]._data.shape == (0, 8) and none["stop_at"].shape == (0,)
    # ... and the C entry point itself: B = 0 with NULL fields and outputs is not an error, with or without a stop communicator
    for c in (None, comm._ptr):
        assert lib.gclm_set_stop_comm(h.ptr, c) == 0
        assert lib.gclm_calibrate(h.ptr, None, None, None, None, 0, 96, 128, None, None, None, None, 0, None, None, None,
                                  torch.cuda.current_stream(dev).cuda_stream) == 0, _lib.last_error(h.ptr)
        assert lib.gclm_solve(h.ptr, None, None, None, None, 0, 96, 128, None, None, None,
                              torch.cuda.current_stream(dev).cuda_stream) == 0, _lib.last_error(h.ptr)
    assert lib.gclm_set_stop_comm(h.ptr, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(to_np(opt(data))["camera"], whole["camera"])   # the handle is as good as before
    # (2) shards on their own: each stops when ITS images have converged -- generally not where the batch stops
    lo, hi = run_dev(conf, {k: v[:6] for k, v in data.items()}), run_dev(conf, {k: v[6:] for k, v in data.items()})
    assert max(lo["stop_at"][0], hi["stop_at"][0]) == whole["stop_at"][0]        # the slower shard IS the batch's stop
    if lo["stop_at"][0] != hi["stop_at"][0]:
        early = lo if lo["stop_at"][0] < hi["stop_at"][0] else hi
        sl = slice(0, 6) if early is lo else slice(6, 12)
        assert not np.array_equal(early["camera"], whole["camera"][sl])           # per-shard decisions change the answer


@pytest.mark.parametrize("model", HIP_MODELS)
def test_sharded_early_stop_at_the_baseline_size_through_rccl(dev, model):
    """VERDICT r04 #6a: `calibrate_sharded(early_stop=True)` at B = 1024, 640x480 with the stop counters travelling through
    gclm_set_stop_comm, on the direct RCCL route AND on the torch ("nccl") route, one rank: `stop_at` and every result bit
    equal the plain call's (tests/rccl_stop_probe.py, its own process: it initialises a process group)."""
    import json
    import subprocess
    import sys
    from conftest import MEASURED, ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NCCL_DEBUG", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_stop_probe.py"), model, "1024", str(29700 + os.getpid() % 200)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    MEASURED[f"rccl_stop/{model}"] = out
    st = out["stop_at"]
    assert st["plain"] == st["gclm_comm"] == st["torch_nccl"] and 1 < st["plain"] < 30, st
    assert out["identical_to_plain"] == {"gclm_comm": True, "torch_nccl": True}, out
    assert out["torch_route_made_a_stop_communicator"] and out["rccl"]["runtime"] // 10000 == 2, out


def test_rccl_c_abi_single_rank(dev):
    """gclm_comm_* (direct RCCL behind the C ABI) with a one-rank communicator: both collectives are the identity,
    and the shared-intrinsics split driven through it equals the plain solve.  (Eight ranks are the driver's.)"""
    from geocalib_amd import LMOptimizer
    from geocalib_amd.parallel import RcclComm, SharedIntrinsicsSplit
    comm = RcclComm(RcclComm.unique_id(), 1, 0, 0)
    x = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3)
    assert torch.equal(comm.all_gather(x), x)
    y = x.clone()
    assert torch.equal(comm.all_reduce_sum_(y), x)
    conf, d = conf_for("shared_pinhole", "bench"), data_for("shared_pinhole", "bench")
    single = run(conf, d, dev)
    out = to_np(SharedIntrinsicsSplit(LMOptimizer(conf).eval(), 1, comm=comm)(to_dev(d, dev), torch.zeros(4, dtype=torch.int32)))
    assert np.array_equal(out["camera"], single["camera"]) and np.array_equal(out["gravity"], single["gravity"])
    torch.cuda.synchronize()


# Per-draw gates of the fuzz: [focal rel, gravity abs, distortion abs, final-cost rel] <= FUZZ_GATE + 10 x (what the
# yardstick itself moves on that draw).  north_star's 1e-4 on every quantity of every model; named exception:
#   simple_divisional distortion 5e-3    the reference's k-column cancels in float32 (camera.py:913)
FUZZ_GATE = {m: np.array([1e-4, 1e-4, 1e-4, 1e-4]) for m in ALL_MODELS}
FUZZ_GATE["simple_divisional"] = np.array([1e-4, 1e-4, 5e-3, 1e-4])
# Draws on which the yardstick moves by more than 1e-3 between float32 and float64 (oracle) or under a 1-ulp input
# perturbation (reference, simple_divisional) cannot gate anything: they only have to stay finite.  Their number is
# pinned for the committed default seed (so that a regression cannot hide in "undetermined"); other seeds: at most 30 %.
FUZZ_UNDETERMINED_2024 = {"pinhole": 1, "simple_radial": 1, "radial": 2, "simple_divisional": 8}


def yardstick_instability(oracle, data, conf, deviation=None, gate=None, hip=None):
    """Second opinion on a draw that missed its gate: is the comparison itself meaningful there?
    Returns a reason (str) or None.  Five diagnoses of the reference ALGORITHM in float32 (the oracle's own evaluation
    does not hold up on the draw) and three of the draw (`hip` = callable conf -> HIP result, for re-runs at other step
    counts):
      * its float32 and float64 trajectories part by more than 1e-3 at some step (the k-column of simple_divisional
        cancels for small |k|, camera.py:913: fuzz 19/198 -- at step 2 float32 and float64 differ by 6 % in the focal and
        0.1 in k; where an implementation lands there decides whether it then stalls for 20 steps);
      * past convergence its cost climbs back above its own minimum by more than 1e-3 (LM without step rejection,
        lm_optimizer.py:606-613: a cost that rises by rounding multiplies lambda by 10 (:95-106), the step shrinks, the
        (roll, pitch) parametrisation's 1e-4 guard (gravity.py:66) is no longer compensated and the estimate DRIFTS --
        fuzz 13/82: the oracle drifts from step 14 on, the reference itself and the HIP path stay put);
      * ... or would within 8 more steps: the fixed point it sits on is unstable, who leaves it first is rounding
        (fuzz 25/211, 42/0: the HIP path starts to drift at step 9, the oracle at step 12);
      * its cost has converged and its iterates keep moving (a limit cycle, fuzz 20/136);
      * eight MORE 1-ulp perturbations of its input move it by enough to cover the deviation (the first pass tries two:
        a lambda flip in one image of the batch is a coin toss per perturbation);
      * the draw had CONVERGED and the HIP path matched the oracle within the gate at the step the oracle's cost stopped
        moving; what follows is the HIP path's own post-convergence drift (the mirror image of fuzz 13/82 -- fuzz 25/211,
        42/0: (roll, pitch) parametrisation, 6-9 more steps without step rejection, the oracle happens to sit on an exact
        floating-point fixed point, the HIP path's costs still flicker in their last bit, lambda runs away, gravity.py:66's
        guard is no longer compensated);
      * the same with early_stop on: an image that has converged keeps taking steps until the batch-global stop fires; the HIP
        path matched the oracle within the gate at an earlier step, the oracle's cost on the deviating images did not improve
        from there to the stop, and the HIP path's final cost there is no higher (fuzz 307/137, 309/232);
      * only the COST is beyond its gate, on a draw that has not converged: the cost is first-order in the parameters
        there, and the oracle's own cost still moves by more than the deviation per step (fuzz 31/208: parameters within
        1e-5, one image of five still descending at step 7)."""
    t32 = oracle.solve(data, conf, precision="f32", trace=True)
    t64 = oracle.solve(data, conf, precision="f64", trace=True)
    a, b = t32["trace"], t64["trace"]
    n = min(int((a["lambda"][:, 0] > 0).sum()), int((b["lambda"][:, 0] > 0).sum()))
    if n > 0:
        f = np.abs(a["cam"][:n, :, :2] / b["cam"][:n, :, :2] - 1).max(axis=(1, 2))
        k = np.abs(a["cam"][:n, :, 2:] - b["cam"][:n, :, 2:]).max(axis=(1, 2))
        g = np.abs(a["gravity"][:n] - b["gravity"][:n]).max(axis=(1, 2))
        worst = np.maximum(np.maximum(f, k), g)
        if worst.max() > 1e-3:
            return f"float32 / float64 trajectories of the oracle differ by {worst.max():.1e} after step {int(worst.argmax()) + 1}"
        cost = np.concatenate([(a["cost_up"] + a["cost_lat"])[:n], t32["final_cost"][None]], 0)
        floor = 1e-3 * np.abs(t32["initial_cost"]).max()
        drift = ((t32["final_cost"] - cost.min(0)) / np.maximum(cost.min(0), floor)).max()
        if drift > 1e-3:
            return f"the oracle's final cost sits {drift:.1e} above its own minimum (post-convergence drift)"
        converged = t32["stop_at"][0] < n
        if n >= 2 and converged:
            # its cost has converged (the allclose test passed at stop_at) and yet its iterates keep moving: a limit
            # cycle (fuzz 20/136: radial on a latitude-only draw, k2 pinned at its 0.7 clamp (camera.py:700), k1 alternating
            # between -0.2417 and -0.2440 from step to step) -- which phase an implementation ends in is one step's luck
            last = max(np.abs(a["cam"][n - 1, :, :2] / a["cam"][n - 2, :, :2] - 1).max(),
                       np.abs(a["cam"][n - 1, :, 2:] - a["cam"][n - 2, :, 2:]).max(),
                       np.abs(a["gravity"][n - 1] - a["gravity"][n - 2]).max())
            if last > 1e-4:
                return f"the oracle's iterates still move by {last:.1e} per step at a converged cost (limit cycle)"
        if converged and not conf["early_stop"]:
            from conftest import result_spread
            more = oracle.solve(data, {**conf, "num_steps": conf["num_steps"] + 8}, precision="f32")
            away = result_spread(more, t32).max()
            if away > 1e-3:
                return f"8 more steps move the oracle by {away:.1e} from its converged estimate (unstable fixed point, drift onset)"
    if deviation is not None:
        from conftest import perturbed, result_spread
        deviation, gate = np.asarray(deviation), np.asarray(gate)
        prng = np.random.default_rng([int(np.abs(t32["final_cost"]).sum() * 1e12) % (1 << 31), 11])
        own8 = np.zeros(4)
        for _ in range(8):
            own8 = np.maximum(own8, result_spread(oracle.solve(perturbed(data, prng), conf, precision="f32"), t32))
        if (deviation < gate + 10.0 * own8.max()).all():
            return f"eight more 1-ulp perturbations of its input move the oracle by {own8.max():.1e}"
        stop = int(t32["stop_at"][0])
        if hip is not None and n > 0 and stop < n and not conf["early_stop"]:
            # `stop_at` = the number of updates after which the cost was first "close" (lm_optimizer.py:619-620): the state a
            # default-conf solve (early_stop) would have returned.  The HIP path's own drift may set in right there (fuzz 91/6:
            # radial, (roll, pitch), noise-free; costs 5.3561e-10 / 5.3566e-10 at step 4, then every HIP cost a last bit ABOVE
            # the one before -- lambda x 10 per step instead of the oracle's x 10 / x 0.1 two-cycle -- gravity 2.5e-6 off at
            # step 7, 1.5e-5 at step 8, 1.2e-4 at step 12), so the match is looked for at that step and at the one after it
            for steps in (stop, stop + 1):
                if steps < 1:
                    continue
                at = {**conf, "num_steps": steps}
                d_at = result_spread(hip(at), oracle.solve(data, at, precision="f32"))
                if (d_at < gate).all():
                    return (f"matched the oracle within the gate ({d_at.max():.1e}) after step {steps}, where the oracle's cost had "
                            f"converged; then drifted over the remaining {n - steps} steps (no step rejection)")
        if hip is not None and n >= 2 and conf["early_stop"] and stop >= n:
            # The same drift BEFORE a stop: the stop is batch-global (lm_optimizer.py:619-625), so an image that has converged
            # keeps taking steps until the last image of its batch has -- steps that no longer lower its cost but, without step
            # rejection, still move it along a flat direction (radial's k2 on a small image).  fuzz 307/137: image 2 at its
            # minimum from step 11 on, the oracle's k2 moves 2.1e-4 in step 13 at a cost 1e-6 higher, the HIP path's 4.5e-5;
            # fuzz 309/232: image 1 at its minimum after step 7, the oracle's float32 AND float64 costs creep up 3e-7 per step,
            # lambda x 10 each time, k2 4.2e-4 away at the stop (step 16, cost +3.4e-4), the HIP path's costs flicker in their
            # last bits, its lambda goes up and down and it stays put (profiles/r06_fuzz_307_309_trace.log).  Excused only if
            # the two matched within the gate at an earlier step k AND the oracle's cost on every deviating image did not
            # improve from step k to the stop (what followed was no descent) AND the HIP path's final cost there is no higher.
            h_end, o_end = hip(conf), t32
            per_image = np.stack([np.abs(h_end["camera"][:, 2:4] / o_end["camera"][:, 2:4] - 1).max(1),
                                  np.abs(h_end["gravity"] - o_end["gravity"]).max(1),
                                  np.abs(h_end["camera"][:, 6:] - o_end["camera"][:, 6:]).max(1) if o_end["camera"].shape[1] > 6
                                  else np.zeros(len(o_end["camera"]))], 1)
            bad = (per_image >= gate[None, :3]).any(1)
            cost = np.concatenate([(a["cost_up"] + a["cost_lat"])[:n], t32["final_cost"][None]], 0)       # cost[i] = after i steps
            if bad.any() and (h_end["final_cost"][bad] <= t32["final_cost"][bad] * (1 + 1e-5)).all():
                for k in range(n - 1, max(0, n - 11), -1):
                    if not (cost[k, bad] <= cost[n, bad] * (1 + 1e-6)).all():
                        break                                            # the oracle was still descending there: not this
                    at = {**conf, "num_steps": k, "early_stop": False}
                    d_at = result_spread(hip(at), oracle.solve(data, at, precision="f32"))
                    if (d_at < gate).all():
                        return (f"matched the oracle within the gate ({d_at.max():.1e}) after step {k}; over the remaining {n - k} steps to the "
                                f"batch-global stop the oracle's cost on the deviating image(s) {np.flatnonzero(bad).tolist()} did not improve "
                                f"(x {float((cost[n, bad] / cost[k, bad]).max()):.7f}): post-convergence drift along a flat direction, no step rejection")
        if n >= 2 and stop >= n and (deviation[:3] < gate[:3] + 10.0 * own8.max()).all():
            cost = np.concatenate([(a["cost_up"] + a["cost_lat"])[:n], t32["final_cost"][None]], 0)
            moving = (np.abs(cost[-1] - cost[-2]) / np.maximum(np.abs(t32["final_cost"]).max(), 1e-30)).max()
            if moving > deviation[3]:
                return (f"only the cost is beyond its gate ({deviation[3]:.1e}) on an unconverged draw whose cost still moves by "
                        f"{moving:.1e} per step")
    return None


def test_randomised_configurations_against_oracle(dev, oracle):
    """Seeded fuzz: random shapes (vector and scalar paths), batch sizes, ALL FOUR camera models, conf knobs, missing
    confidences / up field, priors and scales -- the HIP path against the oracle on identical inputs, and for
    `simple_divisional` against the REFERENCE's own result on that draw (tests/golden/make_golden_div.py: seeds 2024 and
    11..22), gated by the yardstick's own reproducibility.  GCLM_FUZZ_SEED / GCLM_FUZZ_CASES: the soak
    (scripts/fuzz_soak.sh -> profiles/archive/r03_fuzz_soak.txt)."""
    from conftest import MEASURED, fuzz_draws, perturbed, result_spread
    seed = int(os.environ.get("GCLM_FUZZ_SEED", "2024"))                             # soak: GCLM_FUZZ_CASES=300
    n_cases, n_models = int(os.environ.get("GCLM_FUZZ_CASES", "80")), int(os.environ.get("GCLM_FUZZ_MODELS", "4"))
    div_path = os.path.join(GOLDEN, "golden_div_fuzz.npz")
    div = np.load(div_path) if n_models == 4 and os.path.exists(div_path) else None
    worst, against_reference, spent, failures, stop_shifts, excused = {}, 0, np.zeros(2), [], [], []
    undetermined = {m: 0 for m in ALL_MODELS}
    first_step = {m: 0 for m in ALL_MODELS}     # of the undetermined draws: compared after ONE LM step instead
    drawn = {m: 0 for m in ALL_MODELS}
    for case, model, (H, W), B, data, conf, cams, gravs in fuzz_draws(seed, n_cases, n_models):
        t0 = time.perf_counter()
        ref = oracle.solve(data, conf, precision="f32")
        ref64 = oracle.solve(data, conf, precision="f64")
        t1 = time.perf_counter()
        out = run(conf, data, dev)
        spent += np.array([t1 - t0, time.perf_counter() - t1])
        drawn[model] += 1
        assert np.array_equal(out["camera"][:, [0, 1, 4, 5]], ref["camera"][:, [0, 1, 4, 5]])
        assert all(np.isfinite(out[k]).all() for k in ("camera", "gravity", "final_cost")), (case, model)
        # unconverged / ill-conditioned draws (few steps, tiny images at the focal clamp, radial k2 on a sliver of an
        # image, noise-free costs ~1e-8) amplify rounding chaotically: where the yardstick itself moves by more than
        # 1e-3 the draw only has to stay finite; elsewhere the gate is tight (plus what the yardstick moves).
        # "Moves" = the larger of float32-vs-float64 of the same algorithm and float32 under two 1-ulp perturbations of
        # its input (the damping rule and the batch-global stop are discontinuous in the last bit of the costs).
        prng = np.random.default_rng([seed, case, 7])
        own = result_spread(ref, ref64)
        for _ in range(2):
            own = np.maximum(own, result_spread(oracle.solve(perturbed(data, prng), conf, precision="f32"), ref))
        yard, gate, kind = ref, FUZZ_GATE[model], "oracle"
        if div is not None and model == "simple_divisional" and f"{seed}/{case}/camera" in div.files:
            # the reference's float32 result is the yardstick; it is only as sharp as the reference is reproducible
            # (1-ulp input perturbations) and as its cancelling formulas are accurate in float32 at this draw (the
            # same algorithm in float64 -- a different operation order moves a float32 implementation that far)
            yard = {k: div[f"{seed}/{case}/{k}"] for k in ("camera", "gravity", "final_cost", "initial_cost", "stop_at")}
            own = np.maximum(div[f"{seed}/{case}/spread"], own)
            kind = "reference"
        elif model == "simple_divisional":
            # no reference golden for this draw (another seed, or siclib knobs the inference optimiser does not have):
            # the oracle is the only yardstick.  It follows the same cancelling formulas in another operation order, so
            # it can leave (or stay in) a stall the reference and the HIP path share (fuzz 11/35): a
            # LOOSE gate -- wide enough for that, tight enough to catch a broken kernel -- instead of none
            gate, kind = np.array([2e-3, 2e-3, 5e-3, 2e-3]), "oracle-loose"
        if conf["early_stop"] and not np.array_equal(out["stop_at"], yard["stop_at"]):
            # The batch-global stop is torch.allclose on costs that agree to their last bits (lm_optimizer.py:90-92,
            # 619-625): WHICH step it fires at is rounding noise, and rtol = 1e-8 on a locally quadratic cost pins the
            # parameters only to ~sqrt(1e-8) = 1e-4.  A draw on which the HIP path stops at a neighbouring step is
            # compared with the yardstick's trajectory AT THAT STEP (the oracle run for exactly that many steps).
            # (how far apart the two stops are says nothing: a cost creeping along a flat valley crosses the 1e-8
            # threshold wherever its last bits let it -- seed 11 case 44, pinhole: step 16 vs 13 -- so the step itself is
            # only asserted where the reference's goldens pin it, test_hip_matches_reference_small / _full_size)
            hip_stop = int(out["stop_at"][0])
            stop_shifts.append(hip_stop - int(np.asarray(yard["stop_at"]).ravel()[0]))
            at = {**conf, "num_steps": hip_stop, "early_stop": False}
            yard = oracle.solve(data, at, precision="f32")
            own = result_spread(yard, oracle.solve(data, at, precision="f64"))
            for _ in range(2):
                own = np.maximum(own, result_spread(oracle.solve(perturbed(data, prng), at, precision="f32"), yard))
            if model == "simple_divisional":
                gate = np.array([2e-3, 2e-3, 5e-3, 2e-3])
            kind = ("oracle-loose" if model == "simple_divisional" else "oracle") + "@hip-stop"
        rec = {"model": model, "yardstick": kind, "own": own.tolist()}
        if own.max() > 1e-3:
            # The END of this draw's trajectory is not a yardstick -- but its BEGINNING is: rounding noise needs steps to be
            # amplified.  One LM step from the same start (no early stop) is compared instead, under the same rule (the
            # oracle in float32 against itself in float64 and under two 1-ulp perturbations decides whether even that
            # is sharp); only a draw whose FIRST step is not reproducible is left at "finite".  
            undetermined[model] += 1
            at1 = {**conf, "num_steps": 1, "early_stop": False}
            y1 = oracle.solve(data, at1, precision="f32")
            own1 = result_spread(y1, oracle.solve(data, at1, precision="f64"))
            for _ in range(2):
                own1 = np.maximum(own1, result_spread(oracle.solve(perturbed(data, prng), at1, precision="f32"), y1))
            rec = {**rec, "undetermined": True, "own_first_step": own1.tolist()}
            if own1.max() <= 1e-3:
                d1 = result_spread(run(at1, data, dev), y1)
                tol1 = FUZZ_GATE[model] + 10.0 * own1.max()
                first_step[model] += 1
                rec |= {"first_step_spread": d1.tolist(), "first_step_tol": tol1.tolist()}
                if not (d1 < tol1).all():
                    failures.append((case, model, (H, W), B, at1, d1.tolist(), tol1.tolist(), "first LM step of an undetermined draw"))
                    rec["FAILED"] = True
            MEASURED[f"fuzz/{seed}/{case}"] = rec
            continue
        against_reference += kind == "reference"
        worst[case] = result_spread(out, yard)
        tol = gate + 10.0 * own.max()          # one sensitivity per draw: a draw that is touchy in one quantity is in all
        MEASURED[f"fuzz/{seed}/{case}"] = {**rec, "spread": worst[case].tolist(), "tol": tol.tolist()}
        if not (worst[case] < tol).all():
            # Beyond its gate.  Before this counts as a parity failure the yardstick gets a second, stronger examination
            # (only here: it costs two traced solves) -- is ITS OWN float32 evaluation trustworthy on this draw?
            why = yardstick_instability(oracle, data, conf if "@hip-stop" not in kind else at, worst[case], gate,
                                        hip=lambda c: run(c, data, dev))
            rec2 = (case, model, (H, W), B, conf, worst[case].tolist(), tol.tolist(), why)
            (excused if why else failures).append(rec2)
            MEASURED[f"fuzz/{seed}/{case}"]["excused" if why else "FAILED"] = why or True
    w = np.array(list(worst.values()))
    compared = {m: round(1.0 - undetermined[m] / drawn[m], 3) for m in ALL_MODELS if drawn[m]}
    compared_any = {m: round(1.0 - (undetermined[m] - first_step[m]) / drawn[m], 3) for m in ALL_MODELS if drawn[m]}
    MEASURED[f"fuzz_summary/{seed}"] = {"draws": drawn, "undetermined": undetermined, "undetermined_compared_at_first_step": first_step,
                                        "fraction_compared_at_the_end": compared, "fraction_compared_at_all": compared_any}
    print(f"fuzz seed {seed}: fraction of the draws compared with the yardstick at the END of the solve {compared}, "
          f"at the end or (undetermined there) after the first step {compared_any}")
    print(f"fuzz seed {seed}: {n_cases} draws {drawn}, undetermined {undetermined}, {against_reference} simple_divisional "
          f"draws gated by the reference, {len(stop_shifts)} compared at the HIP path's stop step (shifts {sorted(stop_shifts)}), {len(failures)} beyond their gate, {len(excused)} beyond it on an unstable yardstick {[(e[0], e[1], e[-1]) for e in excused]}, median spread {np.median(w, axis=0)}, worst "
          f"{w.max(axis=0)}, seconds in the oracle {spent[0]:.1f} / in the HIP path {spent[1]:.1f}")
    assert not failures, failures
    assert len(excused) <= max(1, n_cases // 100), excused       # at most 1 % of the draws (soak: 0-2 of 300 per seed)
    if seed == 2024 and n_cases == 80 and n_models == 4:
        assert undetermined == FUZZ_UNDETERMINED_2024 and not excused, (undetermined, excused)
    assert sum(undetermined.values()) <= 0.3 * n_cases, undetermined
    for m in ALL_MODELS:          # no model's fuzz may degenerate into "finite only" (VERDICT r04 #4b); small samples: 80-draw runs
        if drawn[m] >= 10:
            # at the END of the solve: 0.52 ... 0.76 of the simple_divisional draws over seeds 11-30 (0.556 for the committed seed,
            # pinned by FUZZ_UNDETERMINED_2024 above); at the end OR after the first step: 0.97 ... 1.0 of every model's
            assert compared[m] >= (0.5 if seed == 2024 else 0.4), (m, compared, drawn, undetermined)
            assert compared_any[m] >= 0.8, (m, compared_any, drawn, undetermined, first_step)
    if div is not None and seed in (2024, *range(11, 23)):
        assert against_reference >= 5, against_reference      # simple_divisional really was drawn and TIGHTLY gated by the reference
    med = np.median(w, axis=0)
    assert med[0] < 2e-5 and med[1] < 2e-5 and med[3] < 2e-5, med


@pytest.mark.parametrize("shape", [((2, 2, 24, 32), (61, 83)), ((3, 17, 23), (17, 23)), ((1, 1, 32, 48), (480, 720)), ((2, 40, 30), (20, 15)),
                                   ((3, 2, 240, 320), (480, 640)),      # x2: float4 tiles, the wide-load strip (320 px = 1.25 strips)
                                   ((2, 320, 416), (768, 1000)),        # x2.4 in both axes, a last strip with idle lanes
                                   ((2, 1, 90, 120), (100, 160)),       # x1.33 in x: tiled, but the four-float window does not hold
                                   ((1, 2, 37, 52), (111, 208)),        # x4 in x, x3 in y
                                   ((2, 7, 3), (20, 12)), ((1, 5, 4), (9, 8)),      # sources narrower than / exactly one window
                                   ((2, 64, 80), (32, 40)),             # downsampling through the float4 path
                                   ((11, 5, 240, 320), (480, 640)),     # rows of whole lines, 8 rows per wave, source rows prefetched
                                   ((2, 5, 320, 480), (1080, 1620)),    # ragged rows (6480 B): phase-rotated units, 8 rows per wave
                                   ((1, 5, 320, 480), (1080, 1620)),    # ... one image: 4 rows per wave
                                   ((3, 9, 12), (27, 36)),              # ragged rows AND ragged planes (3888 B): a phase per plane
                                   ((2, 100, 64), (120, 160)),          # window in x, x1.2 in y: no prefetch of the source rows
                                   ((1, 8, 2048), (8, 3072)),           # exactly x1.5 on a wide row: the window path
                                   ((1, 8, 2730), (16, 4096)),          # a hair under x1.5 at 4096 wide: fp32 cannot promise the 4-float window (the 5-float one holds)
                                   ((2, 5, 320, 448), (400, 560)),      # x1.25: five-float window, rows of 17.5 lines (2240 B: half lines, no rotation)
                                   ((2, 5, 320, 448), (400, 564)),      # x1.26: five-float window, ragged rows (2256 B): phase-rotated
                                   ((9, 5, 300, 400), (375, 512)),      # x1.28 / x1.25: five-float window, whole lines, ROWS + 2 source rows prefetched
                                   ((2, 300, 400), (375, 512)),         # ... one image's worth: 4 rows per wave
                                   ((2, 48, 64), (48, 64)),             # the identity resize (exact copy)
                                   ((2, 100, 64), (50, 160)),           # window in x, DOWNsampling in y: source rows loaded as the walk reaches them
                                   ((1, 6, 7), (7, 8)),                 # a source row of barely one five-float window
                                   ((1, 4, 4095), (4, 4096))])          # a hair under x1 at 4096 wide: per-lane gathers
def test_upsample_fields_matches_torch_interpolate(dev, shape):
    """gclm_upsample_fields (GeoCalib._post_process, extractor.py:60-63) against F.interpolate bilinear."""
    from geocalib_amd.fields import upsample_fields
    src_shape, size = shape
    x = torch.randn(*src_shape, generator=torch.Generator().manual_seed(1)).to(dev)
    ref = torch.nn.functional.interpolate(x if x.dim() == 4 else x[:, None], size=size, mode="bilinear")
    ref = ref if x.dim() == 4 else ref[:, 0]
    out = upsample_fields(x, size)
    assert out.shape == ref.shape
    assert torch.allclose(out, ref, atol=2e-6, rtol=1e-6), (out - ref).abs().max().item()
    if tuple(x.shape[-2:]) == tuple(size):
        assert torch.equal(out, x)


def test_upsample_paths_agree_bitwise(dev):
    """The scalar, gather, consecutive-row and phase-rotated kernels of gclm_upsample_fields evaluate the same roundings per
    output value (csrc/gclm_update.hip: up_lerp): a 4-byte shifted destination forces the scalar kernel, which every float4
    path must reproduce bit for bit."""
    from geocalib_amd import _lib
    lib = _lib.load()
    for (planes, h, w), (H, W) in (((10, 240, 320), (480, 640)), ((56, 240, 320), (480, 640)), ((3, 320, 480), (1080, 1620)),
                                   ((11, 320, 480), (1080, 1620)), ((3, 90, 120), (100, 160)), ((2, 100, 64), (120, 160)),
                                   ((4, 9, 12), (27, 36)), ((1, 8, 2730), (16, 4096)), ((1, 8, 2048), (8, 3072)),
                                   ((10, 320, 448), (400, 560)), ((3, 320, 448), (400, 560)), ((45, 300, 400), (375, 512)),
                                   ((10, 320, 448), (400, 564)), ((3, 320, 448), (400, 564)),
                                   ((3, 300, 400), (375, 512)), ((2, 48, 64), (48, 64)), ((2, 100, 64), (50, 160)), ((1, 4, 4095), (4, 4096))):
        x = torch.randn(planes, h, w, generator=torch.Generator().manual_seed(planes)).to(dev)
        vec = torch.full((planes * H * W,), float("nan"), device=dev)
        sca = torch.full((planes * H * W + 4,), float("nan"), device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        assert lib.gclm_upsample_fields(x.data_ptr(), planes, h, w, H, W, vec.data_ptr(), stream) == 0
        assert lib.gclm_upsample_fields(x.data_ptr(), planes, h, w, H, W, sca.data_ptr() + 4, stream) == 0
        torch.cuda.synchronize()
        assert torch.isnan(sca[0]) and torch.isnan(sca[-3:]).all()
        assert torch.equal(vec, sca[1:-3]), ((planes, h, w), (H, W), (vec - sca[1:-3]).abs().max().item())


def test_upsample_random_shapes_agree_with_the_scalar_kernel(dev):
    """Seeded fuzz over shapes: every float4 kernel the plan can pick (windows of 4 / 5 floats, consecutive / phase-rotated
    rows, 4 / 8 rows per wave, three prefetch depths, gathers) against the scalar kernel bit for bit, and against
    F.interpolate at 2e-6.  Ratios from 0.4 to 6, widths around the strip (256 px) and line (32 px) boundaries, 1-40 planes."""
    import torch.nn.functional as F
    from geocalib_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(20260924)
    stream = torch.cuda.current_stream(dev).cuda_stream
    seen = set()
    for it in range(160):
        W = int(rng.choice([8, 12, 36, 64, 100, 128, 252, 256, 260, 516, 640, 1000, 1620])) if it % 3 else 4 * int(rng.integers(1, 420))
        H = int(rng.integers(1, 70))
        rx, ry = float(rng.choice([0.4, 0.9, 1.0, 1.1, 1.25, 1.49, 1.5, 1.51, 2.0, 3.375, 6.0])), float(rng.choice([0.5, 1.0, 1.2, 1.5, 2.0, 4.0]))
        w, h = max(1, int(round(W / rx))), max(1, int(round(H / ry)))
        planes = int(rng.choice([1, 2, 5, 40])) if W * H < 20000 else int(rng.choice([1, 5]))
        if rng.random() < 0.1:
            planes = 1 + int(16.0e6 // (H * W))               # past the "small job" cut: 8 rows per wave
        x = torch.randn(planes, h, w, generator=torch.Generator().manual_seed(it)).to(dev)
        vec = torch.full((planes * H * W,), float("nan"), device=dev)
        sca = torch.full((planes * H * W + 4,), float("nan"), device=dev)
        assert lib.gclm_upsample_fields(x.data_ptr(), planes, h, w, H, W, vec.data_ptr(), stream) == 0
        assert lib.gclm_upsample_fields(x.data_ptr(), planes, h, w, H, W, sca.data_ptr() + 4, stream) == 0
        assert torch.equal(vec, sca[1:-3]), ((planes, h, w), (H, W), (vec - sca[1:-3]).abs().max().item())
        ref = F.interpolate(x[:, None], size=(H, W), mode="bilinear")[:, 0].reshape(-1)
        assert torch.allclose(vec, ref, atol=2e-6, rtol=1e-6), ((planes, h, w), (H, W), (vec - ref).abs().max().item())
        seen.add((w >= 4 and 3 * w <= 2 * W, w >= 5 and w <= W, (W // 4) % 4 == 0, 3 * h <= 2 * H, h <= H, planes * H * W >= 16.0e6))
    assert len(seen) >= 20, len(seen)                       # the draws really spread over the plan's branches


def test_upsample_more_planes_than_a_grid_dimension(dev):
    """Beyond 65 535 planes the launch strides over them (grid.z of the float4 kernels, grid.y of the scalar one); the
    multi-tensor entry point no longer has a plane limit."""
    import torch.nn.functional as F
    from geocalib_amd.fields import upsample_fields, upsample_fields_multi
    x = torch.randn(70001, 4, 8, generator=torch.Generator().manual_seed(3)).to(dev)
    for size in ((8, 16), (7, 15)):                     # float4 window path / scalar path (15 is not a multiple of 4)
        ref = F.interpolate(x[:, None], size=size, mode="bilinear")[:, 0]
        out = upsample_fields(x, size)
        assert torch.allclose(out, ref, atol=2e-6, rtol=1e-6), (size, (out - ref).abs().max().item())
    a, b = upsample_fields_multi([x[:40000], x[40000:]], (8, 16))
    assert torch.equal(torch.cat([a, b]), upsample_fields(x, (8, 16)))


def test_upsample_multi_ragged_planes(dev):
    """gclm_upsample_fields_multi on tensors whose planes are not whole 128-byte lines: the store rotation takes its phase
    from every plane's own address."""
    import torch.nn.functional as F
    from geocalib_amd.fields import upsample_fields_multi
    g = torch.Generator().manual_seed(5)
    for (h, w), size in (((9, 12), (27, 36)), ((320, 480), (1080, 1620)), ((240, 320), (480, 640))):
        ts = [torch.randn(2, 2, h, w, generator=g).to(dev), torch.randn(2, h, w, generator=g).to(dev),
              torch.randn(2, 1, h, w, generator=g).to(dev), torch.randn(2, h, w, generator=g).to(dev)]
        outs = upsample_fields_multi(ts, size)
        for t, o in zip(ts, outs):
            ref = F.interpolate(t if t.dim() == 4 else t[:, None], size=size, mode="bilinear")
            ref = ref if t.dim() == 4 else ref[:, 0]
            assert o.shape == ref.shape and torch.allclose(o, ref, atol=2e-6, rtol=1e-6), (o - ref).abs().max().item()


def test_calibrate_front_end(dev):
    """GeoCalib.calibrate (extractor.py:72-127) with a stand-in field network: preprocessing bookkeeping,
    LM on the HIP path, undo of scale / crop, fields resized back to the input resolution."""
    from geocalib_amd import Gravity, camera_models, perspective_fields as pf
    from geocalib_amd.extractor import GeoCalib
    H0, W0 = 480, 700                                        # input image; preprocess -> 320 x 448 fields
    cam = camera_models["pinhole"].from_dict({"height": torch.tensor([320.0]), "width": torch.tensor([448.0]),
                                              "vfov": torch.tensor([0.9])})
    grav = Gravity.from_rp(torch.tensor([0.15]), torch.tensor([-0.2]))

    def field_model(img_data):
        assert img_data["image"].shape[-2:] == (320, 448)
        up, lat = pf.get_perspective_field(cam, grav)
        ones = torch.ones(1, 320, 448)
        return {"up_field": up.to(dev), "latitude_field": lat.to(dev), "up_confidence": ones.to(dev),
                "latitude_confidence": ones.to(dev)}

    model = GeoCalib(field_model)            # as constructed, like the reference (extractor.py:43 evals its model): uncertainties on
    assert not model.optimizer.training
    res = model.calibrate(torch.rand(3, H0, W0, device=dev))
    assert {"roll_uncertainty", "pitch_uncertainty", "gravity_uncertainty", "focal_uncertainty", "vfov_uncertainty"} <= set(res)
    assert "covariance" not in model.train().calibrate(torch.rand(3, H0, W0, device=dev))     # .train() reaches the optimiser
    model.eval()
    assert res["up_field"].shape == (1, 2, H0, W0) and res["latitude_confidence"].shape == (1, H0, W0)
    c = res["camera"]
    assert c.size[0].tolist() == pytest.approx([W0, H0], abs=1e-3)
    # the vertical field of view is invariant to the resize; the crop trims the width only
    assert c.vfov.item() == pytest.approx(0.9, abs=2e-3)
    assert torch.allclose(res["gravity"].vec3d.cpu(), grav.vec3d, atol=2e-3)
    assert res["covariance"].shape == (1, 3, 3) and res["focal_uncertainty"].shape == (1,)


def test_post_process_fast_path_equals_the_reference_route(dev):
    """GeoCalib._post_process with the host-side bookkeeping of default_preprocess (one fused multiply-add on the camera
    rows, ONE bilinear launch for the four tensors, no device-to-host read) against the reference's route
    (`undo_scale_crop`, size read back from the camera, one interpolate per tensor): same bits for the camera and the
    focal uncertainty, F.interpolate's values for the fields; several images (shared intrinsics) as well."""
    import torch.nn.functional as F
    from geocalib_amd import GeoCalib, camera_models
    from geocalib_amd.extractor import default_preprocess
    for B, (H0, W0) in ((1, (768, 1024)), (3, (480, 700)), (1, (333, 517))):
        img = torch.rand(B, 3, H0, W0, device=dev)
        data = default_preprocess(img)
        h, w = data["image"].shape[-2:]
        cam = camera_models["simple_radial"](torch.tensor([[w, h, 300.0, 310.0, w / 2 + 1.5, h / 2 - 0.75, -0.1, -0.1]], device=dev).repeat(B, 1))
        fields = {"up_field": torch.randn(B, 2, h, w, device=dev), "latitude_field": torch.randn(B, 1, h, w, device=dev),
                  "up_confidence": torch.rand(B, h, w, device=dev), "latitude_confidence": torch.rand(B, h, w, device=dev),
                  "focal_uncertainty": torch.rand(B, device=dev)}
        model = GeoCalib(lambda d: {})
        c1, o1 = model._post_process(cam, data, dict(fields))
        c2, o2 = model._post_process(cam, {k: v for k, v in data.items() if k != "_host"}, dict(fields))
        assert torch.equal(c1._data, c2._data) and torch.equal(o1["focal_uncertainty"], o2["focal_uncertainty"])
        assert c1.size[0].tolist() == pytest.approx([W0, H0], abs=1e-3)
        for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence"):
            assert o1[k].shape == o2[k].shape == fields[k].shape[:-2] + (H0, W0) and torch.equal(o1[k], o2[k])
            src = fields[k] if fields[k].dim() == 4 else fields[k][:, None]
            ref = F.interpolate(src, size=(H0, W0), mode="bilinear", align_corners=False).reshape(o1[k].shape)
            assert torch.allclose(o1[k], ref, atol=2e-6, rtol=1e-6), k


def test_hip_matches_reference_shared_radial(dev):
    """Shared intrinsics with the 5-parameter radial model: 3x3 Schur complement vs the reference's dense solve."""
    g = np.load(os.path.join(GOLDEN, "golden_extra.npz"))
    ref = {k.split("/", 2)[2]: g[k] for k in g.files if k.startswith("radial/shared/")}
    conf = {"camera_model": "radial", "shared_intrinsics": True, "num_steps": 20, "early_stop": False}
    out = run(conf, data_for("radial", "bench"), dev)
    compare_result(out, ref, {**TOL, "cost": 5e-4, "unc": 5e-3, "cov": 5e-3}, "radial/shared")
    assert np.abs(out["camera"][:, 6:] - out["camera"][0, 6:]).max() < 1e-6      # one (k1, k2) for the group


def test_c_abi_from_plain_c(dev, tmp_path):
    """examples/calibrate_c_abi.c: the library driven from C99 (gcc, no Python / torch in the process) must give
    the same calibration as the Python host path on the same device-generated fields."""
    import re
    import subprocess
    from conftest import ROOT
    from geocalib_amd import LMOptimizer
    exe = str(tmp_path / "calibrate_c_abi")
    lib = os.path.join(ROOT, "geocalib_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "calibrate_c_abi.c"),
                    "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-L" + lib, "-lgeocalib_hip",
                    "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + lib, "-o", exe],
                   check=True, capture_output=True)
    for model_id, model in ((0, "pinhole"), (1, "simple_radial")):
        txt = subprocess.run([exe, str(model_id), "3", "120", "160"], check=True, capture_output=True, text=True).stdout
        rows = re.findall(r"image \d+: f (\S+) \(gt (\S+)\) k1 (\S+) .*? g \((\S+) (\S+) (\S+)\)", txt)
        assert len(rows) == 3, txt
        data, gtc, _ = synth_device(model, 3, 120, 160, dev, seed=42)
        out = to_np(LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()(data))
        got = np.array(rows, dtype=np.float64)
        assert np.allclose(got[:, 0], out["camera"][:, 3], rtol=2e-6), (got[:, 0], out["camera"][:, 3])
        assert np.allclose(got[:, 1], gtc[:, 3].cpu().numpy(), rtol=2e-6)
        assert np.allclose(got[:, 2], out["camera"][:, 6], atol=2e-5)
        assert np.allclose(got[:, 3:6], out["gravity"], atol=2e-5)


@pytest.mark.parametrize("extra", [[], ["--shared-group", "16"], ["--shared-group", "16", "--shared-by-group"],
                                   ["--shared-group", "16", "--virtual-world", "8"]])
def test_bench_multi_rank_path_on_one_gpu(dev, extra):
    """bench.py's N>1 code path (image sharding + ONE gather; shared-intrinsics frame split + ONE all-reduce per
    step) with two real processes that share this GPU and talk over gloo: the JSON line must describe the
    whole job and the solve must still recover the ground truth (bench.py asserts that itself)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--batch", "64", "--steps", "2", "--warmup", "1", "--cpu-sample", "0"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["unit"] == "images/sec" and out["steps"] == 2
    assert out["check"]["median_focal_rel_err_vs_gt"] < 5e-3
    assert "cpu_baseline" not in out and out["roofline"]["launches_timed"] == out["repeats"] * 2 * 21
    # the N>1 line is attributable: ranks seen by the communicator, every rank's own step time, time in the collective
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == 2 and len(mg["per_rank_ms"]) == 2 and all(t > 0 for t in mg["per_rank_ms"])
    split = "--shared-group" in extra and "--shared-by-group" not in extra
    assert mg["collective_ms"] >= 0 and mg["collectives_per_step"] == (20 if split else 1)
    if "--virtual-world" in extra:      # two real ranks with the per-rank shape of an 8-GPU run: 2 frames of each of 32 groups
        assert mg["virtual_world"] == 8 and mg["collective_bytes"] == 32 * 32 * 4
        assert "per-rank shape of a 8-GPU run" in out["config"]["workload"]
    elif split:
        assert mg["collective_bytes"] == 8 * 32 * 4                      # 128 frames = 8 groups of 16, split over the 2 ranks
    assert len(out["ms_per_step_repeats"]) == out["repeats"] == 3
    # every N > 1 line carries a parity check of its own: the rows the gather delivered for the OTHER rank are
    # bit for bit a one-GPU solve of that rank's images; a frame split lands within 1e-4 of whole groups solved on rank 0
    par = mg["parity"]
    if "--virtual-world" in extra:
        assert "skipped" in par
    elif split:
        assert par["groups_checked"] == 4 and par["frames_compared"] == 32 and par["within_gate"] is True, par
        assert max(par["max_focal_rel"], par["max_gravity_abs"], par["max_final_cost_rel"]) < 1e-4
    else:
        assert par["rank_checked"] == 1 and par["images"] == 64 and par["bit_identical"] is True and par["keys_compared"] >= 10, par
    assert all(0 < f <= 1.05 for f in out["roofline"]["per_rank_read_ceiling_frac"])


@pytest.mark.parametrize("extra", [[], ["--shared-group", "16"]])
def test_bench_starts_its_own_ranks(dev, extra):
    """`python bench.py --gpus 2` WITHOUT a launcher (how the driver records its single-GPU command): the
    script re-executes itself under torch.distributed.run, rank 0 prints exactly ONE line on stdout, and that line
    describes both ranks.  (gloo: the two ranks share this box's one GPU, which RCCL refuses.)"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "64", "--steps", "2",
           "--warmup", "1", "--cpu-sample", "2"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128 and out["value"] > 0
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == 2 and len(mg["per_rank_ms"]) == 2 and "bench.py itself" in mg["launched_by"]
    # the N > 1 line is as complete as the N = 1 line: the CPU baseline on rank 0's sample, every rank's own
    # sweep fraction, `frac` = the slowest rank's; no choice among allocations anywhere
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] in ("port", "reference") and cb["cores"] >= 1
    assert cb["unit"] == ("frames/sec" if extra else "images/sec") and "rank 0" in cb.get("sample", cb.get("port", {}).get("sample", ""))
    rf = out["roofline"]
    assert len(rf["per_rank_frac"]) == 2 and all(0 < f < 1 for f in rf["per_rank_frac"]) and rf["frac"] == min(rf["per_rank_frac"])
    assert "best_of_n" not in out["placement"] and out["placement"]["solve_ms"] is None
    assert mg["rccl"]["compiled"] >= 22000 and mg["rccl"]["runtime"] >= 22000
    # ... and carries its own parity: the gathered rows of rank 1 / the split result against a one-GPU solve, and rank 0's
    # first images against the CPU oracle (independent images; rank 0's slice of a frame split is no sub-problem of its own)
    if extra:
        assert mg["parity"]["within_gate"] is True and out["check"]["vs_oracle"] is None
    else:
        assert mg["parity"]["bit_identical"] is True and mg["parity"]["rank_checked"] == 1
        vo = out["check"]["vs_oracle"]
        assert vo["images"] == 2 and vo["within_gate"] is True and vo["gate"] == 1e-4 and vo["max_focal_rel"] < 1e-4, vo
    # with the real backend two ranks need two GPUs: on a one-GPU box the answer is ONE JSON line with an error, not a trace
    if torch.cuda.device_count() < 2:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64"], capture_output=True,
                             text=True, timeout=300, cwd=ROOT, env=env)
        lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
        assert res.returncode != 0 and len(lines) == 1
        err = json.loads(lines[0])
        assert err["value"] is None and "only 1 HIP device" in err["error"] and err["n_gpus"] == 2


def test_bench_single_gpu_line_carries_secondary_overlap_and_best_of_n(dev):
    """The N=1 line of record (small shapes here): `secondary` = configs[3] (simple_radial) and configs[4]'s shape
    (shared-16) with their own roofline blocks, `overlap` = the same batch as two halves on two streams (bit-identical),
    `value` / `roofline` from the FIRST allocation with `placement.best_of_n` (the fastest of three allocations) beside them,
    `cpu_baseline` from this box (the port; the reference checkout does not exist on the GPU box)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    # 832 images of 640x480: halves of 416 images are cut into workgroups exactly like the whole batch (bit-identity regime)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "832", "--steps", "2", "--warmup", "1", "--cpu-sample", "4",
           "--placement-tries", "3"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["streams"] == 1
    sec = out["secondary"]
    assert set(sec) == {"simple_radial_B832", "shared16_pinhole", "radial_B832", "simple_divisional_B832"}
    for name, rec in sec.items():
        assert rec["value"] > 0 and rec["steps"] == 5 and rec["roofline"]["launches_timed"] == 5 * 21
        assert 0 < rec["roofline"]["frac"] < 1 and rec["check"]["median_focal_rel_err_vs_gt"] < 5e-3
        # parity INSIDE the record: the first 64 images of the timed batch against the CPU oracle
        vo = rec["check"]["vs_oracle"]
        assert vo["images"] == 64 and vo["gate"] == 1e-4, vo
        if "divisional" in name:
            # the model's k column cancels in float32 (camera.py:913; the reference itself moves by 3e-4 under 1-ulp inputs,
            # tests/golden/make_golden_div_small.py): the record states how many of the 64 images sit within 1e-4 and the medians
            assert vo["images_within_gate"] >= 48 and max(vo["median_focal_rel"], vo["median_gravity_abs"], vo["median_final_cost_rel"]) <= 1e-4, vo
            # ... and how sharp the yardstick is there: every image is within 1e-4 + 10 x (oracle float32 vs its float64 build)
            ys = vo["yardstick"]
            assert ys["images_within_gate_plus_10x_own"] == 64 and ys["images_where_it_exceeds_gate"] >= 64 - vo["images_within_gate"], ys
            assert all(b["hip_vs_oracle32"] <= 1e-4 + 10 * b["oracle32_vs_oracle64"] for b in ys["hip_beyond_gate"]), ys
        else:
            assert vo["within_gate"] is True and vo["images_within_gate"] == 64, vo
            assert max(vo["max_focal_rel"], vo["max_gravity_abs"], vo["max_final_cost_rel"]) <= 1e-4
        assert 0 < rec["roofline"]["frac"] <= rec["roofline"]["read_ceiling_frac"] * 1.02 <= 1.05
    # ... the control for the row-pair walk of simple_divisional's sweep: the same solves with the one-row walk, then the default again
    for name in ("simple_divisional_B832", "radial_B832"):
        sd = sec[name]
        rp = sd["row_pairs_off"]
        assert rp["value"] > 0 and 0 < rp["frac"] < sd["roofline"]["frac"] and rp["on_again"]["frac"] > rp["frac"], (name, rp, sd["roofline"])
        assert rp["median_focal_rel_vs_default"] < 1e-5 and rp["median_gravity_abs_vs_default"] < 1e-5, (name, rp)
    assert "row_pairs_off" not in sec["simple_radial_B832"]
    # ... and the control for the scratch plane on the driver's own box: the same solves with the plane off, then on again
    sr = sec["simple_radial_B832"]
    assert sr["slat_plane_bytes"] == 832 * 480 * 640 * 4 and sr["workspace_bytes"] - sr["slat_plane_bytes"] < 16 * 2 ** 20
    so = sr["slat_off"]
    assert so["bit_identical"] is True and so["value"] > 0 and 0 < so["frac"] < 1 and so["on_again"]["value"] > 0
    assert sec["shared16_pinhole"]["slat_plane_bytes"] == 0 and "slat_off" not in sec["shared16_pinhole"]
    vo = out["check"]["vs_oracle"]
    assert vo["images"] == 4 and vo["within_gate"] is True and vo["max_focal_rel"] < 1e-4, vo
    rf = out["roofline"]
    assert 0 < rf["frac"] <= rf["read_ceiling_frac"] * 1.02 <= 1.05 and abs(rf["frac_of_read_ceiling"] - rf["frac"] / rf["read_ceiling_frac"]) < 2e-3
    assert "ANOTHER box" in (rf["traffic_source"] or "ANOTHER box")
    assert sec["shared16_pinhole"]["unit"] == "frames/sec" and "configs[4]" in sec["shared16_pinhole"]["workload"]
    ov = out["overlap"]
    assert ov["streams"] == 2 and ov["value"] > 0 and ov["bit_identical"] is True
    # the line of record is measured on the FIRST allocation; the best of three allocations rides beside it
    pl = out["placement"]
    assert pl["tries"] == 3 and len(pl["solve_ms"]) == 3 and pl["chosen"] == pl["solve_ms"].index(min(pl["solve_ms"]))
    bn = pl["best_of_n"]
    assert bn["chosen"] == pl["chosen"] and bn["value"] > 0 and bn["ms_per_step"] > 0 and 0 < bn["frac"] < 1
    if bn["chosen"] == 0:
        assert bn["value"] == out["value"] and bn["frac"] == out["roofline"]["frac"]
    assert "first_allocation" not in pl and out["roofline"]["per_rank_frac"] == [out["roofline"]["frac"]]
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] in ("port", "reference") and cb["reference_on_this_box"] == (cb["kind"] == "reference")


@pytest.mark.gpu
@pytest.mark.parametrize("model", ALL_MODELS)
@pytest.mark.parametrize("tag", ["loop", "rpf"])
def test_jacobian_fields_match_reference(dev, oracle, model, tag):
    """gclm_jacobian_fields (the sweep's own pixel code writing its rows out) against the reference's
    J_perspective_field goldens (perspective_fields.py:323-365) and the oracle; host API shapes and flags."""
    from geocalib_amd import perspective_fields as pf
    from geocalib_amd.camera import camera_models
    from geocalib_amd.gravity import Gravity
    g = np.load(os.path.join(GOLDEN, "golden_jac.npz"))
    sph = tag == "loop"
    cam = camera_models[model](torch.from_numpy(g[f"{model}/camera"]).float().to(dev))
    grav = Gravity(torch.from_numpy(g[f"{model}/gravity"]).float().to(dev))
    J_up, J_lat = pf.J_perspective_field(cam, grav, spherical=sph, log_focal=sph)
    ref_up, ref_lat = g[f"{model}/{tag}/J_up"], g[f"{model}/{tag}/J_lat"]
    assert J_up.shape == ref_up.shape and J_lat.shape == ref_lat.shape
    tol = 1e-3 if model == "simple_divisional" else 1e-5      # fp32 on |J| <= 2.2; divisional: reference's own cancellation
    assert np.abs(J_up.cpu().numpy() - ref_up).max() < tol
    assert np.abs(J_lat.cpu().numpy() - ref_lat).max() < tol
    o_up, o_lat = oracle.jacobian_fields(model, 12, 16, g[f"{model}/camera"], g[f"{model}/gravity"], sph, sph, precision="f32")
    assert np.abs(J_up.cpu().numpy() - o_up).max() < tol and np.abs(J_lat.cpu().numpy() - o_lat).max() < tol
    # single-field entry points, disabled fields, unbatched camera
    assert torch.equal(pf.J_up_field(cam, grav, sph, sph), J_up) and torch.equal(pf.J_latitude_field(cam, grav, sph, sph), J_lat)
    z_up, only_lat = pf.J_perspective_field(cam, grav, use_up=False, spherical=sph, log_focal=sph)
    assert z_up.abs().max() == 0 and torch.equal(only_lat, J_lat)
    assert torch.equal(pf.J_up_field(cam[0], grav[0], sph, sph), J_up[:1])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pf.J_up_field(cam.cpu(), grav.cpu())


@pytest.mark.gpu
def test_jacobian_fields_contract_to_the_sweep_system(dev):
    """sum_px w J^T J and w J^T r built from the emitted Jacobian rows reproduce gclm_system's Hessian / gradient
    (squared loss, unit confidences): the emitted rows ARE what the sweep contracts."""
    from geocalib_amd import LMOptimizer, perspective_fields as pf
    from oracle import synth
    model = "simple_radial"
    data, cams, gravs = synth.make_fields(11, range(2), model, 40, 52, noise=0.02, confidences=False)
    opt = LMOptimizer({"camera_model": model, "loss_fn": "squared_loss"}).eval()
    td = {k: torch.from_numpy(v).to(dev) for k, v in data.items()}
    cam = opt.camera_model(torch.from_numpy(cams).to(dev))
    from geocalib_amd.gravity import Gravity
    grav = Gravity(torch.from_numpy(gravs).to(dev))
    sysm = opt.system(td, cam, grav)
    J_up, J_lat = pf.J_perspective_field(cam, grav, spherical=True, log_focal=True)
    up, lat = pf.get_perspective_field(cam, grav)
    r_up = (td["up_field"] - up).permute(0, 2, 3, 1).double()                       # (B,H,W,2)
    r_lat = (torch.sin(td["latitude_field"]) - torch.sin(lat)).permute(0, 2, 3, 1).double()
    J = torch.cat([J_up, J_lat], -2).double()                                          # (B,H,W,3,P)
    r = torch.cat([r_up, r_lat], -1)
    Hs = torch.einsum("bhwrp,bhwrq->bpq", J, J)
    Gs = torch.einsum("bhwrp,bhwr->bp", J, r)
    scale = Hs.abs().amax((1, 2), keepdim=True)
    assert ((Hs - sysm["H"].double()).abs() / scale).max() < 2e-5, ((Hs - sysm["H"]).abs() / scale).max()
    assert ((Gs - sysm["G"].double()).abs() / Gs.abs().amax(1, keepdim=True)).max() < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("model", ALL_MODELS)
def test_jacobian_fields_match_autograd_of_the_forward_model(dev, model):
    """The reference's own Jacobian test (siclib/geometry/gradient_checker.py: analytic J vs jacfwd, atol 5e-3)
    restated: gclm_jacobian_fields against torch.func.jacfwd of the float64 host forward model, random cameras in
    the ranges of TestLM (roll/pitch +-45 deg, vfov 20..90 deg, fx != fy)."""
    from torch.func import jacfwd
    from geocalib_amd import perspective_fields as pf
    from geocalib_amd.camera import camera_models
    from geocalib_amd.gravity import Gravity
    H, W = 18, 22
    nd = 0 if model == "pinhole" else camera_models[model].num_dist_params()
    rng = np.random.default_rng(77)

    def camera_of(theta):
        f, k = theta[2], theta[3:3 + nd]
        kk = torch.cat([k, k]) if nd == 1 else (k if nd == 2 else theta.new_zeros(2))
        return torch.cat([theta.new_tensor([W, H]), f[None] + 1.5, f[None], theta.new_tensor([W / 2 - 0.7, H / 2 + 0.4]), kk])[None]

    def fwd(theta):
        cam, grav = camera_models[model](camera_of(theta)), Gravity.from_rp(theta[0][None], theta[1][None])
        up, lat = pf.get_perspective_field(cam, grav)
        return torch.cat([up[0].permute(1, 2, 0), torch.sin(lat[0]).permute(1, 2, 0)], -1)     # (H,W,3)

    for _ in range(3):
        vfov = np.radians(rng.uniform(20, 90))
        theta = [np.radians(rng.uniform(-45, 45)), np.radians(rng.uniform(-45, 45)), H / 2 / np.tan(vfov / 2)]
        theta += [rng.uniform(-0.3, 0.1), rng.uniform(-0.03, 0.03)][:nd]
        theta = torch.tensor(theta, dtype=torch.float64)
        J_ad = jacfwd(fwd)(theta)                                                                # (H,W,3,P)
        cam = camera_models[model](camera_of(theta).float().to(dev))
        grav = Gravity.from_rp(theta[0][None].float().to(dev), theta[1][None].float().to(dev))
        J_up, J_lat = pf.J_perspective_field(cam, grav, spherical=False, log_focal=False)
        J = torch.cat([J_up[0], J_lat[0]], -2).double().cpu()
        tol = 2e-3 if model == "simple_divisional" else 2e-4
        assert (J - J_ad).abs().max() < tol * max(1.0, J_ad.abs().max().item()), (model, theta, (J - J_ad).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--shared-group", "16"], ["--comm", "rccl"], ["--shared-group", "16", "--comm", "rccl"]])
def test_bench_collectives_through_rccl_with_one_rank(dev, extra):
    """The N>1 code path of bench.py with the REAL backend: torch.distributed "nccl" (= RCCL) process group,
    barrier, max-reduce of the timing, the result all-gather / the per-step all-reduce of the Schur partials --
    one rank (GCLM_FORCE_COLLECTIVES=1), so it runs on a one-GPU box."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl",
           "--batch", "64", "--steps", "2", "--warmup", "1", "--cpu-sample", "0"] + extra
    env = dict(os.environ, GCLM_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NCCL_DEBUG", None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                   # ONE JSON line on stdout, no RCCL banner
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["global_batch"] == 64
    assert out["multi_gpu"]["ranks_seen"] == 1 and out["multi_gpu"]["backend"] == "nccl"
    assert out["multi_gpu"]["collective_ms"] > 0          # the collective really went through RCCL on the stream
    v = out["multi_gpu"]["rccl"]                          # which librccl this process bound vs the header of the build
    assert v["compiled"] == 22707 and v["runtime"] // 10000 == 2
    if "--comm" in extra:
        assert "gclm_comm_" in out["multi_gpu"]["comm"]
    # the line's own parity block, through the REAL collective (one rank: its own shard / whole groups)
    par = out["multi_gpu"]["parity"]
    if "--shared-group" in extra:
        assert par["groups_checked"] == 4 and par["within_gate"] is True, par
    else:
        assert par["rank_checked"] == 0 and par["bit_identical"] is True, par


@pytest.mark.gpu
@pytest.mark.parametrize("model", HIP_MODELS)
def test_result_does_not_depend_on_the_chunking(dev, model, monkeypatch):
    """The partition of an image into workgroup chunks (and with it the striped / flat reduction of the partial
    records) only changes the summation order: 2, 7 and 20 iterations per workgroup agree to rounding."""
    from geocalib_amd import LMOptimizer
    data, _, _ = synth_device(model, 3, 240, 320, dev, seed=5)
    conf = {"camera_model": model, "num_steps": 20, "early_stop": False}
    outs = []
    from geocalib_amd import _lib
    for iters in (2, 7, 20):
        opt = LMOptimizer(conf).eval()
        h = opt._handle(dev)
        _lib.check(_lib.load().gclm_set_sweep_iters(h.ptr, iters), h.ptr, "gclm_set_sweep_iters")
        outs.append(to_np(opt(data)))
    for o in outs[1:]:
        assert np.abs(o["camera"][:, 2:4] / outs[0]["camera"][:, 2:4] - 1).max() < 2e-6
        assert np.abs(o["gravity"] - outs[0]["gravity"]).max() < 2e-6
        assert np.abs(o["final_cost"] / outs[0]["final_cost"] - 1).max() < 1e-5
        assert np.array_equal(o["stop_at"], outs[0]["stop_at"])


@pytest.mark.gpu
@pytest.mark.parametrize("model", ALL_MODELS)
def test_residuals_and_costs_match_reference(dev, model):
    """LMOptimizer.calculate_residuals / calculate_costs (gclm_residual_fields, gclm_huber_costs) against the
    reference's per-pixel outputs (lm_optimizer.py:248-315): noisy fields, perturbed estimate, both Huber branches."""
    from geocalib_amd import LMOptimizer
    from geocalib_amd.gravity import Gravity
    g = np.load(os.path.join(GOLDEN, "golden_jac.npz"))
    data = {k: torch.from_numpy(g[f"{model}/res/{k}"]).to(dev)
            for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    opt = LMOptimizer({"camera_model": model}).eval()
    cam = opt.camera_model(torch.from_numpy(g[f"{model}/res/camera"]).to(dev))
    grav = Gravity(torch.from_numpy(g[f"{model}/res/gravity"]).to(dev))
    res = opt.calculate_residuals(cam, grav, data)
    tol = 2e-5 if model == "simple_divisional" else 2e-6
    for k in ("up_residual", "latitude_residual"):
        assert res[k].shape == g[f"{model}/res/{k}"].shape
        assert np.abs(res[k].cpu().numpy() - g[f"{model}/res/{k}"]).max() < tol, (model, k)
    costs, weights = opt.calculate_costs({k: torch.from_numpy(g[f"{model}/res/{k}"]).to(dev) for k in res}, data)
    for ck in ("up_cost", "latitude_cost"):
        ref = g[f"{model}/res/{ck}"]
        assert np.abs(costs[ck].cpu().numpy() - ref).max() < 2e-6 * np.abs(ref).max(), (model, ck)
    for wk in ("up_weights", "latitude_weights"):
        assert np.abs(weights[wk].cpu().numpy() - g[f"{model}/res/{wk}"]).max() < 2e-6, (model, wk)
    # missing fields / confidences, and the mean of the per-pixel costs is what the fused sweep reports
    only_lat = opt.calculate_residuals(cam, grav, {"latitude_field": data["latitude_field"]})
    assert list(only_lat) == ["latitude_residual"] and torch.equal(only_lat["latitude_residual"], res["latitude_residual"])
    c2, w2 = opt.calculate_costs(res, {})
    assert (w2["up_weights"] <= 1).all() and (w2["up_weights"] > 0).all()
    c3, _ = opt.calculate_costs(res, data)
    sysm = opt.system(data, cam, grav)
    assert torch.allclose(c3["up_cost"].mean(1), sysm["cost_up"], rtol=2e-5)
    assert torch.allclose(c3["latitude_cost"].mean(1), sysm["cost_lat"], rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("model", HIP_MODELS)
def test_one_lm_step_composed_from_the_public_stages(dev, model):
    """The reference's loop body written with the public pieces -- system (residuals + costs + Jacobians +
    reductions), optimizer_step (damped Cholesky on the device), update_estimate -- reproduces what ONE step of the
    fused solve does; optimizer_step itself is checked against a float64 Cholesky and for its zero-step rule."""
    from geocalib_amd import LMOptimizer
    from geocalib_amd.lm_optimizer import early_stop, get_trivial_estimation, optimizer_step, update_lambda
    data, _, _ = synth_device(model, 4, 96, 128, dev, seed=9)
    opt = LMOptimizer({"camera_model": model, "num_steps": 1, "early_stop": False}).eval()
    cam0, grav0 = get_trivial_estimation(data, opt.camera_model)
    opt.setup_optimization_and_priors(data, shared_intrinsics=False)
    s = opt.system(data, cam0, grav0)
    lam = torch.full((4,), 0.1, device=dev)
    delta = optimizer_step(s["G"], s["H"], lam)
    cam1, grav1 = opt.update_estimate(cam0, grav0, delta)
    fused_cam, fused_grav, _ = opt.optimize(data, cam0, grav0)
    assert torch.allclose(cam1._data, fused_cam._data, rtol=2e-6, atol=1e-6), (cam1._data - fused_cam._data).abs().max()
    assert torch.allclose(grav1._data, fused_grav._data, atol=1e-6)
    # against a float64 solve of the damped system
    H64, G64 = s["H"].double().cpu(), s["G"].double().cpu()
    A = H64 + torch.diag_embed((H64.diagonal(dim1=-2, dim2=-1) * 0.1).clamp(min=1e-6))
    ref = torch.cholesky_solve(G64[..., None], torch.linalg.cholesky(A))[..., 0]
    assert torch.allclose(delta.double().cpu(), ref, rtol=1e-4, atol=1e-7)
    assert torch.allclose(optimizer_step(s["G"], s["H"], torch.tensor(0.1, device=dev)), delta)      # scalar lambda
    # a system that is not positive definite takes a zero step and leaves the others alone
    Hbad = s["H"].clone()
    Hbad[1] = -Hbad[1]
    dbad = optimizer_step(s["G"], Hbad, lam)
    assert dbad[1].abs().max() == 0 and torch.equal(dbad[[0, 2, 3]], delta[[0, 2, 3]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        optimizer_step(s["G"].cpu(), s["H"].cpu(), lam.cpu())
    # the two host-side rules of the loop
    prev, new = torch.tensor([1.0, 2.0, 3.0]), torch.tensor([0.5, 2.5, 3.0])
    assert torch.allclose(update_lambda(torch.tensor([0.1, 0.1, 50.0]), prev, new), torch.tensor([0.01, 1.0, 5.0]))
    assert early_stop(prev, prev.clone(), 1e-8, 1e-8) and not early_stop(new, prev, 1e-8, 1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("model", HIP_MODELS)
def test_reference_stage_methods_agree_with_the_fused_path(dev, model):
    """calculate_residuals -> calculate_costs -> setup_system (materialised Jacobians + device contraction) gives the
    Hessian / gradient of the fused sweep; estimate_uncertainty on them gives forward's covariance and sigmas; the
    shared-intrinsics arrow-head has the reference's layout and the fused Schur solve's solution."""
    from geocalib_amd import LMOptimizer
    data, _, _ = synth_device(model, 3, 96, 128, dev, seed=21)
    opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()
    out = opt(data)
    cam, grav = out["camera"], out["gravity"]
    res = opt.calculate_residuals(cam, grav, data)
    costs, weights = opt.calculate_costs(res, data)
    for as_rpf in (False, True):
        G, H = opt.setup_system(cam, grav, res, weights, as_rpf=as_rpf)
        s = opt.system(data, cam, grav, as_rpf=as_rpf)
        scale = s["H"].abs().amax((1, 2), keepdim=True)
        assert ((H - s["H"]).abs() / scale).max() < 2e-5
        assert ((G - s["G"]).abs() / s["H"].abs().amax((1, 2)).sqrt()[:, None]).max() < 1e-3     # G ~ 0 at the optimum
    unc = opt.estimate_uncertainty(cam, grav, res, weights)
    assert torch.allclose(unc["covariance"], out["covariance"], rtol=2e-3, atol=1e-9)
    for k in ("roll_uncertainty", "pitch_uncertainty", "gravity_uncertainty", "focal_uncertainty", "vfov_uncertainty"):
        assert torch.allclose(unc[k], out[k], rtol=2e-3), k
    # shared intrinsics: dense arrow-head (1, 2B+ni, 2B+ni); its damped solve equals per-frame + shared Schur result
    cam0, grav0 = cam, grav
    Gs, Hs = opt.setup_system(cam0, grav0, res, weights, shared_intrinsics=True)
    ni = opt.n_intrinsic_params
    assert Gs.shape == (1, 6 + ni) and Hs.shape == (1, 6 + ni, 6 + ni)
    assert torch.allclose(Hs, Hs.transpose(1, 2))
    Gf, Hf = opt.setup_system(cam0, grav0, res, weights)
    assert torch.allclose(Hs[0, 6:, 6:], Hf[:, 2:, 2:].sum(0), rtol=1e-5)
    assert torch.allclose(Hs[0, 2:4, 2:4], Hf[1, :2, :2]) and torch.allclose(Hs[0, 2:4, 6:], Hf[1, :2, 2:])
    assert Hs[0, 0:2, 2:4].abs().max() == 0
    assert torch.allclose(Gs[0, :6].reshape(3, 2), Gf[:, :2]) and torch.allclose(Gs[0, 6:], Gf[:, 2:].sum(0), rtol=1e-5)


@pytest.mark.gpu
def test_huber_and_scaled_loss_functions(dev):
    """huber_loss / scaled_loss (lm_optimizer.py:61-87) against their definition in float64: value, first derivative
    (the IRLS weight) and second derivative on both sides of the threshold, incl. x = 0."""
    from geocalib_amd.lm_optimizer import huber_loss, scaled_loss
    x = torch.cat([torch.zeros(1), torch.logspace(-6, 4, 400)]).to(dev)
    for a in (1.0, 1e-2):
        loss, d1, d2 = scaled_loss(x * a * a, huber_loss, a)
        y = x.double().cpu()
        sx = torch.sqrt(y + 1e-8)
        ref_loss = torch.where(y <= 1, y, 2 * sx - 1) * a * a
        ref_d1 = torch.where(y <= 1, torch.ones_like(y), 1 / sx)
        ref_d2 = torch.where(y <= 1, torch.zeros_like(y), -(1 / sx) / (2 * y.clamp(min=1e-30))) / (a * a)
        assert torch.allclose(loss.double().cpu(), ref_loss, rtol=2e-6, atol=1e-12)
        assert torch.allclose(d1.double().cpu(), ref_d1, rtol=2e-6)
        assert torch.allclose(d2.double().cpu(), ref_d2, rtol=5e-6, atol=1e-12)


# ------------------------------------------------------------------ degenerate inputs, limits, streams (through gclm_solve)

@pytest.mark.gpu
@pytest.mark.parametrize("model", HIP_MODELS)
def test_bad_images_are_contained(dev, model):
    """A NaN pixel, an image without any confidence and a healthy image in ONE batch.  The reference zeroes the step
    of the WHOLE batch when any Cholesky fails (lm_optimizer.py:129-133: one poisoned image stalls everybody); here
    the failure is contained: the poisoned image takes zero steps and counts them in `step_failures`, its
    neighbours are bit-identical to a clean run."""
    data, _, _ = synth_device(model, 4, 96, 128, dev, seed=3)
    conf = {"camera_model": model, "num_steps": 20, "early_stop": False}
    clean = run_dev(conf, data)
    bad = {k: v.clone() for k, v in data.items()}
    bad["latitude_field"][1, 0, 5, 7] = float("nan")                   # image 1: NaN gradient -> NaN step -> rejected
    bad["up_confidence"][2] = 0                                        # image 2: no information at all
    bad["latitude_confidence"][2] = 0
    out = run_dev(conf, bad)
    for k in ("camera", "gravity", "final_cost", "covariance", "stop_at"):
        if k == "stop_at":
            continue                                                   # batch-global by definition (:619-620)
        assert np.array_equal(out[k][[0, 3]], clean[k][[0, 3]]), f"{k}: a healthy image changed"
    assert out["step_failures"][1] == 20 and out["step_failures"][[0, 2, 3]].max() == 0
    init = run_dev({**conf, "num_steps": 0}, data)
    # a zero step still re-applies f <- exp(log f + 0) and the manifold retraction (as the reference does): rounding only
    assert np.allclose(out["camera"][1], init["camera"][1], rtol=1e-5) and np.allclose(out["gravity"][1], init["gravity"][1], atol=1e-6)
    # zero confidence: H = G = 0, damping floor 1e-6 (:123-126) -> zero step, no failure, finite estimate
    assert np.allclose(out["camera"][2], init["camera"][2], rtol=1e-5) and np.isfinite(out["gravity"][2]).all()
    assert out["final_cost"][2] == 0


def run_dev(conf, data, training=False):
    from geocalib_amd import LMOptimizer
    opt = LMOptimizer(dict(conf))
    out = (opt.train() if training else opt.eval())(data)
    torch.cuda.synchronize()
    return to_np(out)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["simple_radial", "radial"])
def test_prior_dist_against_oracle(dev, oracle, model):
    """`prior_dist` fixes the distortion (lm_optimizer.py:334-336): the reference cannot run it batched
    (camera.py:74-92 raises), so the oracle is the yardstick."""
    inp = np.load(os.path.join(GOLDEN, f"inputs_{model}.npz"))
    data = {k: inp[k] for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    nd = 2 if model == "radial" else 1
    data["prior_dist"] = inp["gt_camera"][:, 6:6 + nd].copy()
    conf = {"camera_model": model, "num_steps": 20, "early_stop": False}
    ref = oracle.solve(data, conf, precision="f32")
    out = run(conf, data, dev)
    compare_result(out, ref, TOL, f"prior_dist/{model}")
    assert np.array_equal(out["camera"][:, 6:6 + nd], data["prior_dist"])      # untouched
    # like the reference (:335-344), the system keeps the distortion columns (only update_estimate skips them, :541-547)
    assert out["covariance"].shape[1:] == (3 + nd, 3 + nd)
    free = run(conf, {k: v for k, v in data.items() if k != "prior_dist"}, dev)
    assert np.abs(free["camera"][:, 6] - out["camera"][:, 6]).max() > 1e-5      # and it does change the answer


@pytest.mark.gpu
def test_8192_images_in_one_call(dev, oracle):
    """BASELINE configs[2]'s total (8192 pinhole images, 640x480, 20 iterations) as ONE call on one GPU: a slice
    solved on its own gives the same bits (images are independent; what an 8-GPU shard computes), and a sample of
    the device-generated images agrees with the oracle."""
    from geocalib_amd import LMOptimizer
    B, H, W = 8192, 480, 640
    data, gtc, gtg = synth_device("pinhole", B, H, W, dev, seed=21)
    opt = LMOptimizer({"camera_model": "pinhole", "num_steps": 20, "early_stop": False}).eval()
    full = to_np(opt(data))
    assert full["step_failures"].max() == 0 and np.isfinite(full["camera"]).all()
    assert np.median(np.abs(full["camera"][:, 3] / gtc[:, 3].cpu().numpy() - 1)) < 1e-3
    lo, hi = 5 * 1024, 6 * 1024                                         # rank 5's shard of an 8-GPU run
    shard = to_np(opt({k: v[lo:hi] for k, v in data.items()}))
    for k in ("camera", "gravity", "final_cost", "covariance"):
        assert np.array_equal(shard[k], full[k][lo:hi]), k
    sample = [0, 4097, 8191]
    host = {k: v[sample].cpu().numpy() for k, v in data.items()}
    ref = oracle.solve(host, {"camera_model": "pinhole", "num_steps": 20, "early_stop": False}, precision="f32")
    compare_result({k: v[sample] for k, v in full.items()}, ref, TOL, "B8192/sample")
    del data


@pytest.mark.gpu
def test_more_than_65535_images_are_chunked(dev):
    """One C call takes at most 65 535 images (grid.y); LMOptimizer slices larger batches instead of raising."""
    from geocalib_amd import LMOptimizer
    B, H, W = 65535 + 9, 8, 16
    data, gtc, _ = synth_device("pinhole", B, H, W, dev, seed=4)
    opt = LMOptimizer({"camera_model": "pinhole", "num_steps": 5, "early_stop": False}).eval()
    out = to_np(opt(data))
    assert out["camera"].shape == (B, 8) and out["covariance"].shape == (B, 3, 3) and np.isfinite(out["camera"]).all()
    tail = to_np(opt({k: v[65535:] for k, v in data.items()}))
    head = to_np(opt({k: v[:100] for k, v in data.items()}))
    assert np.allclose(out["camera"][65535:], tail["camera"], rtol=1e-6) and np.allclose(out["camera"][:100], head["camera"], rtol=1e-6)


@pytest.mark.gpu
def test_two_streams_do_not_share_a_workspace(dev):
    """One optimiser driven from two torch streams at once (the CNN of batch k+1 overlapping the LM of batch k): a
    gclm_handle owns the whole solve workspace, so every (device, stream) gets its own (include/gclm.h)."""
    from geocalib_amd import LMOptimizer
    conf = {"camera_model": "simple_radial", "num_steps": 20, "early_stop": False}
    a, _, _ = synth_device("simple_radial", 192, 240, 320, dev, seed=8)
    b, _, _ = synth_device("simple_radial", 192, 240, 320, dev, seed=9)
    opt = LMOptimizer(conf).eval()
    ra, rb = to_np(opt(a)), to_np(opt(b))                               # one after the other
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    for _ in range(3):                                                  # overlapped, repeatedly
        with torch.cuda.stream(s1):
            oa = opt(a)
        with torch.cuda.stream(s2):
            ob = opt(b)
        torch.cuda.synchronize()
        oa, ob = to_np(oa), to_np(ob)
        for k in ("camera", "gravity", "final_cost", "stop_at"):
            assert np.array_equal(oa[k], ra[k]) and np.array_equal(ob[k], rb[k]), k
    assert len(opt._handles) == 3                                       # default stream + the two side streams


@pytest.mark.gpu
def test_entry_points_leave_the_current_device_alone(dev):
    """Every C entry point restores the caller's current HIP device (also gclm_destroy at garbage collection)."""
    from geocalib_amd import LMOptimizer
    data, _, _ = synth_device("pinhole", 2, 32, 48, dev, seed=1)
    before = torch.cuda.current_device()
    opt = LMOptimizer({"camera_model": "pinhole", "num_steps": 3}).eval()
    opt(data)
    del opt
    import gc
    gc.collect()
    assert torch.cuda.current_device() == before


@pytest.mark.gpu
@pytest.mark.parametrize("model", ALL_MODELS)
def test_one_launch_per_step_equals_the_two_launch_sequence(dev, model):
    """Small batches run ONE launch per LM step (fused_step_kernel: the update of step k-1 in the prologue of every
    workgroup of sweep k, gclm_set_fused_steps).  Same reduction order, same lm_step(): every output must be
    bit-identical to the sweep / update launch pairs -- with the early stop firing inside a launch (default conf, one
    image), fixed step counts, zero / one step, few and many partial records per image (flat / striped reduction), and
    without confidences."""
    from geocalib_amd import LMOptimizer, _lib
    lib = _lib.load()

    def solve(conf, data, mode):
        opt = LMOptimizer(conf).eval()
        h = opt._handle(dev)
        _lib.check(lib.gclm_set_fused_steps(h.ptr, mode), h.ptr, "gclm_set_fused_steps")
        out = to_np(opt(data))
        torch.cuda.synchronize()
        return out

    cases = [(1, 480, 640, {}),                                                      # the interactive case: early stop on the device
             (1, 480, 640, {"num_steps": 20, "early_stop": False}),
             (1, 96, 128, {}), (1, 96, 128, {"num_steps": 7}),
             (3, 120, 160, {"num_steps": 12, "early_stop": False}),
             (2, 64, 80, {"num_steps": 0, "early_stop": False}), (2, 64, 80, {"num_steps": 1, "early_stop": False}),
             (1, 240, 320, {"num_steps": 30, "atol": 1e-4, "rtol": 1e-4}),            # a stop after a handful of steps
             (1, 240, 320, {"num_steps": 3}), (1, 240, 320, {"use_log_focal": False, "use_spherical_manifold": False})]
    stops = []
    for B, H, W, extra in cases:
        data, _, _ = synth_device(model, B, H, W, dev, seed=31)
        for strip in (False, True):
            d = {k: v for k, v in data.items() if not (strip and "confidence" in k)}
            conf = {"camera_model": model, **extra}
            two, one = solve(conf, d, 0), solve(conf, d, 1)
            for k in two:
                assert np.array_equal(two[k], one[k], equal_nan=True), (model, B, H, W, extra, strip, k)
            stops.append((two["stop_at"][0], conf.get("num_steps", 30)))
    assert any(s < n for s, n in stops) and any(s == n for s, n in stops)     # stops before and at the last step both occurred
    # The first launch of the one-launch-per-step path builds the initial estimate itself (round 4: no init_kernel launch):
    # every source of that estimate -- priors, `scales`, siclib's heuristic (it reads three pixels of the fields), a
    # caller-provided camera / gravity (gclm_solve) -- must give the two-launch path's bits
    data, gt_cam, gt_grav = synth_device(model, 1, 240, 320, dev, seed=32)
    variants = [({}, {"prior_focal": (gt_cam[:, 3] * 1.1).contiguous()}),
                ({}, {"prior_gravity": torch.nn.functional.normalize(gt_grav + 0.05, dim=-1).contiguous()}),
                ({}, {"scales": torch.tensor([0.8, 1.25], device=dev)}),
                ({"init_conf": {"name": "heuristic"}}, {}),
                ({"init_conf": {"name": "heuristic"}, "num_steps": 0}, {})]
    for extra, more in variants:
        conf, d = {"camera_model": model, **extra}, {**data, **more}
        two, one = solve(conf, d, 0), solve(conf, d, 1)
        for k in two:
            assert np.array_equal(two[k], one[k], equal_nan=True), (model, extra, list(more), k)

    def solve_from(conf, mode):                          # LMOptimizer.optimize -> gclm_solve: the caller's initial estimate
        from geocalib_amd.lm_optimizer import get_trivial_estimation
        opt = LMOptimizer(conf).eval()
        h = opt._handle(dev)
        _lib.check(lib.gclm_set_fused_steps(h.ptr, mode), h.ptr, "gclm_set_fused_steps")
        opt.setup_optimization_and_priors(data, shared_intrinsics=False)
        cam0, grav0 = get_trivial_estimation(data, opt.camera_model)
        cam0 = cam0.__class__(cam0._data * torch.tensor([1, 1, 1.2, 1.2, 1, 1, 1, 1], device=dev))
        cam, grav, info = opt.optimize(data, cam0, grav0)
        torch.cuda.synchronize()
        return cam._data.cpu().numpy(), grav._data.cpu().numpy(), info["final_cost"].cpu().numpy()
    for a, b in zip(solve_from({"camera_model": model}, 0), solve_from({"camera_model": model}, 1)):
        assert np.array_equal(a, b, equal_nan=True), model
    # where it is not valid (a batch with the batch-global early stop) the request is ignored, not an error
    data, _, _ = synth_device(model, 3, 64, 80, dev, seed=2)
    a, b = solve({"camera_model": model}, data, 0), solve({"camera_model": model}, data, 1)
    assert all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ALL_MODELS)
def test_slat_plane_changes_nothing_but_the_workspace(dev, model):
    """The distortion models keep sin(latitude_field) in a library-owned scratch plane: the first sweep of a solve stores it,
    every later sweep loads it instead of re-evaluating the polynomial per pixel (gclm_set_slat_plane).  Same
    polynomial, so every output must be bit-identical to the solve that computes it in every sweep -- with the early stop,
    fixed step counts, one / zero steps, both parametrisations, an anisotropic start (`scales`: the general-focal final
    sweep), shared intrinsics in one call and through the split protocol; and the one-launch-per-step path (which never uses
    the plane) must agree with both.  Pinhole has the instantiations too but the built-in choice leaves them alone."""
    from geocalib_amd import LMOptimizer, _lib
    from geocalib_amd.parallel import SharedIntrinsicsSplit
    lib = _lib.load()
    probe = LMOptimizer({"camera_model": model})._handle(dev)
    assert lib.gclm_set_slat_plane(None, 1) == -1
    assert lib.gclm_set_slat_plane(probe.ptr, 2) == -3 and "gclm_set_slat_plane" in _lib.last_error(probe.ptr)
    assert lib.gclm_set_slat_plane(probe.ptr, -1) == 0

    def solve(conf, data, mode, fused=0, split_groups=0):
        opt = LMOptimizer(conf).eval()
        h = opt._handle(dev)
        _lib.check(lib.gclm_set_fused_steps(h.ptr, fused), h.ptr, "gclm_set_fused_steps")
        _lib.check(lib.gclm_set_slat_plane(h.ptr, mode), h.ptr, "gclm_set_slat_plane")
        if split_groups:
            B = data["latitude_field"].shape[0]
            gof = torch.arange(B, dtype=torch.int32) // (B // split_groups)
            out = to_np(SharedIntrinsicsSplit(opt, num_groups=split_groups)(data, gof))
        else:
            out = to_np(opt(data))
        torch.cuda.synchronize()
        return out, lib.gclm_workspace_bytes(h.ptr)

    B, H, W = 6, 240, 320
    plane = B * H * W * 4
    data, gt_cam, _ = synth_device(model, B, H, W, dev, seed=77)
    confs = [{}, {"num_steps": 20, "early_stop": False}, {"num_steps": 1, "early_stop": False}, {"num_steps": 0, "early_stop": False},
             {"use_log_focal": False, "use_spherical_manifold": False, "num_steps": 8},
             {"shared_intrinsics": True, "num_steps": 10, "early_stop": False},
             {"shared_intrinsics": True, "group_size": 3, "num_steps": 10, "early_stop": False}]
    for extra in confs:
        for more in ({}, {"scales": torch.tensor([0.8, 1.25], device=dev)}):
            if more and extra.get("shared_intrinsics"):
                continue
            conf, d = {"camera_model": model, **extra}, {**data, **more}
            (off, ws_off), (on, ws_on), (auto, ws_auto) = solve(conf, d, 0), solve(conf, d, 1), solve(conf, d, -1)
            for k in off:
                assert np.array_equal(off[k], on[k], equal_nan=True), (model, extra, list(more), k)
                assert np.array_equal(off[k], auto[k], equal_nan=True), (model, extra, list(more), k)
            uses = extra.get("num_steps", 30) >= 1
            assert ws_off < plane and (ws_on >= plane) == uses, (model, extra, ws_off, ws_on, plane)
            assert (ws_auto >= plane) == (uses and model != "pinhole"), (model, extra, ws_auto)
            if not extra.get("shared_intrinsics") and not extra.get("early_stop", True):
                one, ws_one = solve(conf, d, 1, fused=1)             # one launch per step: no plane, same bits
                assert ws_one < plane and all(np.array_equal(off[k], one[k], equal_nan=True) for k in off), (model, extra)
    # fields without confidences take the plain sweep (the plane exists for the five-plane instantiation only)
    bare = {k: v for k, v in data.items() if "confidence" not in k}
    (off, _), (on, ws_on) = solve({"camera_model": model}, bare, 0), solve({"camera_model": model}, bare, 1)
    assert ws_on < plane and all(np.array_equal(off[k], on[k], equal_nan=True) for k in off)
    # the split protocol (gclm_shared_begin / reduce / apply / finish): the plane is filled by the session's first sweep
    conf = {"camera_model": model, "shared_intrinsics": True, "num_steps": 10, "early_stop": False}
    for groups in (1, 2):
        (off, ws_off), (on, ws_on) = solve(conf, data, 0, split_groups=groups), solve(conf, data, 1, split_groups=groups)
        assert ws_off < plane <= ws_on and all(np.array_equal(off[k], on[k], equal_nan=True) for k in off), (model, groups)
    # a handle that solved with the plane keeps working when the next call has no use for it, and the other way round
    opt = LMOptimizer({"camera_model": model, "num_steps": 5, "early_stop": False}).eval()
    h = opt._handle(dev)
    _lib.check(lib.gclm_set_fused_steps(h.ptr, 0), h.ptr, "gclm_set_fused_steps")
    _lib.check(lib.gclm_set_slat_plane(h.ptr, 1), h.ptr, "gclm_set_slat_plane")
    first = to_np(opt(data))
    to_np(opt(bare))
    again = to_np(opt(data))
    assert all(np.array_equal(first[k], again[k], equal_nan=True) for k in first)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["radial", "simple_divisional"])
def test_row_pairs_equal_the_one_row_walk_to_summation_order(dev, model):
    """gclm_set_row_pairs: a lane of the sweep takes row H - y along with row y and evaluates what depends on r^2 once for both
    (gclm_pass.hip: row_math_mirror).  Same per-pixel values, another order of a lane's additions:
      * after ONE LM step (nothing amplifies yet) the two walks agree to float32 summation order on every shape -- whole
        tiles, ragged tiles, strips, tiny images; both parametrisations; `scales` (anisotropic focal, principal point off
        the centre of the field: the pairs are walked WITHOUT sharing, decided per image on the device); an explicit
        off-centre camera through optimize(); with and without the scratch plane;
      * where the sweep has no row-pair walker (odd height, a missing confidence, the scalar path) the knob changes nothing,
        bit for bit; mode 0 and the built-in choice on a small launch are the one-row walk, bit for bit;
      * the walk is deterministic (same call, same bits) and shard-invariant (images are independent);
      * a full solve of a BATCH (the built-in choice for both models: > 768 workgroups) recovers the ground truth and
        stays within the fuzz gate of the one-row solve on the well-conditioned images.
    The gates against the REFERENCE run on the row-pair walk in test_hip_matches_reference_small / _full_size_other_models."""
    from conftest import MEASURED, result_spread
    from geocalib_amd import LMOptimizer, _lib
    from geocalib_amd.camera import camera_models
    from geocalib_amd.gravity import Gravity
    lib = _lib.load()
    probe = LMOptimizer({"camera_model": model})._handle(dev)
    assert lib.gclm_set_row_pairs(None, 1) == -1
    assert lib.gclm_set_row_pairs(probe.ptr, 2) == -3 and "gclm_set_row_pairs" in _lib.last_error(probe.ptr)
    assert lib.gclm_set_row_pairs(probe.ptr, -1) == 0

    def solve(conf, data, pairs, slat=-1, fused=-1):
        opt = LMOptimizer({"camera_model": model, **conf}).eval()
        opt.row_pairs = pairs
        opt.overlap_streams = 1
        h = opt._handle(dev)
        _lib.check(lib.gclm_set_slat_plane(h.ptr, slat), h.ptr, "gclm_set_slat_plane")
        _lib.check(lib.gclm_set_fused_steps(h.ptr, fused), h.ptr, "gclm_set_fused_steps")
        out = to_np(opt(data))
        torch.cuda.synchronize()
        return out

    def same_bits(a, b):
        return all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)

    one_step = {"num_steps": 1, "early_stop": False}
    worst = np.zeros(4)
    for (B, H, W) in ((6, 240, 320), (3, 230, 324), (2, 64, 2600), (5, 6, 8), (2, 480, 640)):
        data, _, _ = synth_device(model, B, H, W, dev, seed=5)
        for extra in ({}, {"use_log_focal": False, "use_spherical_manifold": False}):
            for more in ({}, {"scales": torch.tensor([0.8, 1.25], device=dev)}):
                d = {**data, **more}
                off, on = solve({**one_step, **extra}, d, False), solve({**one_step, **extra}, d, True)
                sp = result_spread(on, off)
                worst = np.maximum(worst, sp if H * W >= 1000 else 0.0)
                # (48 pixels: simple_divisional's cost moves by 1e-4 for a change of k in its last bit -- the walk over a
                #  tiny image is checked for its indexing, which would show at O(1))
                assert (sp < (1e-5 if H * W >= 1000 else 1e-3)).all(), (model, (B, H, W), extra, list(more), sp)
                # (the 2600 x 64 strip: focal and distortion are not separable on it -- H_ff H_kk - H_fk^2 cancels to the last
                #  bits, the oracle's own float32 covariance comes out with negative variances there: no uncertainty to compare)
                for k in ("covariance", "focal_uncertainty", "gravity_uncertainty") if W < 2048 and H * W >= 1000 else ():
                    assert np.abs(on[k] - off[k]).max() <= 1e-3 * np.abs(off[k]).max(), (model, (B, H, W), extra, k)
                assert not same_bits(on, off) or H * W < 100, "the knob did not reach the sweep"
                # deterministic; indifferent to the scratch plane; mode 0 = built-in choice on a launch this small
                assert same_bits(on, solve({**one_step, **extra}, d, True))
                assert same_bits(on, solve({**one_step, **extra}, d, True, slat=0))
                assert same_bits(off, solve({**one_step, **extra}, d, None, fused=0))
    MEASURED[f"row_pairs/{model}/one_step_worst"] = worst.tolist()
    # no row-pair walker for these: the knob must change nothing
    data, _, _ = synth_device(model, 3, 231, 320, dev, seed=6)                      # odd height
    assert same_bits(solve(one_step, data, False), solve(one_step, data, True))
    data, _, _ = synth_device(model, 3, 240, 320, dev, seed=6)
    bare = {k: v for k, v in data.items() if k != "up_confidence"}                  # four planes
    assert same_bits(solve(one_step, bare, False), solve(one_step, bare, True))
    odd = {k: v[..., :318].contiguous() for k, v in data.items()}                   # 318 px: the scalar path
    assert same_bits(solve(one_step, odd, False), solve(one_step, odd, True))
    # an explicit camera whose principal point is not the centre (optimize(): gclm_solve)
    data, gt_cam, gt_grav = synth_device(model, 4, 240, 320, dev, seed=9)
    cam0 = gt_cam.clone()
    cam0[:, 4] += 3.0
    cam0[:, 5] -= 2.0
    res = {}
    for pairs in (False, True):
        opt = LMOptimizer({"camera_model": model, "num_steps": 2, "early_stop": False}).eval()
        opt.row_pairs = pairs
        opt.setup_optimization_and_priors(data)
        c, g, info = opt.optimize(data, camera_models[model](cam0.clone()), Gravity(gt_grav.clone()))
        res[pairs] = {"camera": c._data.cpu().numpy(), "gravity": g._data.cpu().numpy(),
                      "final_cost": info["final_cost"].cpu().numpy(), "initial_cost": info["initial_cost"].cpu().numpy()}
    assert (result_spread(res[True], res[False]) < 1e-5).all(), result_spread(res[True], res[False])
    assert np.array_equal(res[True]["camera"][:, 4:6], cam0[:, 4:6].cpu().numpy())
    # a batch: the built-in choice pairs the rows, shards agree with the whole
    B, H, W = 416, 480, 640          # (halves of 208 images: still cut into workgroups like the whole batch, plan_geometry)
    data, gt_cam, gt_grav = synth_device(model, B, H, W, dev, seed=12)
    conf = {"num_steps": 20, "early_stop": False}
    auto, on, off = solve(conf, data, None), solve(conf, data, True), solve(conf, data, False)
    assert same_bits(auto, on) and not same_bits(on, off), "built-in choice for a batch: row pairs (both models)"
    half = solve(conf, {k: v[B // 2:] for k, v in data.items()}, True)
    assert all(np.array_equal(half[k], on[k][B // 2:], equal_nan=True) for k in ("camera", "gravity", "final_cost", "covariance"))
    f_err = np.abs(on["camera"][:, 3] / gt_cam[:, 3].cpu().numpy() - 1)
    assert np.median(f_err) < 1e-3, np.median(f_err)
    sp = np.array([result_spread({k: on[k][i:i + 1] for k in on}, {k: off[k][i:i + 1] for k in off}) for i in range(B)])
    gate = FUZZ_GATE[model] * (1.0 if model == "radial" else 10.0)
    ok = (sp < gate).all(1)
    # (radial: 3 of 416 images pass through a k clamp with a near-singular (focal, k1, k2) block and amplify rounding-level
    # differences a thousandfold, like fuzz 115/45 -- profiles/r06_fuzz_115_45_diagnosis.log)
    MEASURED[f"row_pairs/{model}/batch"] = {"median_spread": np.median(sp, 0).tolist(), "worst": sp.max(0).tolist(), "within_gate": float(ok.mean())}
    # (simple_divisional's k column cancels in float32, camera.py:913: a few images of a hundred amplify ANY rounding-level
    # difference -- the reference moves as far under 1-ulp input perturbations; the median is what a broken walk would move)
    assert np.median(sp, 0).max() < 2e-5 and ok.mean() >= (0.98 if model == "radial" else 0.9), (np.median(sp, 0), sp.max(0), ok.mean())


@pytest.mark.parametrize("model", ["simple_radial", "simple_divisional"])
def test_slat_plane_is_optional_and_exactly_sized(dev, model):
    """VERDICT r05 #3 / ADVICE r05: the sin(latitude) scratch plane is an allocation of its own, of exactly B x H x W x 4 bytes
    (no 25 % headroom: the plane is 400x the rest of the workspace), and a solve that cannot have it -- here: a plane above
    gclm_set_slat_plane_limit, the same branch a failing hipMalloc takes -- goes on WITHOUT it and returns the bits of the
    solve that never had one, instead of failing with -10.  A refused size is remembered (no failing allocation per call)
    until the limit or the mode is set again; gclm_release_workspace gives everything back and the handle stays usable."""
    from geocalib_amd import LMOptimizer, _lib
    lib = _lib.load()
    B, H, W = 6, 240, 320
    plane = B * H * W * 4
    data, _, _ = synth_device(model, B, H, W, dev, seed=5)
    conf = {"camera_model": model, "num_steps": 8, "early_stop": False}

    def fresh(mode=-1, limit=None):
        opt = LMOptimizer(conf).eval()
        h = opt._handle(dev)
        _lib.check(lib.gclm_set_fused_steps(h.ptr, 0), h.ptr, "gclm_set_fused_steps")
        _lib.check(lib.gclm_set_slat_plane(h.ptr, mode), h.ptr, "gclm_set_slat_plane")
        if limit is not None:
            _lib.check(lib.gclm_set_slat_plane_limit(h.ptr, limit), h.ptr, "gclm_set_slat_plane_limit")
        return opt, h

    def solve(opt):
        out = to_np(opt(data))
        torch.cuda.synchronize()
        return out

    def same(a, b):
        return all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)

    opt0, h0 = fresh(mode=0)
    off = solve(opt0)
    core = lib.gclm_workspace_bytes(h0.ptr)
    assert lib.gclm_slat_plane_bytes(h0.ptr) == 0 and 0 < core < plane // 2, core
    opt1, h1 = fresh()
    on = solve(opt1)
    assert lib.gclm_slat_plane_bytes(h1.ptr) == plane                   # exactly: no headroom on the plane
    assert lib.gclm_workspace_bytes(h1.ptr) == core + plane and same(off, on)
    # a plane one byte above the limit: the solve runs without it and returns the same bits; the workspace stays small
    opt2, h2 = fresh(limit=plane - 1)
    capped = solve(opt2)
    assert lib.gclm_slat_plane_bytes(h2.ptr) == 0 and lib.gclm_workspace_bytes(h2.ptr) == core and same(off, capped)
    assert same(off, solve(opt2)) and lib.gclm_slat_plane_bytes(h2.ptr) == 0      # remembered: not retried per call
    # ... raising the limit to the plane's size lets the NEXT solve have it; the built-in rule (0) as well
    assert lib.gclm_set_slat_plane_limit(h2.ptr, plane) == 0
    assert same(off, solve(opt2)) and lib.gclm_slat_plane_bytes(h2.ptr) == plane
    opt3, h3 = fresh(limit=1)
    assert same(off, solve(opt3)) and lib.gclm_slat_plane_bytes(h3.ptr) == 0
    assert lib.gclm_set_slat_plane_limit(h3.ptr, 0) == 0 and same(off, solve(opt3)) and lib.gclm_slat_plane_bytes(h3.ptr) == plane
    # a smaller batch fits the plane it holds; a larger one above the limit keeps the old plane and runs without
    small = {k: v[:3].contiguous() for k, v in data.items()}
    ref_small = to_np(opt0(small))
    assert lib.gclm_set_slat_plane_limit(h2.ptr, plane) == 0
    got_small = to_np(opt2(small))
    torch.cuda.synchronize()
    assert same(ref_small, got_small) and lib.gclm_slat_plane_bytes(h2.ptr) == plane
    assert lib.gclm_set_slat_plane_limit(h2.ptr, plane // 2) == 0
    assert same(off, solve(opt2)) and lib.gclm_slat_plane_bytes(h2.ptr) == plane     # (held, and large enough: used)
    opt4, h4 = fresh(limit=plane // 2)
    assert same(ref_small, {k: v for k, v in to_np(opt4(small)).items()}) and lib.gclm_slat_plane_bytes(h4.ptr) == plane // 2
    assert same(off, solve(opt4)) and lib.gclm_slat_plane_bytes(h4.ptr) == plane // 2    # the 6-image solve: refused, no plane
    # shared intrinsics through the split protocol: the session falls back the same way
    from geocalib_amd.parallel import SharedIntrinsicsSplit
    sconf = {**conf, "shared_intrinsics": True}
    outs = []
    for limit in (0, 1):
        opt = LMOptimizer(sconf).eval()
        h = opt._handle(dev)
        _lib.check(lib.gclm_set_slat_plane_limit(h.ptr, limit), h.ptr, "gclm_set_slat_plane_limit")
        outs.append(to_np(SharedIntrinsicsSplit(opt, num_groups=1)(data, torch.zeros(B, dtype=torch.int32))))
        torch.cuda.synchronize()
        assert lib.gclm_slat_plane_bytes(h.ptr) == (plane if limit == 0 else 0)
    assert same(outs[0], outs[1])
    # gclm_release_workspace: everything goes back, the handle solves again (and re-allocates)
    assert lib.gclm_release_workspace(None) == -1 and lib.gclm_release_workspace(h1.ptr) == 0
    assert lib.gclm_workspace_bytes(h1.ptr) == 0 and lib.gclm_slat_plane_bytes(h1.ptr) == 0
    assert same(off, solve(opt1)) and lib.gclm_workspace_bytes(h1.ptr) == core + plane
    assert lib.gclm_set_slat_plane_limit(None, 0) == -1 and lib.gclm_slat_plane_bytes(None) == 0


@pytest.mark.gpu
def test_merge_stop_at_skips_empty_parts(dev):
    """a part of zero images (a reused handle, or one that never solved) is skipped by gclm_merge_stop_at
    instead of failing its size check; the non-empty parts still get the whole batch's stop_at."""
    from geocalib_amd import LMOptimizer, _lib
    lib, C = _lib.load(), _lib.C
    B, H, W = 4, 96, 128
    data, _, _ = synth_device("pinhole", B, H, W, dev, seed=9)
    conf = {"camera_model": "pinhole", "num_steps": 12, "early_stop": False}
    opts = [LMOptimizer(conf).eval() for _ in range(3)]
    hs = [o._handle(dev) for o in opts]
    for h in hs:
        _lib.check(lib.gclm_set_sweep_iters(h.ptr, 4), h.ptr, "gclm_set_sweep_iters")     # same cut whatever the part size
    hw = LMOptimizer(conf).eval()
    _lib.check(lib.gclm_set_sweep_iters(hw._handle(dev).ptr, 4), None, "gclm_set_sweep_iters")
    whole = to_np(hw(data))
    opts[1](data)                                     # part 1's handle is REUSED: it last solved 4 images, now holds none
    cuts = [(0, 3), (3, 3), (3, 4)]
    raws = []
    for o, (lo, hi) in zip(opts, cuts):
        if hi > lo:
            o({k: v[lo:hi].contiguous() for k, v in data.items()})
            raws.append(o._last_raw[2])
        else:
            raws.append(torch.zeros((0, _lib.INFO_STRIDE), device=dev))
    torch.cuda.synchronize()
    parts = (C.c_void_p * 3)(*[h.ptr.value for h in hs])
    infos = (C.c_void_p * 3)(*[r.data_ptr() if r.numel() else None for r in raws])
    sizes = (C.c_int * 3)(3, 0, 1)
    stream = torch.cuda.current_stream(dev).cuda_stream
    assert lib.gclm_merge_stop_at(parts, infos, sizes, 3, stream) == 0, _lib.last_error(hs[0].ptr)
    torch.cuda.synchronize()
    got = torch.cat([raws[0], raws[2]])[:, _lib.INFO["stop_at"]].cpu().numpy()
    assert np.array_equal(got, whole["stop_at"]) and len(set(got.tolist())) == 1
    never = LMOptimizer(conf).eval()._handle(dev)     # a handle that never solved anything, as an empty part
    parts2 = (C.c_void_p * 3)(hs[0].ptr.value, never.ptr.value, hs[2].ptr.value)
    assert lib.gclm_merge_stop_at(parts2, infos, sizes, 3, stream) == 0
    assert lib.gclm_merge_stop_at(parts2, infos, (C.c_int * 3)(3, 1, 1), 3, stream) == -3     # ... but not as a non-empty one
    assert lib.gclm_merge_stop_at(parts, infos, (C.c_int * 3)(0, 0, 0), 3, stream) == 0       # nothing to do
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_read_probe_streams_the_callers_planes(dev):
    """gclm_read_probe (bench.py: roofline.read_ceiling_frac): one launch over the caller's own planes; argument checks."""
    from geocalib_amd import _lib
    lib, C = _lib.load(), _lib.C
    planes = [torch.randn(64 * 1024, device=dev) for _ in range(5)]
    arr = (C.c_void_p * 5)(*[p.data_ptr() for p in planes])
    s = torch.cuda.current_stream(dev).cuda_stream
    before = [p.clone() for p in planes]
    assert lib.gclm_read_probe(arr, 5, planes[0].numel(), s) == 0
    assert lib.gclm_read_probe(arr, 5, 0, s) == 0
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, planes))              # read-only
    assert lib.gclm_read_probe(arr, 9, 1024, s) == -3 and lib.gclm_read_probe(arr, 5, 1022, s) == -3
    odd = (C.c_void_p * 1)(planes[0][1:].data_ptr())
    assert lib.gclm_read_probe(odd, 1, 1024, s) == -3 and lib.gclm_read_probe(None, 1, 1024, s) == -3


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["pinhole", "simple_divisional"])
def test_paced_launches_change_how_many_launches_are_issued_and_nothing_else(dev, model):
    """gclm_set_paced_launches: a single-image solve with early stop issues launch k only after launch k - depth has
    reported and none after the stop.  Every output stays bit-identical (the device-side skip is what decides), fewer
    sweeps are issued when the stop comes early, back-to-back solves without a sync in between do not read each other's
    reports (epoch tag), and the knob is ignored where it cannot help (a batch, early_stop = False, two-launch path)."""
    import ctypes as C
    from geocalib_amd import LMOptimizer, _lib
    lib = _lib.load()

    def solve(conf, data, depth, repeats=1, fused=-1):
        opt = LMOptimizer(conf).eval()
        opt.paced_launches = depth
        h = opt._handle(dev)
        _lib.check(lib.gclm_set_fused_steps(h.ptr, fused), h.ptr, "gclm_set_fused_steps")
        opt(data)
        torch.cuda.synchronize()
        _lib.check(lib.gclm_set_timing(h.ptr, 1), h.ptr, "gclm_set_timing")
        outs = [opt(data) for _ in range(repeats)]                      # no sync in between
        torch.cuda.synchronize()
        n, ms = C.c_int(0), C.c_float(0)
        _lib.check(lib.gclm_last_pass_timing(h.ptr, C.byref(n), C.byref(ms)), h.ptr, "gclm_last_pass_timing")
        return [to_np(o) for o in outs], n.value // repeats

    assert lib.gclm_set_paced_launches(None, 3) == -1
    probe = LMOptimizer({"camera_model": model})._handle(dev)
    assert lib.gclm_set_paced_launches(probe.ptr, 17) == -3 and "gclm_set_paced_launches" in _lib.last_error(probe.ptr)
    assert lib.gclm_set_paced_launches(probe.ptr, 0) == 0

    saved = 0
    for H, W, extra in ((480, 640, {}), (240, 320, {"atol": 1e-4, "rtol": 1e-4}), (96, 128, {"num_steps": 3}), (96, 128, {"num_steps": 12})):
        data, _, _ = synth_device(model, 1, H, W, dev, seed=31)
        conf = {"camera_model": model, **extra}
        (ref,), n_ref = solve(conf, data, 0)
        assert n_ref == conf.get("num_steps", 30) + 1                  # one sweep launch per step + the final one
        for depth in (1, 3, 16):
            outs, n = solve(conf, data, depth, repeats=4)
            for o in outs:
                for k in ref:
                    assert np.array_equal(ref[k], o[k], equal_nan=True), (model, H, W, extra, depth, k)
            stop = int(ref["stop_at"][0])
            assert n <= n_ref and n >= min(stop + 1, n_ref), (n, n_ref, stop)     # never fewer than the device needs
            if stop + depth + 3 < n_ref:
                assert n <= stop + depth + 3, (n, stop, depth)                    # the detecting launch, at most depth more, the final one
                saved += n_ref - n
    assert saved > 0
    # ignored: a batch (the stop is batch-global), early_stop = False, the two-launch path
    data, _, _ = synth_device(model, 2, 96, 128, dev, seed=5)
    for conf, fused in (({"camera_model": model}, -1), ({"camera_model": model, "early_stop": False, "num_steps": 6}, -1)):
        (a,), na = solve(conf, data, 0, fused=fused)
        (b,), nb = solve(conf, data, 3, fused=fused)
        assert na == nb and all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)
    data, _, _ = synth_device(model, 1, 96, 128, dev, seed=5)
    (a,), na = solve({"camera_model": model}, data, 0, fused=0)
    (b,), nb = solve({"camera_model": model}, data, 3, fused=0)
    assert na == nb and all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)


@pytest.mark.gpu
def test_fastest_placement_returns_one_of_its_candidates(dev):
    """geocalib_amd.fields.fastest_placement: `tries` allocations alive side by side, each solved three times (warm-up + the better of two timed), the fastest kept;
    tries <= 1 allocates once and times nothing."""
    from geocalib_amd import LMOptimizer
    from geocalib_amd.fields import fastest_placement
    opt = LMOptimizer({"camera_model": "pinhole", "num_steps": 3, "early_stop": False}).eval()
    made = []

    def allocate():
        d, _, _ = synth_device("pinhole", 4, 96, 128, dev, seed=9)
        made.append(d)
        return d
    fields, ms = fastest_placement(allocate, opt, tries=3)
    assert len(made) == 3 and len(ms) == 3 and all(t > 0 for t in ms)
    assert fields is made[min(range(3), key=ms.__getitem__)]
    ptrs = {d["latitude_field"].data_ptr() for d in made}
    assert len(ptrs) == 3                                       # alive at the same time: three different allocations
    one, none = fastest_placement(allocate, opt, tries=1)
    assert none == [] and one is made[3]


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["pinhole", "simple_radial"])
def test_overlap_streams_equals_the_single_call(dev, model):
    """LMOptimizer.overlap_streams: a batch of independent images with a fixed step count solved as n contiguous parts on
    n side streams (fork / join by events).  Every output equals the single call's bit for bit when the parts are cut like
    the whole batch (here: >= 256 images of 640x480 per part), with an uneven split and per-image priors as well; the knob
    is ignored with early_stop (one decision over the whole batch), shared intrinsics and parts below 256 images."""
    from geocalib_amd import LMOptimizer
    B, H, W = 513, 480, 640
    data, gt_cam, _ = synth_device(model, B, H, W, dev, seed=3)

    def solve(conf, d, n):
        opt = LMOptimizer({"camera_model": model, **conf}).eval()
        opt.overlap_streams = n
        out = opt(d)
        out2 = opt(d)                                     # side streams and their handles are reused
        torch.cuda.synchronize()
        a, b = to_np(out), to_np(out2)
        assert all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)
        return a, opt

    fixed = {"num_steps": 4, "early_stop": False}
    one, _ = solve(fixed, data, 1)
    two, opt2 = solve(fixed, data, 2)
    assert opt2._overlap_parts(B) == 2 and len(opt2._handles) >= 2
    for k in one:
        assert np.array_equal(one[k], two[k], equal_nan=True), (model, k)
    eight, opt8 = solve(fixed, data, 8)                                   # capped: 513 // 256 = 2 parts
    assert opt8._overlap_parts(B) == 2
    assert all(np.array_equal(one[k], eight[k], equal_nan=True) for k in one)
    # infos["stop_at"] is one number for the WHOLE batch: with 20 steps the two halves become "close" at different steps
    # (seed 2024: 13 / 14), and the overlapped solve must still report the single call's (gclm_merge_stop_at sums the parts'
    # per-step counters)
    d20, _, _ = synth_device(model, B, H, W, dev, seed=2024)
    long = {"num_steps": 20, "early_stop": False}
    o1, _ = solve(long, d20, 1)
    o2, _ = solve(long, d20, 2)
    assert 1 < o1["stop_at"][0] < 20 and len(set(o1["stop_at"].tolist())) == 1
    for k in o1:
        assert np.array_equal(o1[k], o2[k], equal_nan=True), (model, k)
    withp = {**data, "prior_focal": gt_cam[:, 3].contiguous()}
    p1, _ = solve(fixed, withp, 1)
    p2, _ = solve(fixed, withp, 2)
    assert all(np.array_equal(p1[k], p2[k], equal_nan=True) for k in p1)
    # ignored: early stop, shared intrinsics
    assert LMOptimizer({"camera_model": model})._overlap_parts(B) == 1
    sh = LMOptimizer({"camera_model": model, "shared_intrinsics": True, "early_stop": False})
    sh.overlap_streams = 2
    assert sh._overlap_parts(B) == 1
    small = LMOptimizer({"camera_model": model, "early_stop": False})
    small.overlap_streams = 2
    assert small._overlap_parts(300) == 1 and small._overlap_parts(512) == 2
    # the default (None): the library decides -- two parts only far inside the regime where a part is cut like the batch
    auto = LMOptimizer({"camera_model": model, "early_stop": False})
    assert auto.overlap_streams is None
    assert auto._overlap_parts(1024, 480, 640) == 2 and auto._overlap_parts(820, 480, 640) == 2
    assert auto._overlap_parts(513, 480, 640) == 1 and auto._overlap_parts(4096, 96, 128) == 1
    assert LMOptimizer({"camera_model": model})._overlap_parts(1024, 480, 640) == 1            # early stop
    # ... and asks the LIBRARY whether the parts are cut like the batch (gclm_plan_cut), instead of mirroring its rule
    from geocalib_amd import _lib
    lib, C = _lib.load(), _lib.C
    h = auto._handle(dev)

    def cut(n, hh=480, ww=640):
        rows, chunks = C.c_int(0), C.c_int(0)
        assert lib.gclm_plan_cut(h.ptr, n, hh, ww, 1, C.byref(rows), C.byref(chunks)) == 0
        return rows.value, chunks.value
    # 640x480: two rows per loop iteration, 20 (pinhole) / 30 iterations per chunk; below 2048 chunks per call the
    # library takes fewer rows per chunk -- where that regime ends depends on the camera model (pinhole: 137 images of 15
    # chunks, the others: 205 of 10), which is why the Python side no longer mirrors the rule
    assert cut(1024) == cut(512) == cut(256) and cut(1)[0] < cut(1024)[0]
    assert cut(1024) == ((40, 15) if model == "pinhole" else (60, 10))
    assert (cut(137) == cut(1024)) == (model == "pinhole") and cut(205) == cut(1024)
    assert auto._overlap_parts(1024, 480, 640, h, True) == 2
    assert lib.gclm_set_sweep_iters(h.ptr, 7) == 0 and cut(1024)[0] == 14 and lib.gclm_set_sweep_iters(h.ptr, 0) == 0
    assert lib.gclm_plan_cut(h.ptr, 0, 480, 640, 1, None, None) == -3 and lib.gclm_plan_cut(None, 1, 480, 640, 1, None, None) == -1
    # gclm_merge_stop_at refuses a part whose handle last solved another batch size (its counters are not that part's)
    hs = list(opt2._handles.values())[-2:]
    info = torch.zeros((B, _lib.INFO_STRIDE), device=dev)
    parts = (C.c_void_p * 2)(*[x.ptr.value for x in hs])
    infos = (C.c_void_p * 2)(info[:256].data_ptr(), info[256:].data_ptr())
    good = (C.c_int * 2)(*[B * 1 // 2, B - B // 2])
    stream = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize()
    assert lib.gclm_merge_stop_at(parts, infos, good, 2, stream) == 0, _lib.last_error(hs[0].ptr)
    bad = (C.c_int * 2)(B // 2, B - B // 2 - 1)
    assert lib.gclm_merge_stop_at(parts, infos, bad, 2, stream) == -2 and "last solve" in _lib.last_error(hs[0].ptr)
    torch.cuda.synchronize()
