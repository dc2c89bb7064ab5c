"""CPU: the host-side mirror of the reference API (Camera / Gravity / manifolds / fields / LMOptimizer
configuration logic).  Analytic Jacobians are checked against autograd like the reference's only test
file (siclib/geometry/gradient_checker.py); when /root/reference is present the classes are also
compared method-by-method with the reference's."""
import math

import numpy as np
import pytest
import torch
from torch.func import jacfwd, vmap

from geocalib_amd import Gravity, LMOptimizer, camera_models, get_trivial_estimation
from geocalib_amd import misc, perspective_fields as pf
from geocalib_amd.utils import deg2rad, focal2fov, fov2focal, rad2rotmat

MODELS = ["pinhole", "simple_radial", "radial", "simple_divisional"]


def make(model, B=3, H=48, W=64, dtype=torch.float64):
    d = {"height": torch.full((B,), float(H), dtype=dtype), "width": torch.full((B,), float(W), dtype=dtype),
         "vfov": torch.tensor([0.9, 1.2, 0.6], dtype=dtype)[:B]}
    if model != "pinhole":
        d["k1"] = torch.tensor([-0.1, 0.05, -0.2], dtype=dtype)[:B]
    if model == "radial":
        d["k2"] = torch.tensor([0.01, 0.0, 0.02], dtype=dtype)[:B]
    cam = camera_models[model].from_dict(d)
    grav = Gravity.from_rp(torch.tensor([0.2, -0.3, 2.5], dtype=dtype)[:B], torch.tensor([0.1, 0.4, -0.6], dtype=dtype)[:B])
    return cam, grav


def test_conversions():
    f = fov2focal(torch.tensor(math.radians(60.0)), torch.tensor(480.0))
    assert f.item() == pytest.approx(240 / math.tan(math.radians(30)), rel=1e-6)
    assert focal2fov(f, torch.tensor(480.0)).item() == pytest.approx(math.radians(60), rel=1e-6)
    assert deg2rad(180.0) == pytest.approx(math.pi)
    R = rad2rotmat(torch.tensor([0.3]), torch.tensor([-0.2]), torch.tensor([0.5]))
    assert torch.allclose(R @ R.transpose(-1, -2), torch.eye(3)[None], atol=1e-6)


def test_small_utils():
    """The small public helpers of geocalib/utils.py:217-309 and perspective_fields.get_horizon_line (:18-44) that the
    reference's demos / viz import: definitions in float64, and against the reference where it is mounted."""
    from geocalib_amd.utils import get_device, pitch2rho, rho2pitch, skew_symmetric
    v = torch.tensor([[0.3, -1.2, 2.0], [1.0, 0.5, -0.7]], dtype=torch.float64)
    w = torch.tensor([[-0.4, 0.9, 0.1], [0.2, 0.2, 2.0]], dtype=torch.float64)
    S = skew_symmetric(v)
    assert S.shape == (2, 3, 3) and torch.allclose(S, -S.transpose(-1, -2))
    assert torch.allclose((S @ w[..., None])[..., 0], torch.linalg.cross(v, w))
    pitch, f, h = torch.tensor([0.3, -0.5]), torch.tensor([500.0, 320.0]), torch.tensor([480.0, 480.0])
    assert torch.allclose(rho2pitch(pitch2rho(pitch, f, h), f, h), pitch, atol=1e-6)
    assert torch.allclose(pitch2rho(pitch, f, h), torch.tan(pitch) * f / h)
    assert get_device() in ("cuda", "mps", "cpu") and (get_device() == "cuda") == torch.cuda.is_available()
    # horizon line: a level camera sees it at the principal point's height; roll tilts it around the projected midpoint
    cam, _ = make("pinhole", B=1)
    flat = pf.get_horizon_line(cam, Gravity.from_rp(torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)))
    assert torch.allclose(flat, torch.full((2,), 0.5, dtype=torch.float64), atol=1e-9)
    g = Gravity.from_rp(torch.tensor([0.2], dtype=torch.float64), torch.tensor([0.1], dtype=torch.float64))
    hl, hl_px = pf.get_horizon_line(cam, g), pf.get_horizon_line(cam, g, relative=False)
    assert torch.allclose(hl * cam.size[0, 1], hl_px)
    assert ((hl_px[0] - hl_px[1]) / cam.size[0, 0]).item() == pytest.approx(math.tan(0.2), rel=3e-4)   # slope = tan(roll); Gravity.roll carries the 1e-4 guard of gravity.py:66
    from oracle import ref_import
    if ref_import.available():
        ref = ref_import.load()
        # (the reference's own get_horizon_line raises an IndexError on every input -- it indexes the batch dimension it
        # has just added, perspective_fields.py:29-36 -- so only its definition can be followed, not its output)
        assert torch.allclose(ref.utils.skew_symmetric(v), S) and torch.allclose(ref.utils.pitch2rho(pitch, f, h), pitch2rho(pitch, f, h))


def test_gravity_roundtrip_and_update():
    roll, pitch = torch.tensor([0.3, -1.0, 2.8, -2.9]), torch.tensor([0.2, -0.7, 0.5, 1.0])
    g = Gravity.from_rp(roll, pitch)
    assert torch.allclose(g.vec3d.norm(dim=-1), torch.ones(4), atol=1e-6)
    assert torch.allclose(g.roll, roll, atol=2e-4) and torch.allclose(g.pitch, pitch, atol=1e-5)
    # spherical update: stays on the sphere, moves by |delta|, zero delta is the identity
    d = torch.tensor([[0.05, -0.02]] * 4)
    g2 = g.update(d, spherical=True)
    assert torch.allclose(g2.vec3d.norm(dim=-1), torch.ones(4), atol=1e-6)
    ang = torch.acos((g2.vec3d * g.vec3d).sum(-1).clamp(-1, 1))
    assert torch.allclose(ang, d.norm(dim=-1), atol=1e-4)
    assert torch.allclose(g.update(torch.zeros(4, 2), spherical=True).vec3d, g.vec3d, atol=1e-6)
    g3 = g.update(d, spherical=False)
    assert torch.allclose(g3.roll, roll + 0.05, atol=5e-4) and torch.allclose(g3.pitch, pitch - 0.02, atol=1e-5)


def test_spherical_manifold_jacobian_matches_autograd():
    x = torch.nn.functional.normalize(torch.randn(6, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0)), dim=-1)
    J = misc.SphericalManifold.J_plus(x)
    Jad = vmap(jacfwd(lambda d, xx: misc.SphericalManifold.plus(xx[None], d[None])[0]))(torch.zeros(6, 2, dtype=torch.float64), x)
    assert torch.allclose(J, Jad, atol=1e-6)
    # columns span the tangent plane and are orthonormal
    assert torch.allclose((J * x[..., None]).sum(-2), torch.zeros(6, 2, dtype=torch.float64), atol=1e-9)
    assert torch.allclose(J.transpose(-1, -2) @ J, torch.eye(2, dtype=torch.float64).expand(6, 2, 2), atol=1e-9)


def test_vecnorm_and_focal2fov_jacobians():
    v = torch.randn(5, 3, dtype=torch.float64)
    Jad = vmap(jacfwd(lambda t: t / t.norm()))(v)
    assert torch.allclose(misc.J_vecnorm(v), Jad, atol=1e-9)
    f, h = torch.tensor([300.0, 800.0], dtype=torch.float64), torch.tensor([480.0, 480.0], dtype=torch.float64)
    Jad = vmap(jacfwd(lambda ff, hh: focal2fov(ff, hh)))(f, h)
    assert torch.allclose(misc.J_focal2fov(f, h), Jad, atol=1e-12)


@pytest.mark.parametrize("model", MODELS[1:])
def test_distortion_jacobians_match_autograd(model):
    cam, _ = make(model)
    pts = (torch.rand(3, 40, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(1)) - 0.5) * 0.9
    for b in range(3):
        c = cam[b:b + 1]
        Jd = vmap(jacfwd(lambda p: c.distort(p[None, None])[0][0, 0]))(pts[b])
        assert torch.allclose(c.J_distort(pts[b:b + 1], "pts")[0], Jd, atol=1e-7), model
        Ju = vmap(jacfwd(lambda p: c.undistort(p[None, None])[0][0, 0]))(pts[b])
        assert torch.allclose(c.J_undistort(pts[b:b + 1], "pts")[0], Ju, atol=1e-7), model
        Js = vmap(jacfwd(lambda p: c.distort(p[None, None], return_scale=True)[0][0, 0, 0]))(pts[b])
        assert torch.allclose(c.up_projection_offset(pts[b:b + 1])[0], Js, atol=1e-7), model


@pytest.mark.parametrize("model", MODELS)
def test_distort_and_undistort_accept_both_reference_keywords(model):
    """The reference names the argument `pts` on BaseCamera.distort / undistort (camera.py:212,242) and `p2d` on the
    distortion models (camera.py:611,631,712,737,829,863; Pinhole: distort(p2d), undistort(pts)).  One body serves the
    distortion models here, so both keywords and the positional form give the same result; passing none or both is an error."""
    cam, _ = make(model)
    pts = (torch.rand(3, 10, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(7)) - 0.5) * 0.8
    d0, v0 = cam.distort(pts)
    u0, w0 = cam.undistort(pts)
    if model == "pinhole":
        assert torch.equal(cam.distort(p2d=pts)[0], d0) and torch.equal(cam.undistort(pts=pts)[0], u0)
        assert torch.equal(d0, pts) and torch.equal(u0, pts)
        return
    for kw in ("pts", "p2d"):
        d, v = cam.distort(**{kw: pts})
        u, w = cam.undistort(**{kw: pts})
        assert torch.equal(d, d0) and torch.equal(v, v0) and torch.equal(u, u0) and torch.equal(w, w0), (model, kw)
        assert torch.equal(cam.distort(return_scale=True, **{kw: pts})[0], cam.distort(pts, True)[0])
    with pytest.raises(TypeError):
        cam.distort()
    with pytest.raises(TypeError):
        cam.undistort(pts, p2d=pts)


@pytest.mark.parametrize("model", MODELS[1:])
def test_up_projection_offset_jacobians(model):
    """J_up_projection_offset / J_distort(scale2dist): the closed forms of the polynomial models and the generic
    autograd path of BaseCamera (the only one simple_divisional has) agree with each other and with jacfwd."""
    from geocalib_amd.camera import BaseCamera
    cam, _ = make(model)
    pts = (torch.rand(3, 25, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(3)) - 0.5) * 0.8
    for wrt in ("uv", "dist"):
        pub, gen = cam.J_up_projection_offset(pts, wrt), BaseCamera.J_up_projection_offset(cam, pts, wrt)
        assert pub.shape == gen.shape == (3, 25, 2, 2 if wrt == "uv" else cam.num_dist_params())
        assert torch.allclose(pub, gen, atol=1e-10), (model, wrt)
    for b in range(3):
        c = cam[b:b + 1]
        J = vmap(jacfwd(lambda p: c.up_projection_offset(p[None, None])[0, 0]))(pts[b])
        assert torch.allclose(cam.J_up_projection_offset(pts, "uv")[b], J, atol=1e-9), model
    assert torch.allclose(cam.J_distort(pts, "scale2dist"), BaseCamera.J_distort(cam, pts, "scale2dist"), atol=1e-10)
    with pytest.raises(NotImplementedError):
        cam.J_up_projection_offset(pts, "focal")


@pytest.mark.parametrize("model", MODELS)
def test_camera_bookkeeping(model):
    cam, _ = make(model, dtype=torch.float32)
    s = cam.scale((0.5, 0.25))
    assert torch.allclose(s.size, cam.size * torch.tensor([0.5, 0.25])) and torch.allclose(s.f, cam.f * torch.tensor([0.5, 0.25]))
    c = cam.crop((-4.0, 6.0))
    assert torch.allclose(c.size, cam.size + torch.tensor([-4.0, 6.0])) and torch.allclose(c.c, cam.c + torch.tensor([-2.0, 3.0]))
    back = s.undo_scale_crop({"scales": torch.tensor([0.5, 0.25])})
    assert torch.allclose(back._data[:, :6], cam._data[:, :6], atol=1e-4)
    # focal clamp: both bounds come from the image HEIGHT (camera.py:141-145)
    lo, hi = 24 / math.tan(math.radians(75)), 24 / math.tan(math.radians(2.5))
    assert torch.allclose(cam.update_focal(torch.full((3, 1), -50.0), as_log=True).f[:, 1], torch.full((3,), lo), rtol=1e-5)
    assert torch.allclose(cam.update_focal(torch.full((3, 1), 50.0), as_log=True).f[:, 1], torch.full((3,), hi), rtol=1e-5)
    up = cam.update_focal(torch.full((3, 1), 0.1), as_log=True)
    assert torch.allclose(up.f[:, 1], cam.f[:, 1] * math.exp(0.1), rtol=1e-5)
    xy = cam.pixel_coordinates()
    assert xy.shape == (3, 48 * 64, 2) and xy[0, 1].tolist() == [1.0, 0.0] and xy[0, 64].tolist() == [0.0, 1.0]
    if model != "pinhole":
        lim = 3.0 if model == "simple_divisional" else 0.7
        assert cam.update_dist(torch.full((3, 1), 9.0)).k1.tolist() == pytest.approx([lim] * 3)
        p = (torch.rand(3, 10, 2, generator=torch.Generator().manual_seed(4)) - 0.5) * 0.2
        rt = cam.undistort(cam.distort(p)[0])[0]
        assert torch.allclose(rt, p, atol=2e-3 if model != "simple_divisional" else 1e-4)   # fp32 cancellation in 1 - sqrt(1 - 4 k r2)


@pytest.mark.parametrize("model", MODELS)
def test_perspective_field_properties(model):
    cam, grav = make(model, dtype=torch.float64)
    up, lat = pf.get_perspective_field(cam, grav)
    assert up.shape == (3, 2, 48, 64) and lat.shape == (3, 1, 48, 64)
    assert torch.allclose(up.norm(dim=1), torch.ones(3, 48, 64, dtype=torch.float64), atol=1e-9)
    # at the principal point (u = v = 0) the up vector is (a, b)/|(a, b)| and sin(lat) = c
    g = grav.vec3d
    assert torch.allclose(up[:, :, 24, 32], torch.nn.functional.normalize(g[:, :2], dim=-1), atol=1e-9)
    assert torch.allclose(torch.sin(lat[:, 0, 24, 32]), g[:, 2], atol=1e-9)


def test_reference_api_comparison():
    """Method-by-method against the reference classes (only where /root/reference exists)."""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference not mounted")
    ref = ref_import.load()
    for model in MODELS:
        cam, grav = make(model, dtype=torch.float32)
        rc, rg = ref.camera.camera_models[model](cam._data.clone()), ref.gravity.Gravity(grav._data.clone())
        pts = (torch.rand(3, 30, 2, generator=torch.Generator().manual_seed(2)) - 0.5)
        pairs = [(rg.roll, grav.roll), (rg.J_rp(), grav.J_rp()), (rg.R, grav.R), (rc.K, cam.K), (rc.vfov, cam.vfov),
                 (rg.update(torch.full((3, 2), 0.05), True)._data, grav.update(torch.full((3, 2), 0.05), True)._data),
                 (rc.update_focal(torch.full((3, 1), 0.3), True)._data, cam.update_focal(torch.full((3, 1), 0.3), True)._data),
                 (rc.image2world(pts * 60)[0], cam.image2world(pts * 60)[0]),
                 (rc.J_image2world(pts * 60, "f"), cam.J_image2world(pts * 60, "f")),
                 (rc.pixel_coordinates(), cam.pixel_coordinates())]
        pairs += list(zip(ref.perspective_fields.get_perspective_field(rc, rg), pf.get_perspective_field(cam, grav)))
        if model != "pinhole":
            pairs += [(rc.distort(pts)[0], cam.distort(pts)[0]), (rc.undistort(pts)[0], cam.undistort(pts)[0]),
                      (rc.up_projection_offset(pts), cam.up_projection_offset(pts)),
                      (rc.J_undistort(pts, "dist"), cam.J_undistort(pts, "dist")),
                      (rc.J_distort(pts, "scale2dist"), cam.J_distort(pts, "scale2dist")),
                      (rc.J_up_projection_offset(pts, "uv"), cam.J_up_projection_offset(pts, "uv")),
                      (rc.J_up_projection_offset(pts, "dist"), cam.J_up_projection_offset(pts, "dist"))]
        # simple_divisional in float32: the reference's own closed forms cancel (flagged unstable at camera.py:913)
        atol = 5e-4 if model == "simple_divisional" else 2e-5
        for i, (a, b) in enumerate(pairs):
            assert a.shape == b.shape and torch.allclose(a, b, atol=atol, rtol=1e-5), (model, i)


def test_trivial_estimation_and_plan():
    data = {"up_field": torch.zeros(2, 2, 480, 640), "latitude_field": torch.zeros(2, 1, 480, 640)}
    cam, grav = get_trivial_estimation(data, camera_models["simple_radial"])
    assert cam._data.shape == (2, 8) and cam._data[0, :2].tolist() == [640.0, 480.0]
    assert cam._data[0, 2].item() == pytest.approx(448.0, rel=1e-6) and cam._data[0, 4:].tolist() == [320.0, 240.0, 0.0, 0.0]
    assert torch.allclose(grav.vec3d, torch.tensor([[0.0, -1.0, 0.0]] * 2), atol=1e-7)
    cam2, _ = get_trivial_estimation({**data, "scales": torch.tensor([0.5, 0.6])}, camera_models["pinhole"])
    assert cam2._data[0, 2].item() == pytest.approx(448.0 * 0.5 / 0.6, rel=1e-6)
    with pytest.raises(KeyError):          # like lm_optimizer.py:31
        get_trivial_estimation({"up_field": torch.zeros(1, 2, 8, 8)}, camera_models["pinhole"])
    opt = LMOptimizer({"camera_model": "simple_radial"})
    assert (opt.gravity_delta_dims, opt.focal_delta_dims, opt.dist_delta_dims, opt.n_intrinsic_params) == ((0, 1), (2,), (3,), 2)
    opt.setup_optimization_and_priors({"prior_gravity": 0})
    assert (opt.gravity_delta_dims, opt.focal_delta_dims, opt.dist_delta_dims) == ((-1,), (0,), (1,))
    opt.setup_optimization_and_priors({"prior_focal": 0})          # the reference's overlap quirk
    assert (opt.gravity_delta_dims, opt.focal_delta_dims, opt.dist_delta_dims) == ((0, 1), (-1,), (0,))
    assert LMOptimizer.default_conf["num_steps"] == 30 and LMOptimizer.default_conf["early_stop"] is True
    with pytest.raises(AssertionError):
        LMOptimizer({"camera_model": "fisheye"})
    cfg = LMOptimizer({"camera_model": "pinhole", "num_steps": 20, "early_stop": False}).eval()._config()
    assert (cfg.num_steps, cfg.early_stop, cfg.compute_uncertainty, cfg.camera_model) == (20, 0, 1, 0)
    assert LMOptimizer({}).train()._config().compute_uncertainty == 0
    assert LMOptimizer({"camera_model": "radial"})._config().camera_model == 2


def test_loop_rules_against_the_reference():
    """update_lambda / early_stop / update_estimate (host forms) against the reference's (only where it is mounted)."""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference not mounted")
    ref = ref_import.load()
    from geocalib_amd import LMOptimizer, lm_optimizer as lm
    gen = torch.Generator().manual_seed(8)
    prev, new, lamb = torch.rand(6, generator=gen), torch.rand(6, generator=gen), torch.rand(6, generator=gen) * 50
    assert torch.equal(lm.update_lambda(lamb, prev, new), ref.lm_optimizer.update_lambda(lamb, prev, new))
    for a, b in ((new, prev), (prev, prev + 1e-10)):
        assert lm.early_stop(a, b, 1e-8, 1e-8) == ref.lm_optimizer.early_stop(a, b, 1e-8, 1e-8)
    for model in MODELS:
        cam, grav = make(model, dtype=torch.float32)
        rc, rg = ref.camera.camera_models[model](cam._data.clone()), ref.gravity.Gravity(grav._data.clone())
        for data in ({}, {"prior_gravity": 0}, {"prior_focal": 0} if model == "pinhole" else {}):
            mine, theirs = LMOptimizer({"camera_model": model}), ref.lm_optimizer.LMOptimizer({"camera_model": model})
            mine.setup_optimization_and_priors(data, shared_intrinsics=False)
            theirs.setup_optimization_and_priors(data, shared_intrinsics=False)
            n = 2 * mine.estimate_gravity + mine.estimate_focal + (cam.num_dist_params() if mine.estimate_dist else 0)
            delta = (torch.rand(3, n, generator=gen) - 0.5) * 0.2
            c1, g1 = mine.update_estimate(cam, grav, delta)
            c2, g2 = theirs.update_estimate(rc, rg, delta)
            assert torch.allclose(c1._data, c2._data, atol=1e-5, rtol=1e-6), (model, data)
            assert torch.allclose(g1._data, g2._data, atol=1e-6), (model, data)


def test_training_mode_warns_that_no_gradient_flows(caplog):
    """The HIP path is inference-only (an opaque C call under no_grad): in training mode, inputs that require gradients
    get ONE warning (upstream's training-time optimiser backpropagates through the solve); the CPU tensors then hit the
    no-CPU-fallback error as always."""
    import logging
    import pytest
    from geocalib_amd import LMOptimizer
    opt = LMOptimizer({"camera_model": "pinhole"}).train()
    data = {"up_field": torch.zeros(1, 2, 8, 8, requires_grad=True), "latitude_field": torch.zeros(1, 1, 8, 8)}
    with caplog.at_level(logging.WARNING, logger="geocalib_amd.lm_optimizer"):
        for _ in range(2):
            with pytest.raises(RuntimeError, match="no CPU fallback"):
                opt(data)
    assert sum("inference-only" in r.message for r in caplog.records) == 1
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="geocalib_amd.lm_optimizer"):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            LMOptimizer({"camera_model": "pinhole"}).eval()(data)
    assert not caplog.records                                   # eval mode: nothing to warn about
