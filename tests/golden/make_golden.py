"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_golden.py

For every case the reference's own `geocalib.lm_optimizer.LMOptimizer` (CPU, float32, eval mode,
torch.no_grad) is run on seeded synthetic fields (oracle/synth.py) and its outputs are stored.
Small cases also store their inputs; the 640x480 cases store only outputs plus an input checksum
(the inputs are regenerated from the seed by the tests).  Files:

    inputs_<model>.npz     committed inputs of the small (64x96) sets
    golden_small.npz       reference outputs for every (set, conf-variant) on the small sets
    golden_trace.npz       per-step Grad/Hess/delta/lambda/cost of the reference on the small sets
    golden_system.npz      single-pass costs / J^T W r / J^T W J at fixed parameters
    golden_full.npz        reference outputs for the 640x480 BASELINE configurations (B=4 each)
    golden_cnn.npz         fields of the (randomly initialised, seeded) reference CNN on
                           assets/pinhole-church.jpg + the reference LM result (BASELINE config 1)
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 1234
SMALL = (64, 96)
FULL = (480, 640)
BENCH = {"num_steps": 20, "early_stop": False}

ref = ref_import.load()
torch.set_num_threads(os.cpu_count())


def to_t(data):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in data.items()}


def run_reference(data, conf, training=False):
    opt = ref.lm_optimizer.LMOptimizer(dict(conf))
    opt = opt.train() if training else opt.eval()
    with torch.no_grad():
        out = opt(to_t(data))
    res = {}
    for k, v in out.items():
        if k in ("camera", "gravity"):
            res[k] = v._data.numpy().copy()
        elif torch.is_tensor(v):
            res[k] = v.numpy().copy()
        else:
            res[k] = np.asarray(v)
    return res


def run_reference_trace(data, conf, steps):
    """Re-run the reference's own loop body (lm_optimizer.py:576-627) step by step, recording."""
    opt = ref.lm_optimizer.LMOptimizer(dict(conf)).eval()
    td = to_t(data)
    rec = {k: [] for k in ("G", "H", "delta", "lambda", "cost_up", "cost_lat", "camera", "gravity")}
    with torch.no_grad():
        cam, grav = ref.lm_optimizer.get_trivial_estimation(td, opt.camera_model)
        opt.setup_optimization_and_priors(td, shared_intrinsics=opt.shared_intrinsics)
        B = td["up_field"].shape[0]
        lamb = torch.ones(1 if opt.shared_intrinsics else B) * opt.conf.lambda_
        prev = None
        for i in range(steps):
            err = opt.calculate_residuals(cam, grav, td)
            costs, w = opt.calculate_costs(err, td)
            if i == 0:
                prev = sum(c.mean(-1) for c in costs.values())
            G, H = opt.setup_system(cam, grav, err, w, shared_intrinsics=opt.shared_intrinsics)
            delta = ref.lm_optimizer.optimizer_step(G, H, lamb)
            rec["G"].append(G.numpy().copy())
            rec["H"].append(H.numpy().copy())
            rec["delta"].append(delta.numpy().copy())
            rec["lambda"].append(lamb.numpy().copy())
            rec["cost_up"].append(costs["up_cost"].mean(-1).numpy().copy())
            rec["cost_lat"].append(costs["latitude_cost"].mean(-1).numpy().copy())
            if opt.shared_intrinsics:
                ni = opt.n_intrinsic_params
                dg = delta[..., :-ni].reshape(B, 2)
                df = delta[..., -ni].expand(B, 1)
                dd = delta[..., -ni + 1:].expand(B, -1)
                delta = torch.cat([dg, df, dd], -1)
            cam, grav = opt.update_estimate(cam, grav, delta)
            new, _ = opt.calculate_costs(opt.calculate_residuals(cam, grav, td), td)
            new = sum(c.mean(-1) for c in new.values())
            if not opt.conf.fix_lambda and not opt.shared_intrinsics:
                lamb = ref.lm_optimizer.update_lambda(lamb, prev, new)
            prev = new
            rec["camera"].append(cam._data.numpy().copy())
            rec["gravity"].append(grav._data.numpy().copy())
    return {k: np.stack(v) for k, v in rec.items()}


def run_reference_system(data, conf, cam8, grav3, as_rpf):
    opt = ref.lm_optimizer.LMOptimizer(dict(conf)).eval()
    td = to_t(data)
    with torch.no_grad():
        cam = opt.camera_model(torch.from_numpy(cam8))
        grav = ref.gravity.Gravity(torch.from_numpy(grav3))
        opt.setup_optimization_and_priors(td, shared_intrinsics=False)
        err = opt.calculate_residuals(cam, grav, td)
        costs, w = opt.calculate_costs(err, td)
        G, H = opt.setup_system(cam, grav, err, w, as_rpf=as_rpf)
    return {"G": G.numpy(), "H": H.numpy(), "cost_up": costs["up_cost"].mean(-1).numpy(),
            "cost_lat": costs["latitude_cost"].mean(-1).numpy()}


def checksum(data):
    return np.array([np.float64(np.asarray(v, np.float64).sum()) for _, v in sorted(data.items())])


def small_sets():
    sets = {}
    for model, B in (("pinhole", 6), ("simple_radial", 6), ("radial", 4), ("simple_divisional", 4)):
        data, cams, gravs = synth.make_fields(SEED, range(B), model, *SMALL)
        sets[model] = (data, cams, gravs)
    return sets


def shared_set(model, B=4):
    """Frames of one camera: same intrinsics (those of index 100), different gravities."""
    cam0, _, _ = synth.gt_params(SEED, 100, model, *SMALL)
    cams = np.stack([cam0] * B)
    gravs = np.stack([synth.gt_params(SEED, 100 + i, model, *SMALL)[1] for i in range(B)])
    up, lat = synth.lm_oracle.render(model, *SMALL, cams, gravs, precision="f64")
    rng = np.random.default_rng([SEED, 100, 7])
    up += rng.normal(0, 0.02, up.shape).astype(np.float32)
    lat += rng.normal(0, 0.02, lat.shape).astype(np.float32)
    up /= np.sqrt((up.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    data = {"up_field": up, "latitude_field": lat,
            "up_confidence": rng.uniform(0, 1, (B,) + SMALL).astype(np.float32),
            "latitude_confidence": rng.uniform(0, 1, (B,) + SMALL).astype(np.float32)}
    return data, cams.astype(np.float32), gravs.astype(np.float32)


def variants(model, data, cams, gravs):
    """(name, conf, data) triples exercised on a small set."""
    nd = {"pinhole": 0, "simple_radial": 1, "radial": 2, "simple_divisional": 1}[model]
    base = {"camera_model": model, **BENCH}
    noconf = {k: v for k, v in data.items() if "confidence" not in k}
    v = [("bench", base, data),
         ("default", {"camera_model": model}, data),
         ("noconf", base, noconf)]
    if model in ("pinhole", "simple_radial"):
        # note: an input without latitude_field raises KeyError in the reference
        # (lm_optimizer.py:31 evaluates data["latitude_field"] eagerly), so there is no "up_only".
        v += [("lat_only", base, {k: data[k] for k in ("latitude_field", "latitude_confidence")}),
              ("euclid", {**base, "use_spherical_manifold": False}, data),
              ("linfocal", {**base, "use_log_focal": False}, data),
              ("fixlambda", {**base, "fix_lambda": True}, data),
              ("scales", base, {**data, "scales": np.array([0.5, 0.6], np.float32)}),
              ("prior_focal", base, {**data, "prior_focal": cams[:, 3].copy()}),
              ("prior_gravity", base, {**data, "prior_gravity": gravs.copy()}),
              ("loss_scale", {**base, "up_loss_fn_scale": 5e-2, "lat_loss_fn_scale": 2e-2}, data)]
        # note: "prior_dist" cannot be exercised: camera.py:74-92 stacks k1 of shape (B,1) with
        # (B,) tensors and raises for every batched prior_dist.
    return v


def main():
    sets = small_sets()
    small, trace, system = {}, {}, {}
    for model, (data, cams, gravs) in sets.items():
        np.savez_compressed(os.path.join(HERE, f"inputs_{model}.npz"), gt_camera=cams,
                            gt_gravity=gravs, **data)
        for name, conf, d in variants(model, data, cams, gravs):
            out = run_reference(d, conf)
            for k, val in out.items():
                small[f"{model}/{name}/{k}"] = val
            print(model, name, out["camera"][0, 2:4], out["stop_at"][0])
        out = run_reference(data, {"camera_model": model, **BENCH}, training=True)
        small[f"{model}/training/camera"] = out["camera"]
        small[f"{model}/training/keys"] = np.array(sorted(out.keys()))
        if model in ("pinhole", "simple_radial"):
            tr = run_reference_trace(data, {"camera_model": model, **BENCH}, steps=6)
            for k, val in tr.items():
                trace[f"{model}/{k}"] = val
        # single pass at perturbed GT parameters
        pc, pg = cams.copy(), gravs.copy()
        pc[:, 2:4] *= 1.07
        pg += np.float32(0.05) * np.array([1, -1, 1], np.float32)
        pg /= np.linalg.norm(pg, axis=1, keepdims=True)
        if model != "pinhole":
            pc[:, 6] += np.float32(0.03)
        for as_rpf in (False, True):
            s = run_reference_system(data, {"camera_model": model}, pc, pg, as_rpf)
            for k, val in s.items():
                system[f"{model}/{'rpf' if as_rpf else 'loop'}/{k}"] = val
        system[f"{model}/camera"], system[f"{model}/gravity"] = pc, pg
    # shared intrinsics (whole batch = one group, lm_optimizer.py:350-383)
    for model in ("pinhole", "simple_radial"):
        data, cams, gravs = shared_set(model)
        np.savez_compressed(os.path.join(HERE, f"inputs_shared_{model}.npz"), gt_camera=cams,
                            gt_gravity=gravs, **data)
        out = run_reference(data, {"camera_model": model, "shared_intrinsics": True, **BENCH})
        for k, val in out.items():
            small[f"shared_{model}/bench/{k}"] = val
        tr = run_reference_trace(data, {"camera_model": model, "shared_intrinsics": True, **BENCH}, 4)
        for k, val in tr.items():
            trace[f"shared_{model}/{k}"] = val
        print("shared", model, out["camera"][:, 2:4].ravel(), cams[0, 2])
    np.savez_compressed(os.path.join(HERE, "golden_small.npz"), **small)
    np.savez_compressed(os.path.join(HERE, "golden_trace.npz"), **trace)
    np.savez_compressed(os.path.join(HERE, "golden_system.npz"), **system)

    # ---- BASELINE.json configs 2/4 at full size (B=4 each): outputs + input checksums only
    full = {}
    for model in ("pinhole", "simple_radial"):
        data, cams, gravs = synth.make_fields(SEED, range(4), model, *FULL)
        out = run_reference(data, {"camera_model": model, **BENCH})
        for k, val in out.items():
            full[f"{model}/{k}"] = val
        full[f"{model}/input_checksum"] = checksum(data)
        full[f"{model}/gt_camera"], full[f"{model}/gt_gravity"] = cams, gravs
        print("full", model, out["camera"][:, 2], cams[:, 2])
    np.savez_compressed(os.path.join(HERE, "golden_full.npz"), **full)

    # ---- BASELINE.json config 1 restated: reference CNN, seeded random init
    from PIL import Image
    gc = importlib.import_module("geocalib.geocalib")
    torch.manual_seed(0)
    model = gc.GeoCalib().eval()
    img = Image.open(os.path.join(ref_import.REFERENCE_ROOT, "assets", "pinhole-church.jpg"))
    x = torch.from_numpy(np.asarray(img.convert("RGB"))).permute(2, 0, 1).float() / 255
    h, w = x.shape[-2:]
    s = 160 / min(h, w)
    x = torch.nn.functional.interpolate(x[None], size=(round(h * s), round(w * s)), mode="bilinear",
                                        antialias=True, align_corners=False)
    H, W = x.shape[-2:]
    H32, W32 = H // 32 * 32, W // 32 * 32
    x = x[..., (H - H32) // 2:(H - H32) // 2 + H32, (W - W32) // 2:(W - W32) // 2 + W32]
    with torch.no_grad():
        feats = {"hl": model.backbone({"image": x})["features"], "ll": model.ll_enc({"image": x})["features"]}
        fields = model.perspective_decoder({"features": feats})
    data = {k: fields[k].numpy().copy() for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")}
    cnn = dict(data)
    for name, conf in (("default", {}), ("bench", BENCH)):
        out = run_reference(data, conf)
        for k, val in out.items():
            cnn[f"{name}/{k}"] = val
        print("cnn", name, out["camera"], out["gravity"], out["stop_at"])
    np.savez_compressed(os.path.join(HERE, "golden_cnn.npz"), **cnn)


if __name__ == "__main__":
    main()
