"""Regenerate draw <case> of the seeded fuzz test (tests/test_gpu_parity.py::test_randomised_configurations_against_oracle)
and print the oracle (float32 / float64) and, on a GPU box, the HIP result.  usage: fuzz_case.py <seed> <case> [n_models]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import synth, lm_oracle as oracle
ALL_MODELS = ["pinhole", "simple_radial", "radial", "simple_divisional"]
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    model = ALL_MODELS[rng.integers(0, int(sys.argv[3]) if len(sys.argv) > 3 else 3)]
    H, W = int(rng.integers(24, 90)), int(rng.integers(24, 120))
    if rng.random() < 0.1:
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
print(case, model, (H, W), B, conf, list(data.keys()))
r32 = oracle.solve(data, conf, precision="f32")
r64 = oracle.solve(data, conf, precision="f64")
np.set_printoptions(precision=6, linewidth=200)
print("gt f", cams[:, 3], "k", cams[:, 6:8].ravel())
for k in ("stop_at", "final_cost"):
    print(k, "f32", r32[k], "f64", r64[k])
print("f32 cam", r32["camera"][:, [3, 6, 7]].ravel())
print("f64 cam", r64["camera"][:, [3, 6, 7]].ravel())
print("f32 grav", r32["gravity"].ravel()); print("f64 grav", r64["gravity"].ravel())

try:
    import torch
    if torch.cuda.is_available():
        from geocalib_amd import LMOptimizer
        dev = torch.device("cuda:0")
        td = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in data.items()}
        out = LMOptimizer(conf).eval()(td)
        print("HIP cam", out["camera"]._data[:, [3, 6, 7]].cpu().numpy().ravel())
        print("HIP grav", out["gravity"]._data.cpu().numpy().ravel())
        print("HIP stop_at", out["stop_at"].cpu().numpy(), "final_cost", out["final_cost"].cpu().numpy(), "fails", out["step_failures"].cpu().numpy())
except ImportError:
    pass
if len(sys.argv) > 4 and sys.argv[4] == "system":     # compare the single-sweep systems at the HIP end point
    cam_np, grav_np = out["camera"]._data.cpu().numpy(), out["gravity"]._data.cpu().numpy()
    o = oracle.system(data, cam_np, grav_np, conf, precision="f64")
    opt = LMOptimizer(conf).eval()
    opt.setup_optimization_and_priors(td, shared_intrinsics=False)
    s = opt.system(td, out["camera"], out["gravity"])
    np.set_printoptions(precision=5, linewidth=220, suppress=False)
    for b in range(cam_np.shape[0]):
        print("image", b, "G hip", s["G"][b].cpu().numpy(), "G oracle", np.asarray(o["G"][b])[:4])
        print("   H hip diag", np.diag(s["H"][b].cpu().numpy()), "oracle", np.diag(np.asarray(o["H"][b]))[:4])
        print("   H hip row3", s["H"][b, 3].cpu().numpy(), "oracle", np.asarray(o["H"][b])[3][:4])
