"""Regenerate draw <case> of the seeded fuzz test (tests/test_gpu_parity.py::test_randomised_configurations_against_oracle)
and print the oracle (float32 / float64) and, on a GPU box, the HIP result.  usage: fuzz_case.py <seed> <case> [n_models]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import synth, lm_oracle as oracle
from conftest import fuzz_draws
seed, want = int(sys.argv[1]), int(sys.argv[2])
for case, model, (H, W), B, data, conf, cams, gravs in fuzz_draws(seed, want + 1, int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    pass
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
