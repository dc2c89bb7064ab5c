"""GPU box: for one fuzz draw, evaluate the single-sweep system (costs, G, H) of the HIP path and of the oracle AT THE SAME
parameters (the oracle's end point): separates "the sweep differs" from "the trajectories differ"."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import fuzz_draws
from geocalib_amd import LMOptimizer, Gravity, camera_models
from oracle import lm_oracle as oracle
seed, want = int(sys.argv[1]), int(sys.argv[2])
for case, model, (H, W), B, data, conf, cams, gravs in fuzz_draws(seed, want + 1, 4):
    pass
dev = torch.device("cuda:0")
td = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in data.items()}
np.set_printoptions(precision=8, linewidth=200)
o = oracle.solve(data, conf, precision="f32")
opt = LMOptimizer(conf).eval()
out = opt(td)
hip = {k: (v._data if hasattr(v, "_data") else v).cpu().numpy() for k, v in out.items()}
print("final cost hip   ", hip["final_cost"], "\nfinal cost oracle", o["final_cost"], "\nrel", hip["final_cost"] / o["final_cost"] - 1)
print("lat cost hip", hip["final_latitude_cost"], "oracle", o["final_latitude_cost"])
opt.setup_optimization_and_priors(td, shared_intrinsics=False)
for name, cam, grav in (("oracle end point", o["camera"], o["gravity"]), ("hip end point", hip["camera"], hip["gravity"])):
    s = opt.system(td, camera_models[model](torch.from_numpy(cam).to(dev)), Gravity(torch.from_numpy(grav).to(dev)))
    so = oracle.system(data, cam, grav, conf, precision="f32")
    so64 = oracle.system(data, cam, grav, conf, precision="f64")
    cu, cl = s["cost_up"].cpu().numpy(), s["cost_lat"].cpu().numpy()
    print(name, ": sweep cost up hip/oracle-1", cu / np.asarray(so["cost_up"]) - 1, " lat", cl / np.asarray(so["cost_lat"]) - 1,
          " oracle f32/f64-1 (lat)", np.asarray(so["cost_lat"]) / np.asarray(so64["cost_lat"]) - 1)
