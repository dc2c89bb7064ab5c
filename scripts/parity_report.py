"""GPU box: measured HIP-vs-reference differences on the committed goldens (what the tolerances of the suite cover)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import conf_for, data_for, golden_cases, golden_outputs
from geocalib_amd import LMOptimizer
dev = torch.device("cuda:0")
worst = {}
for setname, variant in golden_cases(("pinhole", "simple_radial", "radial", "simple_divisional")):
    ref = golden_outputs(setname, variant)
    conf, data = conf_for(setname, variant), data_for(setname, variant)
    out = LMOptimizer(conf).eval()({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in data.items()})
    cam, grav = out["camera"]._data.cpu().numpy(), out["gravity"]._data.cpu().numpy()
    f = np.abs(cam[:, 2:4] / ref["camera"][:, 2:4] - 1).max()
    k = np.abs(cam[:, 6:] - ref["camera"][:, 6:]).max()
    g = np.abs(grav - ref["gravity"]).max()
    c = np.abs(out["final_cost"].cpu().numpy() - ref["final_cost"]).max() / np.abs(ref["final_cost"]).max()
    model = setname.replace("shared_", "")
    w = worst.setdefault(model, np.zeros(4))
    worst[model] = np.maximum(w, [f, k, g, c])
for m, w in worst.items():
    print(f"{m:18s} worst over its golden cases: focal rel {w[0]:.1e}  dist abs {w[1]:.1e}  gravity abs {w[2]:.1e}  final cost rel {w[3]:.1e}")
