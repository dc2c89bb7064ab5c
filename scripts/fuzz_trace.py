"""GPU box: one draw of the seeded fuzz, step by step -- the HIP path and the oracle (float32 / float64) run for k = 1..n
fixed steps, their distance after every step, and where each would stop.  usage: fuzz_trace.py <seed> <case> [n_models] [image]
Shows WHERE two float32 evaluations of the reference algorithm part ways on a draw (a lambda flip, a stall of
simple_divisional's k-column, a stop that fires a step apart)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from conftest import fuzz_draws, result_spread  # noqa: E402
from geocalib_amd import LMOptimizer  # noqa: E402
from oracle import lm_oracle as oracle  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
for case, model, (H, W), B, data, conf, cams, gravs in fuzz_draws(seed, want + 1, int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    pass
img = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # the image whose focal / k / cost are printed
dev = torch.device("cuda:0")
td = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in data.items()}
np.set_printoptions(precision=6, linewidth=220)
print(case, model, (H, W), B, conf, list(data))
print("ground truth f", cams[:, 3], "k", cams[:, 6])


def hip(c):
    out = LMOptimizer(c).eval()(td)
    return {k: (v._data if hasattr(v, "_data") else v).cpu().numpy() for k, v in out.items()}


full_h, full_o = hip(conf), oracle.solve(data, conf, precision="f32", trace=True)
print(f"as drawn: stop_at hip {full_h['stop_at'][img]:.0f} oracle {full_o['stop_at'][img]:.0f}; spread {result_spread(full_h, full_o)}")
tr = full_o["trace"]
print(f"oracle trace, image {img}: cost", (tr["cost_up"][:, img] + tr["cost_lat"][:, img]), "lambda", tr["lambda"][:, img])
for k in range(1, conf["num_steps"] + 1):
    c = {**conf, "num_steps": k, "early_stop": False}
    h, o, o64 = hip(c), oracle.solve(data, c, precision="f32"), oracle.solve(data, c, precision="f64")
    print(f"step {k:2d}: hip f {h['camera'][img, 3]:.5f} k {h['camera'][img, 6]:+.6f} {h['camera'][img, 7]:+.6f} cost {h['final_cost'][img]:.6e} | oracle f {o['camera'][img, 3]:.5f} "
          f"k {o['camera'][img, 6]:+.6f} {o['camera'][img, 7]:+.6f} cost {o['final_cost'][img]:.6e} | hip-o32 {result_spread(h, o)} o32-o64 {result_spread(o, o64)}")
