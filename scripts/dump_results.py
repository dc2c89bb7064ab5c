"""GPU box: solve fixed device-generated batches and dump every result tensor (for scripts/ab_bits.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from geocalib_amd import LMOptimizer  # noqa: E402
from geocalib_amd.synth import synth_fields  # noqa: E402

out_path = sys.argv[1]
models = sys.argv[2].split(",") if len(sys.argv) > 2 else ["simple_divisional"]
dev = torch.device("cuda:0")
res = {}
for model in models:
    for (B, H, W) in ((64, 480, 640), (5, 61, 84), (3, 50, 70)):       # BASELINE size, principal point ON a pixel centre (even sizes), scalar path
        data, _, _ = synth_fields(model, B, H, W, dev, seed=77)
        for steps in (1, 20):
            out = LMOptimizer({"camera_model": model, "num_steps": steps, "early_stop": False}).eval()(data)
            torch.cuda.synchronize()
            for k, v in out.items():
                res[f"{model}/{B}x{H}x{W}/{steps}/{k}"] = (v._data if hasattr(v, "_data") else v).cpu().numpy()
    data, _, _ = synth_fields(model, 1, 480, 640, dev, seed=78)          # one image, default conf: one launch per step, early stop
    out = LMOptimizer({"camera_model": model}).eval()(data)
    torch.cuda.synchronize()
    for k, v in out.items():
        res[f"{model}/single/default/{k}"] = (v._data if hasattr(v, "_data") else v).cpu().numpy()
np.savez(out_path, **res)
print(f"{out_path}: {len(res)} tensors")
