"""GPU box: degenerate inputs must neither hang nor crash (zero confidences, NaNs, constant fields, tiny images)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")


def show(tag, out):
    cam, g = out["camera"]._data, out["gravity"]._data
    print(f"{tag:28s} f={cam[:, 3].tolist()} g0={[round(v, 4) for v in g[0].tolist()]} stop_at={out['stop_at'].tolist()} "
          f"final={out['final_cost'].tolist()} fails={out.get('step_failures', torch.zeros(1)).tolist()} "
          f"finite_cov={bool(torch.isfinite(out['covariance']).all())}", flush=True)


for model in ("pinhole", "simple_radial"):
    d, _, _ = synth_fields(model, 2, 96, 128, dev, seed=3)
    opt = LMOptimizer({"camera_model": model}).eval()
    show(model + " normal", opt(d))
    z = dict(d); z["up_confidence"] = torch.zeros_like(d["up_confidence"]); z["latitude_confidence"] = torch.zeros_like(d["latitude_confidence"])
    show(model + " zero confidence", opt(z))
    z = dict(d); z["up_confidence"] = d["up_confidence"].clone(); z["up_confidence"][1] = 0; z["latitude_confidence"] = d["latitude_confidence"].clone(); z["latitude_confidence"][1] = 0
    show(model + " image 1 zero conf", opt(z))
    n = dict(d); n["latitude_field"] = d["latitude_field"].clone(); n["latitude_field"][0, 0, 5, 7] = float("nan")
    show(model + " one NaN latitude", opt(n))
    c = dict(d); c["up_field"] = torch.zeros_like(d["up_field"]); c["up_field"][:, 1] = -1; c["latitude_field"] = torch.zeros_like(d["latitude_field"])
    show(model + " constant fields", opt(c))
    h = dict(d); h["up_field"] = d["up_field"] * 1e20
    show(model + " huge up field", opt(h))
    t, _, _ = synth_fields(model, 2, 1, 4, dev, seed=3)
    show(model + " 1x4 image", opt(t))
print("done")
