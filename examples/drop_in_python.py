"""Drop-in use from Python (MI355X): the two seams of the reference this package replaces.

    python examples/drop_in_python.py

(1) `GeoCalib.optimizer` (geocalib/geocalib.py:106,119): any code that calls `optimizer(data) -> dict` with the CNN's fields.
(2) `GeoCalib.calibrate(img, camera_model=..., priors=..., shared_intrinsics=...)` (geocalib/extractor.py:72-127) with a
    caller-supplied field network.

The CNN is out of scope of this package, so a stand-in "network" renders the perspective fields of a known camera
(gclm_synth_fields) at the resolution the reference's network would see; with the upstream package installed, pass its
model instead."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geocalib_amd import GeoCalib, LMOptimizer  # noqa: E402
from geocalib_amd.synth import synth_fields  # noqa: E402

dev = torch.device("cuda:0")

# (1) the optimiser seam: fields in, camera / gravity / uncertainties out -- same conf keys, same dict keys as the reference
fields, gt_cam, gt_grav = synth_fields("simple_radial", 4, 480, 640, dev, seed=7)
opt = LMOptimizer({"camera_model": "simple_radial"}).eval()          # default conf: 30 steps, early stop on the device
out = opt(fields)
print("optimizer seam   : focal", [round(v, 1) for v in out["camera"].f[:, 1].tolist()], "(truth", [round(v, 1) for v in gt_cam[:, 3].tolist()], ")")
print("                   k1   ", [round(v, 4) for v in out["camera"].k1.tolist()], " stop_at", out["stop_at"].tolist())
print("                   roll / pitch [deg]", [tuple(round(x, 2) for x in rp) for rp in torch.rad2deg(out["gravity"].rp).tolist()])
print("                   focal_uncertainty", [round(v, 2) for v in out["focal_uncertainty"].tolist()])


# (2) the calibrate() front-end with a stand-in field network
def field_model(img_data):
    b, _, h, w = img_data["image"].shape
    f, _, _ = synth_fields("pinhole", b, h, w, img_data["image"].device, seed=3)
    return f


model = GeoCalib(field_model)                                     # paced_launches=3: calibrate() reads the camera right away
img = torch.rand(3, 768, 1024, device=dev)                        # any RGB image in [0, 1]
res = model.calibrate(img, camera_model="pinhole")
cam, grav = res["camera"], res["gravity"]
print("calibrate()      : image size", cam.size[0].tolist(), " focal", round(cam.f[0, 1].item(), 1), " vfov [deg]", round(torch.rad2deg(cam.vfov)[0].item(), 2))
print("                   roll / pitch [deg]", tuple(round(x, 2) for x in torch.rad2deg(grav.rp)[0].tolist()), " keys:", sorted(k for k in res if "uncertainty" in k))
res = model.calibrate(img, camera_model="pinhole", priors={"focal": torch.tensor(900.0, device=dev)})
print("with a focal prior: focal", round(res["camera"].f[0, 1].item(), 1))
