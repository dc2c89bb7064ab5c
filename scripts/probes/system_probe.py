"""GPU box: one draw of the seeded fuzz, any camera model -- is the HIP sweep's evaluation AT A GIVEN POINT as accurate as the
oracle's float32 build?  After `steps` LM steps of the HIP path (its own state), the single-sweep system (costs, gradient,
Hessian), the per-pixel residuals and the per-pixel Jacobian rows of the HIP pixel code and of the oracle's float32 build,
each against the oracle's float64 build at the same parameters.
usage: system_probe.py <seed> <case> [steps=1] [image=0]
Separates "the HIP formulas lose accuracy at this point" (its per-pixel errors exceed the float32 oracle's) from "the
trajectory amplifies rounding" (both are at rounding level, the draw is touchy): fuzz 115/45 -- radial at its k1 = -0.7 clamp."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from conftest import fuzz_draws  # noqa: E402
from geocalib_amd import LMOptimizer  # noqa: E402
from geocalib_amd.perspective_fields import J_perspective_field  # noqa: E402
from oracle import lm_oracle as oracle  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
b = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for case, model, (H, W), B, data, conf, cams, gravs in fuzz_draws(seed, want + 1, 4):
    pass
dev = torch.device("cuda:0")
td = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in data.items()}
np.set_printoptions(precision=4, linewidth=240)
print(case, model, (H, W), B, conf)
c1 = {**conf, "num_steps": steps, "early_stop": False}
out = LMOptimizer(c1).eval()(td)
cam, grav = out["camera"], out["gravity"]
cam_np, grav_np = cam._data.cpu().numpy(), grav._data.cpu().numpy()
print(f"HIP state after {steps} step(s), image {b}: camera {cam_np[b]} gravity {grav_np[b]}")
ci = {**c1, "shared_intrinsics": False}
s = LMOptimizer(ci)
s.setup_optimization_and_priors(td)
hs = {k: v.cpu().numpy().astype(np.float64) for k, v in s.eval().system(td, cam, grav).items()}
o32 = oracle.system(data, cam_np, grav_np, ci, precision="f32")
o64 = oracle.system(data, cam_np, grav_np, ci, precision="f64")
for name in ("cost_up", "cost_lat"):
    print(f"{name}: hip/o64 - 1 {hs[name] / o64[name] - 1}   o32/o64 - 1 {np.asarray(o32[name]) / o64[name] - 1}")
sc = np.sqrt(np.abs(np.diag(o64["H"][b])))
print(f"image {b}: (G - G64) / sqrt(H_ii):  hip {(hs['G'][b] - o64['G'][b]) / sc}\n{'':43s}o32 {(o32['G'][b] - o64['G'][b]) / sc}")
eh, eo = (hs["H"][b] - o64["H"][b]) / np.outer(sc, sc), (o32["H"][b] - o64["H"][b]) / np.outer(sc, sc)
print(f"          (H - H64)_ij / sqrt(H_ii H_jj): max |.| hip {np.abs(eh).max():.3e}  o32 {np.abs(eo).max():.3e}\nhip\n{eh}\no32\n{eo}")

# per pixel: residuals and Jacobian rows of image b
rh = s.calculate_residuals(cam, grav, td)
r32 = oracle.residual_fields(model, data, cam_np, grav_np, precision="f32")
r64 = oracle.residual_fields(model, data, cam_np, grav_np, precision="f64")
Jh_up, Jh_lat = J_perspective_field(cam, grav, spherical=bool(c1["use_spherical_manifold"]), log_focal=bool(c1["use_log_focal"]))
assert tuple(Jh_up.shape[1:3]) == (H, W), Jh_up.shape
J32 = oracle.jacobian_fields(model, H, W, cam_np, grav_np, bool(c1["use_spherical_manifold"]), bool(c1["use_log_focal"]), precision="f32")
J64 = oracle.jacobian_fields(model, H, W, cam_np, grav_np, bool(c1["use_spherical_manifold"]), bool(c1["use_log_focal"]), precision="f64")
fx, fy, cx, cy = cam_np[b, 2], cam_np[b, 3], cam_np[b, 4], cam_np[b, 5]
yy, xx = np.mgrid[0:H, 0:W]
r2 = ((xx - cx) / fx) ** 2 + ((yy - cy) / fy) ** 2


def report(name, a_hip, a_32, a_64, scale):
    """a_*: (H, W, C) of image b; errors in units of `scale`"""
    for who, a in (("hip", a_hip), ("o32", a_32)):
        d = np.abs(a - a_64).max(-1) / scale
        idx = np.argsort(d.ravel())[::-1][:4]
        print(f"  {name} {who}: rms error {np.sqrt(np.mean(d ** 2)):.3e}  max {d.max():.3e}  pixels beyond 1e-5: {(d > 1e-5).sum()}  worst "
              + "  ".join(f"({int(i) // W},{int(i) % W}) r2 {r2.ravel()[i]:.3f} err {d.ravel()[i]:.2e}" for i in idx))


print(f"image {b}: per-pixel errors against the float64 oracle (residuals absolute; Jacobian rows relative to the column's rms)")
if "up_residual" in rh:
    report("up residual ", rh["up_residual"][b].cpu().numpy().astype(np.float64).reshape(H, W, 2), r32["up_residual"][b].reshape(H, W, 2),
           r64["up_residual"][b].reshape(H, W, 2), 1.0)
report("lat residual", rh["latitude_residual"][b].cpu().numpy().astype(np.float64).reshape(H, W, 1), r32["latitude_residual"][b].reshape(H, W, 1),
       r64["latitude_residual"][b].reshape(H, W, 1), 1.0)
P = J64[0].shape[-1]
for col in range(P):
    for fld, (jh, j32, j64) in (("J_up ", (Jh_up, J32[0], J64[0])), ("J_lat", (Jh_lat, J32[1], J64[1]))):
        a64 = j64[b, ..., col]
        rms = np.sqrt(np.mean(a64 ** 2)) + 1e-30
        report(f"{fld}[:, {col}]", jh[b].cpu().numpy().astype(np.float64)[..., col].reshape(H, W, -1), j32[b, ..., col].reshape(H, W, -1),
               a64.reshape(H, W, -1), rms)
