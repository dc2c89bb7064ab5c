"""GPU box: simple_divisional at tiny |k| -- the single-sweep system (costs, gradient, Hessian) of the HIP path against the
oracle's float32 and float64 builds AT THE SAME PARAMETERS, for one draw of the seeded fuzz after `steps` LM steps.
usage: div_tiny_k_probe.py <seed> <case> [steps=1] [crop_w]      (crop_w: evaluate the system on the first crop_w columns only, same parameters: 68 = float4 path, 67 = scalar path)
Separates "the two paths evaluate the same point differently" (a formula / rounding issue) from "the trajectory amplifies a
rounding-level difference" (chaos): fuzz 101/270 -- shared intrinsics, 74x71 (scalar path), k = 1.8e-4 after one step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from conftest import fuzz_draws  # noqa: E402
from geocalib_amd import LMOptimizer  # noqa: E402
from oracle import lm_oracle as oracle  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
crop = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for case, model, (H, W), B, data, conf, cams, gravs in fuzz_draws(seed, want + 1, 4):
    pass
dev = torch.device("cuda:0")
td = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in data.items()}
np.set_printoptions(precision=6, linewidth=220)
print(case, model, data["latitude_field"].shape, conf)
c1 = {**conf, "num_steps": steps, "early_stop": False}
opt = LMOptimizer(c1).eval()
out = opt(td)
cam, grav = out["camera"], out["gravity"]
cam_np, grav_np = cam._data.cpu().numpy(), grav._data.cpu().numpy()
o1 = oracle.solve(data, c1, precision="f32")
print("state after", steps, "step(s): hip f", cam_np[:, 3], "k", cam_np[:, 6], "| oracle f", o1["camera"][:, 3], "k", o1["camera"][:, 6])
print("final_cost hip", out["final_cost"].cpu().numpy(), "oracle32", o1["final_cost"])
if crop:        # the SAME parameters (principal point included) on the first `crop` columns: odd width = scalar path, % 4 = float4 path
    data = {k: (np.ascontiguousarray(v[..., :crop]) if v.ndim >= 3 else v) for k, v in data.items()}
    td = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in data.items()}
    print("system on the first", crop, "columns, same parameters")
s = LMOptimizer({**c1, "shared_intrinsics": False})
s.setup_optimization_and_priors(td)
hs = s.eval().system(td, cam, grav)
o32 = oracle.system(data, cam_np, grav_np, {**c1, "shared_intrinsics": False}, precision="f32")
o64 = oracle.system(data, cam_np, grav_np, {**c1, "shared_intrinsics": False}, precision="f64")
for name, h in (("cost_up", hs["cost_up"]), ("cost_lat", hs["cost_lat"])):
    h = h.cpu().numpy().astype(np.float64)
    print(f"{name}: hip/o64 - 1 {h / o64[name] - 1}   o32/o64 - 1 {np.asarray(o32[name]) / o64[name] - 1}")
G, Hm = hs["G"].cpu().numpy().astype(np.float64), hs["H"].cpu().numpy().astype(np.float64)
P = G.shape[1]
for b in range(G.shape[0]):
    sc = np.sqrt(np.abs(np.diag(o64["H"][b])))
    print(f"image {b}: G / sqrt(H_ii): hip - o64 {(G[b] - o64['G'][b]) / sc}   o32 - o64 {(o32['G'][b] - o64['G'][b]) / sc}")
    print(f"          H_ij / sqrt(H_ii H_jj), k row: hip - o64 {((Hm[b] - o64['H'][b]) / np.outer(sc, sc))[P - 1]}   o32 - o64 {((o32['H'][b] - o64['H'][b]) / np.outer(sc, sc))[P - 1]}")
    print(f"          H_kk: hip {Hm[b][P - 1, P - 1]:.6e} o32 {o32['H'][b][P - 1, P - 1]:.6e} o64 {o64['H'][b][P - 1, P - 1]:.6e};  G_k: hip {G[b][P - 1]:.6e} o32 {o32['G'][b][P - 1]:.6e} o64 {o64['G'][b][P - 1]:.6e}")

# per-pixel: which pixels carry the difference?  The k column of the up-field Jacobian (perspective_fields.py:170-180 through
# camera.py:853-911) of the HIP pixel code against the oracle's float32 / float64 builds at the same parameters.
from geocalib_amd.perspective_fields import J_perspective_field  # noqa: E402
Hh, Ww = data["latitude_field"].shape[-2:]
Jh_up, _ = J_perspective_field(cam, grav, spherical=bool(c1["use_spherical_manifold"]), log_focal=bool(c1["use_log_focal"]), shape=(Hh, Ww)) \
    if "shape" in J_perspective_field.__code__.co_varnames else J_perspective_field(cam, grav, spherical=bool(c1["use_spherical_manifold"]), log_focal=bool(c1["use_log_focal"]))
Jh = Jh_up.cpu().numpy().astype(np.float64)[..., :Ww, :, :] if Jh_up.shape[2] != Ww else Jh_up.cpu().numpy().astype(np.float64)
J32, _ = oracle.jacobian_fields(model, Hh, Ww, cam_np, grav_np, bool(c1["use_spherical_manifold"]), bool(c1["use_log_focal"]), precision="f32")
J64, _ = oracle.jacobian_fields(model, Hh, Ww, cam_np, grav_np, bool(c1["use_spherical_manifold"]), bool(c1["use_log_focal"]), precision="f64")
b = 0
kcol = -1
f, cx, cy, k = cam_np[b, 3], cam_np[b, 4], cam_np[b, 5], cam_np[b, 6]
yy, xx = np.mgrid[0:Hh, 0:Ww]
r2 = ((xx - cx) / cam_np[b, 2]) ** 2 + ((yy - cy) / f) ** 2
mm = 4 * k * r2 / 2.0 ** -24
for name, J in (("hip", Jh), ("o32", J32)):
    d = np.abs(J[b, ..., kcol] - J64[b, ..., kcol]).max(-1)            # (H, W): worst of the two up components
    idx = np.argsort(d.ravel())[::-1][:6]
    print(f"image {b}, |J_up[k] - float64| of {name}: sum of squares {np.square(J[b, ..., kcol] - J64[b, ..., kcol]).sum():.4e}; worst pixels (y, x, r2, 4 k r2 in ulps of 1, error, float64 value):")
    for i in idx:
        y, x = divmod(int(i), Ww)
        print(f"      ({y}, {x})  r2 {r2[y, x]:.3e}  m {mm[y, x]:.2f}  err {d[y, x]:.3e}  J64 {J64[b, y, x, :, kcol]}  {name} {J[b, y, x, :, kcol]}")
