"""Seeded synthetic perspective fields for parity tests (TEST INFRASTRUCTURE, CPU / numpy).

Protocol of SURVEY.md section 8(d): per image roll, pitch ~ U(-45,45) deg, vfov ~ U(20,90) deg,
k1 ~ U(-0.3, 0.1) for distortion models; fields = perspective field of the ground-truth camera
(rendered by the oracle's restatement of perspective_fields.py:278) + N(0, 0.02^2) noise, the up
field re-normalised, the latitude clamped to +-(pi/2 - 1e-3); confidences ~ U(0, 1).
Image i of a set is a function of (seed, i) only, so any sub-batch sees identical data.
"""
import numpy as np

try:
    from . import lm_oracle
except ImportError:  # imported with oracle/ on sys.path
    import lm_oracle


def gt_params(seed: int, index: int, camera_model: str, H: int, W: int):
    rng = np.random.default_rng([seed, index, 0])
    roll, pitch = np.deg2rad(rng.uniform(-45, 45, 2))
    vfov = np.deg2rad(rng.uniform(20, 90))
    f = H / 2 / np.tan(vfov / 2)
    k1 = rng.uniform(-0.3, 0.1) if camera_model != "pinhole" else 0.0
    k2 = rng.uniform(-0.02, 0.02) if camera_model == "radial" else 0.0
    cam = np.array([W, H, f, f, W / 2, H / 2, k1, k2], np.float64)
    sr, cr, sp, cp = np.sin(roll), np.cos(roll), np.sin(pitch), np.cos(pitch)
    grav = np.array([-sr * cp, -cr * cp, sp], np.float64)
    return cam, grav, (roll, pitch, vfov)


def make_fields(seed: int, indices, camera_model: str, H: int, W: int, noise: float = 0.02,
                confidences: bool = True):
    """Return (data dict of float32 arrays, gt cameras (B,8), gt gravities (B,3))."""
    indices = list(indices)
    cams = np.stack([gt_params(seed, i, camera_model, H, W)[0] for i in indices])
    gravs = np.stack([gt_params(seed, i, camera_model, H, W)[1] for i in indices])
    up, lat = lm_oracle.render(camera_model, H, W, cams, gravs, precision="f64")
    B = len(indices)
    upc = np.empty((B, H, W), np.float32)
    latc = np.empty((B, H, W), np.float32)
    for j, i in enumerate(indices):
        rng = np.random.default_rng([seed, i, 1])
        up[j] += rng.normal(0, noise, up[j].shape).astype(np.float32)
        lat[j] += rng.normal(0, noise, lat[j].shape).astype(np.float32)
        upc[j] = rng.uniform(0, 1, (H, W)).astype(np.float32)
        latc[j] = rng.uniform(0, 1, (H, W)).astype(np.float32)
    up /= np.sqrt((up.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    lim = np.float32(np.pi / 2 - 1e-3)
    lat = np.clip(lat, -lim, lim)
    data = {"up_field": up, "latitude_field": lat}
    if confidences:
        data |= {"up_confidence": upc, "latitude_confidence": latc}
    return data, cams.astype(np.float32), gravs.astype(np.float32)


def make_shared_group(seed: int, group: int, camera_model: str, H: int, W: int, frames: int = 16,
                      noise: float = 0.02):
    """Fields of ONE shared-intrinsics group (BASELINE configs[4]: 16 frames of one camera): the intrinsics are
    those of image index 10_000 + group, the gravity of frame i is that of image index (10_000 + group) * 64 + i;
    noise / confidences are keyed by the frame's own index, so every frame is a function of (seed, group, i)."""
    base = 10_000 + group
    cam0 = gt_params(seed, base, camera_model, H, W)[0]
    idx = [base * 64 + i for i in range(frames)]
    cams = np.stack([cam0] * frames)
    gravs = np.stack([gt_params(seed, j, camera_model, H, W)[1] for j in idx])
    up, lat = lm_oracle.render(camera_model, H, W, cams, gravs, precision="f64")
    upc = np.empty((frames, H, W), np.float32)
    latc = np.empty((frames, H, W), np.float32)
    for k, j in enumerate(idx):
        rng = np.random.default_rng([seed, j, 1])
        up[k] += rng.normal(0, noise, up[k].shape).astype(np.float32)
        lat[k] += rng.normal(0, noise, lat[k].shape).astype(np.float32)
        upc[k] = rng.uniform(0, 1, (H, W)).astype(np.float32)
        latc[k] = rng.uniform(0, 1, (H, W)).astype(np.float32)
    up /= np.sqrt((up.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    lim = np.float32(np.pi / 2 - 1e-3)
    lat = np.clip(lat, -lim, lim)
    data = {"up_field": up, "latitude_field": lat, "up_confidence": upc, "latitude_confidence": latc}
    return data, cams.astype(np.float32), gravs.astype(np.float32)
