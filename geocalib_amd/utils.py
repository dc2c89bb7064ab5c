"""Angle / focal conversions used by the calibration path (reference: geocalib/utils.py:272-300)."""
import math

import torch


def fov2focal(fov: torch.Tensor, size: torch.Tensor) -> torch.Tensor:
    """Focal length [px] of a (vertical/horizontal) field of view [rad] over `size` pixels."""
    return size / 2 / torch.tan(fov / 2)


def focal2fov(focal: torch.Tensor, size: torch.Tensor) -> torch.Tensor:
    """Field of view [rad] seen by `size` pixels at focal length `focal` [px]."""
    return 2 * torch.arctan(size / (2 * focal))


def rad2deg(rad):
    return rad / math.pi * 180


def deg2rad(deg):
    return deg / 180 * math.pi


def skew_symmetric(v: torch.Tensor) -> torch.Tensor:
    """[v]_x, the (..., 3, 3) cross-product matrix of (..., 3) vectors (reference: geocalib/utils.py:217-229)."""
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack([torch.stack([o, -z, y], -1), torch.stack([z, o, -x], -1), torch.stack([-y, x, o], -1)], -2)


def pitch2rho(pitch: torch.Tensor, f: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """Offset of the horizon from the principal point as a fraction of the image height (geocalib/utils.py:282-284)."""
    return torch.tan(pitch) * f / h


def rho2pitch(rho: torch.Tensor, f: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """Inverse of pitch2rho (geocalib/utils.py:287-289)."""
    return torch.atan(rho * h / f)


def get_device() -> str:
    """"cuda" (= HIP on ROCm) when a device is visible, else "mps" / "cpu" (geocalib/utils.py:302-309).  The LM path
    itself only runs on a HIP device; this helper serves callers (the reference's demos) that place their tensors."""
    if torch.cuda.is_available():
        return "cuda"
    return "mps" if torch.backends.mps.is_available() else "cpu"


def rad2rotmat(roll: torch.Tensor, pitch: torch.Tensor, yaw: torch.Tensor = None) -> torch.Tensor:
    """Rotation Rz(roll) @ Rx(pitch) @ Ry(yaw) with the reference's sign conventions
    (geocalib/utils.py:232-269)."""
    yaw = torch.zeros_like(roll) if yaw is None else yaw
    zero, one = torch.zeros_like(roll), torch.ones_like(roll)
    cr, sr, cp, sp, cy, sy = roll.cos(), roll.sin(), pitch.cos(), pitch.sin(), yaw.cos(), yaw.sin()

    def mat(rows):
        return torch.stack([torch.stack(r, -1) for r in rows], -2)

    Rx = mat([[one, zero, zero], [zero, cp, sp], [zero, -sp, cp]])
    Ry = mat([[cy, zero, -sy], [zero, one, zero], [sy, zero, cy]])
    Rz = mat([[cr, sr, zero], [-sr, cr, zero], [zero, zero, one]])
    return Rz @ Rx @ Ry


def print_calibration(results) -> None:
    """Human-readable summary of a calibrate() result (reference: geocalib/utils.py:309-325)."""
    camera, gravity = results["camera"], results["gravity"]
    roll, pitch = rad2deg(gravity.rp).unbind(-1)
    print("\nEstimated parameters (Pred):")
    print(f"Roll:  {roll.item():.1f}° (± {rad2deg(results['roll_uncertainty']).item():.1f})°")
    print(f"Pitch: {pitch.item():.1f}° (± {rad2deg(results['pitch_uncertainty']).item():.1f})°")
    print(f"vFoV:  {rad2deg(camera.vfov).item():.1f}° (± {rad2deg(results['vfov_uncertainty']).item():.1f})°")
    print(f"Focal: {camera.f[0, 1].item():.1f} px (± {results['focal_uncertainty'].item():.1f} px)")
    if hasattr(camera, "dist"):
        print(f"Dist:    {camera.dist[0, :camera.num_dist_params()].tolist()}")
