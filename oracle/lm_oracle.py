"""ctypes front-end of the CPU oracle (oracle/lm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package geocalib_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
CAMERA_MODELS = {"pinhole": 0, "simple_radial": 1, "radial": 2, "simple_divisional": 3}
MAXP = 5

DEFAULT_CONF = {  # lm_optimizer.py:144-162
    "camera_model": "pinhole", "shared_intrinsics": False, "num_steps": 30, "lambda_": 0.1,
    "fix_lambda": False, "early_stop": True, "atol": 1e-8, "rtol": 1e-8,
    "use_spherical_manifold": True, "use_log_focal": True,
    "up_loss_fn_scale": 1e-2, "lat_loss_fn_scale": 1e-2, "verbose": False,
    # siclib knobs (siclib/models/optimization/lm_optimizer.py:37-59)
    "loss_fn": "huber_loss", "init_conf": {"name": "trivial"}, "group_size": None,
}


class _Conf(C.Structure):
    _fields_ = [("camera_model", C.c_int), ("shared_intrinsics", C.c_int), ("num_steps", C.c_int),
                ("lambda0", C.c_double), ("fix_lambda", C.c_int), ("early_stop", C.c_int),
                ("atol", C.c_double), ("rtol", C.c_double), ("use_spherical_manifold", C.c_int),
                ("use_log_focal", C.c_int), ("up_loss_fn_scale", C.c_double),
                ("lat_loss_fn_scale", C.c_double), ("training", C.c_int), ("num_threads", C.c_int),
                ("heuristic_init", C.c_int)]


_FP = C.POINTER(C.c_float)


class _Data(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("up", _FP), ("lat", _FP),
                ("up_conf", _FP), ("lat_conf", _FP), ("scales", _FP), ("prior_focal", _FP),
                ("prior_gravity", _FP), ("prior_dist", _FP)]


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (idempotent)."""
    outs = [os.path.join(_BUILD, f"liblm_oracle_{p}.so") for p in ("f32", "f64")]
    src = os.path.join(_HERE, "lm_oracle.c")
    if not force and all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs):
        return
    subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)


_libs = {}


def _lib(precision: str):
    if precision not in _libs:
        path = os.path.join(_BUILD, f"liblm_oracle_{precision}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.lm_oracle_solve.restype = C.c_int
        lib.lm_oracle_system.restype = C.c_int
        lib.lm_oracle_jacobians.restype = C.c_int
        _libs[precision] = lib
    return _libs[precision]


def effective_cpus() -> int:
    """CPUs this process may actually use: min(os.cpu_count(), scheduler affinity, cgroup CPU quota).  A GPU box shows
    256 logical CPUs but grants a container 16 of them (cpu.max); 256 OpenMP threads inside that quota spend their
    time being throttled (a 3 s test took 220 s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                quota, period = int(fq.read()), int(fp.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    return C.cast(None, _FP) if a is None else a.ctypes.data_as(_FP)


def _pack(conf, data, training=False, num_threads=0):
    cf = {**DEFAULT_CONF, **(conf or {})}
    c = _Conf(CAMERA_MODELS[cf["camera_model"]], int(cf["shared_intrinsics"]), int(cf["num_steps"]),
              float(cf["lambda_"]), int(cf["fix_lambda"]), int(cf["early_stop"]), float(cf["atol"]),
              float(cf["rtol"]), int(cf["use_spherical_manifold"]), int(cf["use_log_focal"]),
              float(2 ** 20 if cf["loss_fn"] == "squared_loss" else cf["up_loss_fn_scale"]),
              float(2 ** 20 if cf["loss_fn"] == "squared_loss" else cf["lat_loss_fn_scale"]), int(training),
              int(num_threads) if num_threads > 0 else effective_cpus(), int(cf["init_conf"]["name"] == "heuristic"))
    keep = {k: _f32(data.get(k)) for k in ("up_field", "latitude_field", "up_confidence",
                                           "latitude_confidence", "scales", "prior_focal",
                                           "prior_gravity", "prior_dist")}
    ref = keep["up_field"] if keep["up_field"] is not None else keep["latitude_field"]
    B, _, H, W = ref.shape
    d = _Data(B, H, W, _ptr(keep["up_field"]), _ptr(keep["latitude_field"]),
              _ptr(keep["up_confidence"]), _ptr(keep["latitude_confidence"]), _ptr(keep["scales"]),
              _ptr(keep["prior_focal"]), _ptr(keep["prior_gravity"]), _ptr(keep["prior_dist"]))
    return cf, c, d, keep, (B, H, W)


def solve(data: dict, conf: dict = None, precision: str = "f32", training: bool = False,
          num_threads: int = 0, trace: bool = False) -> dict:
    """Run the restated LMOptimizer.forward (lm_optimizer.py:646-664) on numpy inputs.

    data keys follow the reference: up_field (B,2,H,W), latitude_field (B,1,H,W), up_confidence,
    latitude_confidence (B,H,W), scales (2,), prior_focal (B,), prior_gravity (B,3), prior_dist (B,nd).
    Returns numpy arrays keyed like the reference's output dict (camera -> (B,8), gravity -> (B,3)).
    """
    lib = _lib(precision)
    cf, c, d, keep, (B, H, W) = _pack(conf, data, training, num_threads)
    stride = lib.lm_oracle_info_stride()
    cam = np.zeros((B, 8), np.float32)
    grav = np.zeros((B, 3), np.float32)
    info = np.zeros((B, stride), np.float32)
    tr = None
    if trace:
        tr = np.zeros((cf["num_steps"], B, lib.lm_oracle_trace_stride()), np.float64)
    rc = lib.lm_oracle_solve(C.byref(c), C.byref(d), _ptr(cam), _ptr(grav), _ptr(info),
                             tr.ctypes.data_as(C.POINTER(C.c_double)) if trace else None)
    if rc != 0:
        raise RuntimeError(f"lm_oracle_solve failed: {rc}")
    P = int(info[0, 12])
    out = {"camera": cam, "gravity": grav, "stop_at": info[:, 0].copy(),
           "initial_up_cost": info[:, 1].copy(), "initial_latitude_cost": info[:, 2].copy(),
           "initial_cost": info[:, 3].copy(), "final_up_cost": info[:, 4].copy(),
           "final_latitude_cost": info[:, 5].copy(), "final_cost": info[:, 6].copy(),
           "lambda": info[:, 13].copy()}
    if not training:
        out.update({"covariance": info[:, 16:16 + P * P].reshape(B, P, P).copy(),
                    "roll_uncertainty": info[:, 7].copy(), "pitch_uncertainty": info[:, 8].copy(),
                    "gravity_uncertainty": info[:, 9].copy(), "focal_uncertainty": info[:, 10].copy(),
                    "vfov_uncertainty": info[:, 11].copy()})
    if trace:
        out["trace"] = {"cost_up": tr[..., 0], "cost_lat": tr[..., 1], "lambda": tr[..., 2],
                        "G": tr[..., 3:8], "H": tr[..., 8:33].reshape(tr.shape[0], B, 5, 5),
                        "delta": tr[..., 33:38], "cam": tr[..., 38:42], "gravity": tr[..., 42:45]}
    return out


def system(data: dict, camera: np.ndarray, gravity: np.ndarray, conf: dict = None,
           as_rpf: bool = False, precision: str = "f32") -> dict:
    """One sweep at fixed parameters: mean Huber costs, J^T W r and J^T W J per image."""
    lib = _lib(precision)
    cf, c, d, keep, (B, H, W) = _pack(conf, data)
    cam, grav = _f32(camera).reshape(B, 8), _f32(gravity).reshape(B, 3)
    cu, cl = np.zeros(B), np.zeros(B)
    G, Hm = np.zeros((B, MAXP)), np.zeros((B, MAXP, MAXP))
    dp = C.POINTER(C.c_double)
    P = lib.lm_oracle_system(C.byref(c), C.byref(d), _ptr(cam), _ptr(grav), int(as_rpf),
                             cu.ctypes.data_as(dp), cl.ctypes.data_as(dp), G.ctypes.data_as(dp),
                             Hm.ctypes.data_as(dp))
    return {"cost_up": cu, "cost_lat": cl, "G": G[:, :P].copy(), "H": Hm[:, :P, :P].copy()}


def jacobian_fields(camera_model: str, H: int, W: int, camera: np.ndarray, gravity: np.ndarray,
                    spherical: bool, log_focal: bool, precision: str = "f64"):
    """Per-pixel Jacobians of the predicted fields (perspective_fields.py:323-365): J_up (B,H,W,2,P),
    J_lat (B,H,W,1,P) with P = 3 + number of distortion parameters."""
    lib = _lib(precision)
    cam, grav = _f32(camera).reshape(-1, 8), _f32(gravity).reshape(-1, 3)
    B = cam.shape[0]
    P = 3 + {"pinhole": 0, "simple_radial": 1, "radial": 2, "simple_divisional": 1}[camera_model]
    J_up = np.zeros((B, H, W, 2, P), np.float64)
    J_lat = np.zeros((B, H, W, 1, P), np.float64)
    dp = C.POINTER(C.c_double)
    for b in range(B):
        got = lib.lm_oracle_jacobians(CAMERA_MODELS[camera_model], H, W, _ptr(cam[b]), _ptr(grav[b]), int(spherical),
                                      int(log_focal), J_up[b].ctypes.data_as(dp), J_lat[b].ctypes.data_as(dp))
        assert got == P
    return J_up, J_lat


def residual_fields(camera_model: str, data: dict, camera: np.ndarray, gravity: np.ndarray, precision: str = "f64"):
    """calculate_residuals (lm_optimizer.py:248-274): {"up_residual": (B,N,2), "latitude_residual": (B,N,1)}."""
    lib = _lib(precision)
    cam, grav = _f32(camera).reshape(-1, 8), _f32(gravity).reshape(-1, 3)
    up, lat = _f32(data.get("up_field")), _f32(data.get("latitude_field"))
    ref = lat if lat is not None else up
    B, _, H, W = ref.shape
    dp = C.POINTER(C.c_double)
    r_up = np.zeros((B, H * W, 2), np.float64) if up is not None else None
    r_lat = np.zeros((B, H * W, 1), np.float64) if lat is not None else None
    for b in range(B):
        lib.lm_oracle_residuals(CAMERA_MODELS[camera_model], H, W, _ptr(cam[b]), _ptr(grav[b]),
                                _ptr(None if up is None else up[b]), _ptr(None if lat is None else lat[b]),
                                r_up[b].ctypes.data_as(dp) if up is not None else None,
                                r_lat[b].ctypes.data_as(dp) if lat is not None else None)
    out = {}
    if r_up is not None:
        out["up_residual"] = r_up
    if r_lat is not None:
        out["latitude_residual"] = r_lat
    return out


def huber_costs(residual: np.ndarray, scale: float, conf=None, precision: str = "f64"):
    """calculate_costs (lm_optimizer.py:276-315) of residual rows (..., dim): (cost, weight) of shape (...)."""
    lib = _lib(precision)
    x2 = np.ascontiguousarray((np.asarray(residual, np.float64) ** 2).sum(-1))
    cf = None if conf is None else _f32(conf).reshape(x2.shape)
    cost, weight = np.zeros_like(x2), np.zeros_like(x2)
    dp = C.POINTER(C.c_double)
    lib.lm_oracle_huber_costs(x2.ctypes.data_as(dp), x2.size, C.c_double(scale), _ptr(cf), cost.ctypes.data_as(dp),
                              weight.ctypes.data_as(dp))
    return cost, weight


def render(camera_model: str, H: int, W: int, camera: np.ndarray, gravity: np.ndarray,
           precision: str = "f64"):
    """Perspective field of each (camera, gravity): up (B,2,H,W), lat (B,1,H,W) (perspective_fields.py:278)."""
    lib = _lib(precision)
    cam, grav = _f32(camera).reshape(-1, 8), _f32(gravity).reshape(-1, 3)
    B = cam.shape[0]
    up = np.zeros((B, 2, H, W), np.float32)
    lat = np.zeros((B, 1, H, W), np.float32)
    for b in range(B):
        lib.lm_oracle_render(CAMERA_MODELS[camera_model], H, W, _ptr(cam[b]), _ptr(grav[b]),
                             _ptr(up[b]), _ptr(lat[b]))
    return up, lat
