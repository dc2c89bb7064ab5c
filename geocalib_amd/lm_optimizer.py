"""Levenberg-Marquardt calibration on MI355X: host side of the HIP path.

Drop-in for the reference's `geocalib.lm_optimizer.LMOptimizer` (lm_optimizer.py:141-664): same
constructor conf keys (`default_conf`), `set_camera_model`, `.shared_intrinsics`, `.num_steps`,
`.training`, dict-in / dict-out `forward`.  The whole optimisation (residuals, Huber weights,
analytic Jacobians, J^T W J / J^T W r, damped Cholesky, manifold update, lambda rule, early stop,
uncertainty) runs in hand-written HIP kernels behind the C ABI of include/gclm.h; this module
only prepares the initial estimate, the configuration and the output dict.  torch is used for
device memory and the stream, nothing else.  There is NO CPU fallback: CPU tensors or a missing
libgeocalib_hip.so raise.
"""
import logging
from types import SimpleNamespace
from typing import Any, Dict, Tuple

import torch
import torch.nn as nn

from . import _lib
from .camera import BaseCamera, camera_models
from .gravity import Gravity
from .utils import focal2fov

logger = logging.getLogger(__name__)

_HIP_MODELS = ("pinhole", "simple_radial", "radial", "simple_divisional")


def get_trivial_estimation(data: Dict[str, torch.Tensor], camera_model) -> Tuple[BaseCamera, Gravity]:
    """Initial estimate: roll = pitch = 0, f = 0.7 max(h, w) through a vfov round trip, principal
    point at the centre, priors substituted where given (reference: lm_optimizer.py:20-58).
    Like the reference this needs `latitude_field` to be present."""
    ref = data.get("up_field", data["latitude_field"]).detach()
    B, (h, w) = ref.shape[0], ref.shape[-2:]
    hs, ws = ref.new_ones((B,)) * h, ref.new_ones((B,)) * w
    focal = data.get("prior_focal", 0.7 * torch.max(hs, ws))
    params = {"width": ws, "height": hs, "vfov": focal2fov(focal, h)}
    if "scales" in data:
        params["scales"] = data["scales"]
    if "prior_dist" in data:
        params["dist"] = data["prior_dist"]
    camera = camera_model.from_dict(params).float().to(ref.device)
    gravity = Gravity.from_rp(ref.new_zeros((B,)), ref.new_zeros((B,))).float().to(ref.device)
    if "prior_gravity" in data:
        g = data["prior_gravity"].float().to(ref.device)
        gravity = Gravity(g) if isinstance(g, torch.Tensor) else g
    return camera, gravity


class _Handle:
    """Owns one gclm_handle: one per (device, stream) -- the handle owns the whole solve workspace (states, parameter
    blocks, partial records, early-stop counters), so two streams must never share one (include/gclm.h)."""

    def __init__(self, cfg: _lib.GclmConfig, device: torch.device):
        lib = _lib.load()
        self.ptr = _lib.C.c_void_p()
        assert cfg.device == (device.index or 0)
        rc = lib.gclm_create(_lib.C.byref(self.ptr), _lib.C.byref(cfg))
        _lib.check(rc, None, "gclm_create")
        self.key = cfg.key()
        self.device = cfg.device
        self.paced = 0
        self.row_pairs = -1
        self.signature = None           # LMOptimizer._config_signature the handle was last configured for

    def set_paced(self, depth: int):
        if depth != self.paced:
            _lib.check(_lib.load().gclm_set_paced_launches(self.ptr, int(depth)), self.ptr, "gclm_set_paced_launches")
            self.paced = depth

    def set_row_pairs(self, mode: int):
        if mode != self.row_pairs:
            _lib.check(_lib.load().gclm_set_row_pairs(self.ptr, int(mode)), self.ptr, "gclm_set_row_pairs")
            self.row_pairs = mode

    def destroy(self):
        if self.ptr:
            _lib.load().gclm_destroy(self.ptr)
            self.ptr = _lib.C.c_void_p()

    def configure(self, cfg: _lib.GclmConfig):
        if cfg.key() != self.key:
            _lib.check(_lib.load().gclm_configure(self.ptr, _lib.C.byref(cfg)), self.ptr, "gclm_configure")
            self.key = cfg.key()

    def __del__(self):
        try:
            if self.ptr:
                _lib.load().gclm_destroy(self.ptr)
        except Exception:
            pass


def _unit_gravity(data: torch.Tensor) -> Gravity:
    """Wrap device results that are unit vectors already (skips the constructor's re-normalisation kernels)."""
    g = Gravity.__new__(Gravity)
    g._data = data
    return g


def _raw_stream(device: torch.device) -> int:
    """The current HIP stream of `device` as the integer the C ABI takes (torch.cuda.current_stream(...).cuda_stream builds a
    Stream object first: 4.5 us per call on the single-image path)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    try:
        return torch._C._cuda_getCurrentRawStream(idx)
    except AttributeError:            # a torch without the private accessor
        return torch.cuda.current_stream(device).cuda_stream


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"geocalib_amd: `{name}` must live on a HIP device (got {t.device}); "
                           "the MI355X path has no CPU fallback")
    if t.dtype is torch.float32 and t.is_contiguous() and not t.requires_grad:
        return t                                   # the common case: nothing to convert (saves three dispatcher trips)
    return t.detach().to(torch.float32).contiguous()


def huber_loss(x: torch.Tensor):
    """Huber loss of an already squared residual `x`: (loss, first derivative = IRLS weight, second derivative)
    (reference: lm_optimizer.py:79-87), evaluated by gclm_huber_costs in the sweep's branch-free form."""
    if not x.is_cuda:
        raise RuntimeError("geocalib_amd.huber_loss needs a HIP device tensor (no CPU fallback)")
    xs = x.detach().to(torch.float32).contiguous()
    loss, d1, d2 = torch.empty_like(xs), torch.empty_like(xs), torch.empty_like(xs)
    with torch.cuda.device(xs.device):
        rc = _lib.load().gclm_huber_costs(xs.data_ptr(), xs.numel(), 0, 1.0, None, loss.data_ptr(), d1.data_ptr(),
                                          d2.data_ptr(), torch.cuda.current_stream(xs.device).cuda_stream)
    if rc != 0:
        raise _lib.GclmError(f"gclm_huber_costs failed ({rc})")
    return loss, d1, d2


def scaled_loss(x: torch.Tensor, fn, a: float):
    """`fn` applied to x / a^2, value and derivatives scaled back (reference: lm_optimizer.py:61-76)."""
    a2 = a**2
    loss, d1, d2 = fn(x / a2)
    return loss * a2, d1, d2 / a2


def early_stop(new_cost: torch.Tensor, prev_cost: torch.Tensor, atol: float, rtol: float) -> bool:
    """Batch-global convergence test of the reference (lm_optimizer.py:90-92).  Inside a solve this decision is
    taken on the device from per-step counters; the host form is kept for callers that drive their own loop."""
    return bool(torch.allclose(new_cost, prev_cost, atol=atol, rtol=rtol))


def update_lambda(lamb: torch.Tensor, prev_cost: torch.Tensor, new_cost: torch.Tensor, lambda_min: float = 1e-6,
                  lambda_max: float = 1e2) -> torch.Tensor:
    """Damping rule (lm_optimizer.py:95-106): x10 where the cost went up, x0.1 elsewhere, clamped."""
    factor = torch.where(new_cost > prev_cost, torch.full_like(lamb, 10.0), torch.full_like(lamb, 0.1))
    return (lamb * factor).clamp(lambda_min, lambda_max)


def optimizer_step(G: torch.Tensor, H: torch.Tensor, lambda_: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """One damped Gauss-Newton step per system, delta = (H + diag(clamp(lambda diag H, eps)))^-1 G, solved on the
    device (gclm_optimizer_step; the reference copies H and G to the CPU, lm_optimizer.py:109-137).  G (..., N),
    H (..., N, N), lambda_ a scalar tensor or one value per system; N <= 5.  A system that is not positive definite
    gets a zero step (the reference zeroes the whole batch)."""
    if not (G.is_cuda and H.is_cuda):
        raise RuntimeError("geocalib_amd.optimizer_step needs HIP device tensors (no CPU fallback)")
    N = G.shape[-1]
    assert H.shape[-2:] == (N, N) and H.shape[:-2] == G.shape[:-1], (G.shape, H.shape)
    g = G.detach().to(torch.float32).reshape(-1, N).contiguous()
    h = H.detach().to(torch.float32).reshape(-1, N, N).contiguous()
    B = g.shape[0]
    lam = torch.as_tensor(lambda_, dtype=torch.float32, device=g.device).reshape(-1).contiguous()
    assert lam.numel() in (1, B), f"lambda_ must hold 1 or {B} values"
    delta = torch.empty_like(g)
    with torch.cuda.device(g.device):
        rc = _lib.load().gclm_optimizer_step(g.data_ptr(), h.data_ptr(), lam.data_ptr(), int(lam.numel() == 1), float(eps),
                                             B, N, delta.data_ptr(), None, torch.cuda.current_stream(g.device).cuda_stream)
    if rc != 0:
        raise _lib.GclmError(f"gclm_optimizer_step failed ({rc})")
    return delta.reshape(G.shape)


class LMOptimizer(nn.Module):
    """Batched LM optimiser for camera calibration (HIP / gfx950)."""

    default_conf = {
        "camera_model": "pinhole",
        "shared_intrinsics": False,
        "num_steps": 30,
        "lambda_": 0.1,
        "fix_lambda": False,
        "early_stop": True,
        "atol": 1e-8,
        "rtol": 1e-8,
        "use_spherical_manifold": True,
        "use_log_focal": True,
        "up_loss_fn_scale": 1e-2,
        "lat_loss_fn_scale": 1e-2,
        "verbose": False,
        # extension: frames per shared-intrinsics group (None = the whole batch is one group,
        # which is the reference's only mode; lm_optimizer.py:350-383)
        "group_size": None,
        # knobs of the training-time optimiser (siclib/models/optimization/lm_optimizer.py:37-59)
        "loss_fn": "huber_loss",          # {"huber_loss", "squared_loss"}
        "init_conf": {"name": "trivial"},  # {"trivial", "heuristic"} (siclib/models/optimization/utils.py:16-82)
    }

    # squared_loss (siclib/models/optimization/losses.py:26) is the Huber loss with an unreachable threshold:
    # with a = 2^20 every residual is an inlier, weight = 1 and cost = (x / a^2) a^2 = x exactly (power of two).
    _SQUARED_LOSS_SCALE = float(2 ** 20)

    def __init__(self, conf: Dict[str, Any] = None):
        super().__init__()
        conf = conf or {}
        unknown = set(conf) - set(self.default_conf)
        if unknown:
            logger.debug("ignoring unknown conf keys: %s", sorted(unknown))
        self.conf = conf = SimpleNamespace(**{**self.default_conf, **conf})
        self.num_steps = conf.num_steps
        self.set_camera_model(conf.camera_model)
        self.setup_optimization_and_priors(shared_intrinsics=conf.shared_intrinsics)
        self._handles: Dict[Tuple[int, int], _Handle] = {}
        self._warned_training = False
        self._warned_chunked_stop = False

    # ------------------------------------------------------------------ reference API
    def set_camera_model(self, camera_model: str) -> None:
        assert camera_model in camera_models, f"Unknown camera model: {camera_model} not in {camera_models.keys()}"
        self.camera_model = camera_models[camera_model]
        self.camera_has_distortion = hasattr(self.camera_model, "dist")

    def setup_optimization_and_priors(self, data: Dict[str, torch.Tensor] = None,
                                      shared_intrinsics: bool = False) -> None:
        """Which parameters are free given the priors in `data` (reference: lm_optimizer.py:189-246)."""
        data = data or {}
        estimate_gravity = "prior_gravity" not in data
        estimate_focal = "prior_focal" not in data
        estimate_dist = self.camera_has_distortion and "prior_dist" not in data
        gravity_delta_dims = (0, 1) if estimate_gravity else (-1,)
        focal_delta_dims = (max(gravity_delta_dims) + 1,) if estimate_focal else (-1,)
        dist_delta_dims = None
        nd = self.camera_model.num_dist_params() if self.camera_has_distortion else 0
        if estimate_dist:
            first = focal_delta_dims[-1] + 1
            dist_delta_dims = tuple(range(first, first + nd))
        # plain Python values: written straight into the instance dict (nn.Module.__setattr__ costs ~2 us apiece, and this
        # runs on every forward -- a quarter of the host time of a single-image solve)
        self.__dict__.update(shared_intrinsics=shared_intrinsics, estimate_gravity=estimate_gravity,
                             estimate_focal=estimate_focal, estimate_dist=estimate_dist,
                             gravity_delta_dims=gravity_delta_dims, focal_delta_dims=focal_delta_dims,
                             dist_delta_dims=dist_delta_dims, n_intrinsic_params=estimate_focal + nd)

    # ------------------------------------------------------------------ C-ABI plumbing
    def _config_signature(self, device_index: int):
        """Everything _config() reads, as a tuple that is cheap to build and compare: a handle is only reconfigured (and the
        ctypes struct only rebuilt) when it changes."""
        c = self.conf
        return (device_index, self.camera_model, self.shared_intrinsics, self.num_steps, self.estimate_gravity,
                self.estimate_focal, self.estimate_dist, self.training, c.group_size, c.lambda_, c.fix_lambda, c.early_stop,
                c.atol, c.rtol, c.use_spherical_manifold, c.use_log_focal, c.loss_fn, c.up_loss_fn_scale,
                c.lat_loss_fn_scale, c.init_conf["name"] if isinstance(c.init_conf, dict) else getattr(c.init_conf, "name", "trivial"))

    def _config(self, device_index: int = 0) -> _lib.GclmConfig:
        c = self.conf
        name = self.camera_model.name()
        if name not in _HIP_MODELS:
            raise NotImplementedError(f"camera model `{name}` is not implemented by the HIP path yet "
                                      f"(available: {_HIP_MODELS})")
        cfg = _lib.GclmConfig.default(device_index)
        cfg.camera_model = _lib.CAMERA_MODEL_IDS[name]
        cfg.shared_intrinsics = int(bool(self.shared_intrinsics))
        cfg.group_size = int(c.group_size or 0)
        cfg.num_steps = int(self.num_steps)
        cfg.lambda0 = float(c.lambda_)
        cfg.fix_lambda = int(bool(c.fix_lambda))
        cfg.early_stop = int(bool(c.early_stop))
        cfg.atol, cfg.rtol = float(c.atol), float(c.rtol)
        cfg.use_spherical_manifold = int(bool(c.use_spherical_manifold))
        cfg.use_log_focal = int(bool(c.use_log_focal))
        assert c.loss_fn in ("huber_loss", "squared_loss"), f"Unknown loss_fn: {c.loss_fn}"
        squared = c.loss_fn == "squared_loss"
        cfg.up_loss_fn_scale = self._SQUARED_LOSS_SCALE if squared else float(c.up_loss_fn_scale)
        cfg.lat_loss_fn_scale = self._SQUARED_LOSS_SCALE if squared else float(c.lat_loss_fn_scale)
        init_name = c.init_conf["name"] if isinstance(c.init_conf, dict) else getattr(c.init_conf, "name", "trivial")
        assert init_name in ("trivial", "heuristic"), f"Unknown initialisation: {init_name}"
        cfg.heuristic_init = int(init_name == "heuristic")
        cfg.estimate_gravity = int(self.estimate_gravity)
        cfg.estimate_focal = int(self.estimate_focal)
        cfg.estimate_dist = int(self.estimate_dist)
        cfg.compute_uncertainty = int(not self.training)
        return cfg

    _MAX_HANDLES = 16      # workspaces kept alive per optimiser (each is O(B) small records; LRU beyond this)

    # Host-side knob without a reference counterpart (include/gclm.h: gclm_set_paced_launches).  d > 0: a single-image
    # solve with early stop issues launch k only after launch k - d has reported, and none after the stop -- the call then
    # blocks for about the LM loop's duration instead of returning at once, and finishes ~20 % earlier (no queue turns for
    # the launches after the stop).  For callers that read the result right away; GeoCalib sets 3.  Results are unaffected.
    paced_launches = 0

    # Host-side knob without a reference counterpart (include/gclm.h: gclm_set_row_pairs).  The sweep of a radial model may
    # take row H - y along with row y and evaluate what depends on r^2 once for both (same per-pixel values, another order of
    # additions: results equal the one-row walk's to float32 summation order).  None: the library decides (batches of radial /
    # simple_divisional images); False: never -- the one-row walk, bit for bit; True: wherever the sweep can (small launches too).
    row_pairs = None

    def _handle(self, device: torch.device, stream: int = None) -> _Handle:
        """The gclm_handle of (device, stream): solves issued from different torch streams (e.g. the CNN of batch k+1
        overlapping the LM of batch k) get different workspaces; the same stream reuses its own, in order."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if stream is None:
            stream = torch.cuda.current_stream(torch.device("cuda", idx)).cuda_stream
        key = (idx, int(stream))
        sig = self._config_signature(idx)
        h = self._handles.pop(key, None)
        if h is not None and h.signature == sig:         # the common case of a serving loop: nothing to rebuild
            self._handles[key] = h
            h.set_paced(int(self.paced_launches))
            h.set_row_pairs(-1 if self.row_pairs is None else int(bool(self.row_pairs)))
            return h
        cfg = self._config(idx)
        if h is None:
            while len(self._handles) >= self._MAX_HANDLES:      # drop the least recently used (its stream may be gone)
                old_key = next(iter(self._handles))
                old = self._handles.pop(old_key)
                torch.cuda.synchronize(torch.device("cuda", old_key[0]))   # ITS device: the workspace may still be in flight
                old.destroy()                                             # explicit: not whenever refcounting allows
            h = _Handle(cfg, torch.device("cuda", idx))
        else:
            h.configure(cfg)
        self._handles[key] = h            # most recently used last
        h.signature = sig
        h.set_paced(int(self.paced_launches))
        h.set_row_pairs(-1 if self.row_pairs is None else int(bool(self.row_pairs)))
        return h

    @staticmethod
    def _ptr(t):
        return None if t is None else t.data_ptr()

    def _fields(self, data):
        lat = _dev_f32(data["latitude_field"], "latitude_field")
        up = _dev_f32(data["up_field"], "up_field") if "up_field" in data else None
        upc = _dev_f32(data["up_confidence"], "up_confidence") if "up_confidence" in data and up is not None else None
        latc = _dev_f32(data["latitude_confidence"], "latitude_confidence") if "latitude_confidence" in data else None
        B, _, H, W = lat.shape
        if up is not None:
            assert up.shape == (B, 2, H, W), up.shape
        for c in (upc, latc):
            assert c is None or c.numel() == B * H * W, c.shape
        return up, lat, upc, latc, (B, H, W)

    def optimize(self, data: Dict[str, torch.Tensor], camera_opt: BaseCamera,
                 gravity_opt: Gravity) -> Tuple[BaseCamera, Gravity, Dict[str, torch.Tensor]]:
        """All LM steps + final costs + uncertainty on the device (reference: lm_optimizer.py:551-644)."""
        up, lat, upc, latc, (B, H, W) = self._fields(data)
        device = lat.device
        h = self._handle(device)
        cam = _dev_f32(camera_opt._data, "camera").clone()
        grav = _dev_f32(gravity_opt._data, "gravity").clone()
        info = torch.empty((B, _lib.INFO_STRIDE), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = _lib.load().gclm_solve(h.ptr, self._ptr(up), self._ptr(lat), self._ptr(upc), self._ptr(latc),
                                        B, H, W, cam.data_ptr(), grav.data_ptr(), info.data_ptr(), stream)
        _lib.check(rc, h.ptr, "gclm_solve")
        self._last_raw = (cam, grav, info)     # packed device results (parallel.calibrate_sharded)
        return camera_opt.__class__(cam), _unit_gravity(grav), self._unpack_info(info, up is not None)

    def _unpack_info(self, info: torch.Tensor, has_up: bool) -> Dict[str, torch.Tensor]:
        I = _lib.INFO
        # column k of the packed rows, for the 15 scalar columns in ONE call (15 separate integer selects were 12 us of a
        # 38 us host path; scripts/probes/host_overhead_probe.py)
        col = info.t()[:15].unbind(0)
        out = {"stop_at": col[I["stop_at"]]}
        if has_up:
            out["initial_up_cost"] = col[I["initial_up_cost"]]
        out["initial_latitude_cost"] = col[I["initial_latitude_cost"]]
        out["initial_cost"] = col[I["initial_cost"]]
        if not self.training:
            P = (2 * self.estimate_gravity + self.estimate_focal
                 + (self.camera_model.num_dist_params() if self.camera_has_distortion else 0))
            c0 = I["covariance"]
            out["covariance"] = info[:, c0:c0 + P * P].reshape(-1, P, P)
            for k in ("roll_uncertainty", "pitch_uncertainty", "gravity_uncertainty", "focal_uncertainty",
                      "vfov_uncertainty"):
                out[k] = col[I[k]]
        if has_up:
            out["final_up_cost"] = col[I["final_up_cost"]]
        out["final_latitude_cost"] = col[I["final_latitude_cost"]]
        out["final_cost"] = col[I["final_cost"]]
        out["step_failures"] = col[I["step_failures"]]
        return out

    def forward(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Run the LM optimisation (reference: lm_optimizer.py:646-664)."""
        if self.training and not self._warned_training and any(
                torch.is_tensor(v) and v.requires_grad for v in data.values()):
            # the training-time optimiser of siclib backpropagates through the solve; this path is an opaque C call
            self._warned_training = True
            logger.warning("geocalib_amd.LMOptimizer is inference-only: its outputs carry no autograd graph, so no "
                           "gradient reaches the tensors of `data` that require one (INTEGRATION.md)")
        with torch.no_grad():
            data["latitude_field"]            # KeyError like get_trivial_estimation (lm_optimizer.py:31)
            self.setup_optimization_and_priors(data, shared_intrinsics=self.shared_intrinsics)
            if self.conf.verbose:
                return self._forward_verbose(data)
            camera_opt, gravity_opt, infos = self.calibrate_fields(data)
        return {"camera": camera_opt, "gravity": gravity_opt, **infos}

    def _forward_verbose(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """conf.verbose (lm_optimizer.py:652-662): the reference's five log lines.  Its timing is a host wall clock around
        synchronous tensor ops; the solve here is asynchronous, so the verbose path (only) synchronises the device around it."""
        import time
        from .utils import rad2deg
        device = data["latitude_field"].device
        camera_init, gravity_init = get_trivial_estimation(data, self.camera_model)
        torch.cuda.synchronize(device)
        start = time.time()
        camera_opt, gravity_opt, infos = self.calibrate_fields(data)
        torch.cuda.synchronize(device)
        logger.info(f"Optimization took {(time.time() - start) * 1000:.2f} ms")
        logger.info(f"Initial camera:\n{rad2deg(camera_init.vfov)}")
        logger.info(f"Optimized camera:\n{rad2deg(camera_opt.vfov)}")
        logger.info(f"Initial gravity:\n{rad2deg(gravity_init.rp)}")
        logger.info(f"Optimized gravity:\n{rad2deg(gravity_opt.rp)}")
        return {"camera": camera_opt, "gravity": gravity_opt, **infos}

    _MAX_CALL = 65535      # images per C call (grid.y of the sweep)

    def _calibrate_chunked(self, data: Dict[str, torch.Tensor], B: int):
        """Batches beyond 65 535 images: independent images are solved in slices of one C call each.  The device
        early stop is per call -- per slice -- which is said once in a warning (parallel.calibrate_sharded, where the
        same would make results depend on the world size, refuses early_stop instead)."""
        if self.shared_intrinsics:
            raise ValueError(f"a shared-intrinsics batch is limited to {self._MAX_CALL} frames per call")
        if self.conf.early_stop and not self._warned_chunked_stop:
            # the reference's stop is ONE decision over the whole batch (lm_optimizer.py:90-92, 619-625); here every slice
            # of 65 535 images takes its own, so `stop_at` (and the step a slice's images stop at) can differ per slice
            self._warned_chunked_stop = True
            logger.warning("geocalib_amd.LMOptimizer: a batch of %d images is solved in slices of %d; with early_stop=True "
                           "the batch-global stop is evaluated per slice (pass early_stop=False for slice-independent "
                           "results)", B, self._MAX_CALL)
        per_image = ("up_field", "latitude_field", "up_confidence", "latitude_confidence", "prior_focal",
                     "prior_gravity", "prior_dist")
        cams, gravs, infos, raws = [], [], [], []
        for lo in range(0, B, self._MAX_CALL):
            part = {k: (v[lo:lo + self._MAX_CALL] if k in per_image else v) for k, v in data.items()}
            c, g, i = self.calibrate_fields(part)
            cams.append(c._data); gravs.append(g._data); infos.append(i); raws.append(self._last_raw)
        self._last_raw = tuple(torch.cat([r[j] for r in raws]) for j in range(3))
        info = {k: torch.cat([i[k] for i in infos]) for k in infos[0]}
        return self.camera_model(torch.cat(cams)), _unit_gravity(torch.cat(gravs)), info

    def calibrate_fields(self, data: Dict[str, torch.Tensor]) -> Tuple[BaseCamera, Gravity, Dict[str, torch.Tensor]]:
        """get_trivial_estimation + optimize in ONE C call (gclm_calibrate): the initial estimate is built
        on the device from (H, W) and the priors, so no host-side tensor op precedes the kernels."""
        up, lat, upc, latc, (B, H, W) = self._fields(data)
        if B > self._MAX_CALL:
            return self._calibrate_chunked(data, B)
        device = lat.device
        stream = _raw_stream(device)      # looked up once: the handle's key and the launch stream
        h = self._handle(device, stream)

        def prior(key, shape):
            if key not in data:
                return None
            t = data[key]
            t = t._data if hasattr(t, "_data") else torch.as_tensor(t)
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            assert t.numel() == int(torch.tensor(shape).prod()), (key, t.shape, shape)
            return t

        scales = prior("scales", (2,))
        pf = prior("prior_focal", (B,))
        pg = prior("prior_gravity", (B, 3))
        nd = self.camera_model.num_dist_params() if self.camera_has_distortion else 0
        pd = None
        if "prior_dist" in data and nd:
            pd = data["prior_dist"].detach().to(device=device, dtype=torch.float32).contiguous()
            nd = pd.shape[-1] if pd.dim() > 1 else 1
        cam = torch.empty((B, 8), dtype=torch.float32, device=device)
        grav = torch.empty((B, 3), dtype=torch.float32, device=device)
        info = torch.empty((B, _lib.INFO_STRIDE), dtype=torch.float32, device=device)
        P = self._ptr           # (the library switches to the handle's device itself: no torch.cuda.device() context)
        n_over = self._overlap_parts(B, H, W, h, lambda: all(t is None or t.data_ptr() % 16 == 0 for t in (up, lat, upc, latc)))
        if n_over > 1:
            self._calibrate_overlapped(n_over, device, (up, lat, upc, latc), (B, H, W), scales, (pf, pg, pd), nd, (cam, grav, info))
        else:
            rc = _lib.load().gclm_calibrate(h.ptr, P(up), P(lat), P(upc), P(latc), B, H, W, P(scales), P(pf), P(pg),
                                            P(pd), nd, cam.data_ptr(), grav.data_ptr(), info.data_ptr(), stream)
            if rc != 0:
                _lib.check(rc, h.ptr, "gclm_calibrate")
        self._last_raw = (cam, grav, info)
        return self.camera_model(cam), _unit_gravity(grav), self._unpack_info(info, up is not None)

    # Host-side knob without a reference counterpart.  n > 1: a large batch of INDEPENDENT images with a fixed step count
    # is solved as n contiguous parts on n side streams (forked from / joined to the caller's stream with events), so that
    # one part's per-step update launch and kernel boundaries run under another part's sweep -- the 1.5 % between the
    # sweep's and the whole job's roofline fraction (pinhole +2.3 % at B = 1024, simple_radial +-0).
    # Images are independent, so the results are those of the single call up to the summation order of an image's partial
    # records -- bit-identical whenever the parts are cut like the whole batch (the cut depends on the batch size only
    # below 2048 workgroups per call: 137 images of 640x480).
    #   None (default): the library decides -- 2 parts when every part keeps >= `_OVERLAP_MIN_IMAGES` images AND is far
    #                   inside the regime where the cut does not depend on the batch size (`_OVERLAP_AUTO_PIXELS` pixels per
    #                   part: 410 images of 640x480), i.e. only where the result is the single call's bit for bit; 1 otherwise.
    #   1: never (measurement rigs that time single launches: overlapping launches have no separable durations).
    #   n > 1: n parts wherever every part keeps `_OVERLAP_MIN_IMAGES` images (where the library would cut a part differently
    #          from the whole batch -- gclm_plan_cut -- the split still happens and a warning says, once, that the results
    #          then equal the single call's only up to summation order).
    # Ignored where it does not apply: early_stop (one decision over the whole batch), shared intrinsics.
    overlap_streams = None
    _OVERLAP_MIN_IMAGES = 256
    _OVERLAP_AUTO_PIXELS = 3 * 2048 * 20480      # 3 x (2048 workgroups x 20 480 pixels per workgroup at 20 iterations)

    def _overlap_parts(self, B: int, H: int = 480, W: int = 640, handle=None, aligned16=True) -> int:
        """How many side-stream parts this batch is solved as.  `aligned16`: a bool, or a callable evaluated only once the
        cheap early-outs (early stop, shared intrinsics, batch size) have passed."""
        if self.conf.early_stop or self.shared_intrinsics:
            return 1
        auto = self.overlap_streams is None
        if auto:
            if not (B // 2 >= self._OVERLAP_MIN_IMAGES and (B // 2) * H * W >= self._OVERLAP_AUTO_PIXELS):
                return 1
            n = 2
        else:
            n = int(self.overlap_streams)
            n = max(1, min(n, B // self._OVERLAP_MIN_IMAGES)) if n > 1 else 1
        if n <= 1 or handle is None:
            return n
        # the single call's bits are promised only where every part is cut like the whole batch: the LIBRARY says how it would
        # cut each (gclm_plan_cut; the side-stream handles share this handle's configuration) -- no mirrored rule
        if callable(aligned16):
            aligned16 = aligned16()
        cuts = set()
        for size in {B} | {B * (i + 1) // n - B * i // n for i in range(n)}:
            rows = _lib.C.c_int(0)
            _lib.check(_lib.load().gclm_plan_cut(handle.ptr, size, H, W, int(aligned16), _lib.C.byref(rows), None), handle.ptr,
                       "gclm_plan_cut")
            cuts.add(rows.value)
        if len(cuts) == 1:
            return n
        if auto:
            return 1
        if not self.__dict__.get("_warned_overlap_cut"):
            # an explicit request is honoured, and said once: the parts' partial records are summed in another order
            self.__dict__["_warned_overlap_cut"] = True
            logger.warning("geocalib_amd.LMOptimizer.overlap_streams=%d: parts of a batch of %d images of %dx%d are not cut like "
                           "the whole batch; results equal the single call's up to float32 summation order, not bit for bit "
                           "(overlap_streams=None only splits where they are bit-identical)", n, B, W, H)
        return n

    def _calibrate_overlapped(self, n, device, fields, shape, scales, priors, nd, outs):
        B, H, W = shape
        cur = torch.cuda.current_stream(device)
        side = self.__dict__.setdefault("_side_streams", {})
        streams = side.get(device.index)
        if streams is None or len(streams) < n:
            streams = side[device.index] = [torch.cuda.Stream(device=device) for _ in range(n)]
        fork = cur.record_event()
        lib, P = _lib.load(), self._ptr
        bounds = [B * i // n for i in range(n + 1)]

        def part(t, lo, hi):
            return None if t is None else t[lo:hi]

        handles = []
        for i in range(n):
            lo, hi = bounds[i], bounds[i + 1]
            s = streams[i]
            s.wait_event(fork)                              # the caller's stream produced the fields
            h = self._handle(device, s.cuda_stream)
            up, lat, upc, latc = (part(t, lo, hi) for t in fields)
            pf, pg, pd = (part(t, lo, hi) for t in priors)
            cam, grav, info = (t[lo:hi] for t in outs)
            rc = lib.gclm_calibrate(h.ptr, P(up), P(lat), P(upc), P(latc), hi - lo, H, W, P(scales), P(pf), P(pg), P(pd), nd,
                                    cam.data_ptr(), grav.data_ptr(), info.data_ptr(), s.cuda_stream)
            if rc != 0:
                _lib.check(rc, h.ptr, "gclm_calibrate")
            handles.append(h)
            cur.wait_event(s.record_event())                # join: whatever follows on the caller's stream sees the results
        # infos["stop_at"] is ONE number for the whole batch (the first step after which every image's cost was close,
        # lm_optimizer.py:619-620): re-derived from the sum of the parts' per-step counters, on the caller's stream
        C = _lib.C
        info = outs[2]
        parts = (C.c_void_p * n)(*[h.ptr.value for h in handles])
        infos = (C.c_void_p * n)(*[info[bounds[i]:bounds[i + 1]].data_ptr() for i in range(n)])
        sizes = (C.c_int * n)(*[bounds[i + 1] - bounds[i] for i in range(n)])
        rc = lib.gclm_merge_stop_at(parts, infos, sizes, n, cur.cuda_stream)
        if rc != 0:
            _lib.check(rc, handles[0].ptr, "gclm_merge_stop_at")

    # ------------------------------------------------------------------ kernel-level entry (tests, tools)
    def system(self, data: Dict[str, torch.Tensor], camera: BaseCamera, gravity: Gravity,
               as_rpf: bool = False) -> Dict[str, torch.Tensor]:
        """One fused sweep at fixed parameters: mean costs, J^T W r (B,P_full) and J^T W J
        (reference: calculate_residuals + calculate_costs + setup_system, lm_optimizer.py:248-461)."""
        up, lat, upc, latc, (B, H, W) = self._fields(data)
        device = lat.device
        h = self._handle(device)
        cam = _dev_f32(camera._data, "camera")
        grav = _dev_f32(gravity._data, "gravity")
        cost = torch.empty((B, 2), dtype=torch.float32, device=device)
        grad = torch.empty((B, _lib.MAX_PARAMS), dtype=torch.float32, device=device)
        hess = torch.empty((B, _lib.MAX_PARAMS, _lib.MAX_PARAMS), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = _lib.load().gclm_system(h.ptr, self._ptr(up), self._ptr(lat), self._ptr(upc), self._ptr(latc),
                                         B, H, W, cam.data_ptr(), grav.data_ptr(), int(as_rpf),
                                         cost.data_ptr(), grad.data_ptr(), hess.data_ptr(), stream)
        _lib.check(rc, h.ptr, "gclm_system")
        P = 3 + (self.camera_model.num_dist_params() if self.camera_has_distortion else 0)
        return {"cost_up": cost[:, 0], "cost_lat": cost[:, 1], "G": grad[:, :P], "H": hess[:, :P, :P]}

    # ------------------------------------------------------------------ the reference's per-pixel stages as tensors
    def calculate_residuals(self, camera: BaseCamera, gravity: Gravity,
                            data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """`up_residual` (B,N,2) = up_field - prediction, `latitude_residual` (B,N,1) = sin(latitude_field) -
        sin(prediction), pixels in row-major order (reference: lm_optimizer.py:248-274).  A solve never
        materialises these; this is the sweep's pixel code writing them out (gclm_residual_fields)."""
        lat = _dev_f32(data["latitude_field"], "latitude_field") if "latitude_field" in data else None
        up = _dev_f32(data["up_field"], "up_field") if "up_field" in data else None
        ref = lat if lat is not None else up
        assert ref is not None, "data holds neither an up nor a latitude field"
        B, _, H, W = ref.shape
        device = ref.device
        cam = _dev_f32(camera._data, "camera").reshape(-1, 8)
        grav = _dev_f32(gravity._data, "gravity").reshape(-1, 3)
        r_up = torch.empty((B, H * W, 2), dtype=torch.float32, device=device) if up is not None else None
        r_lat = torch.empty((B, H * W, 1), dtype=torch.float32, device=device) if lat is not None else None
        with torch.cuda.device(device):
            rc = _lib.load().gclm_residual_fields(_lib.CAMERA_MODEL_IDS[self.camera_model.name()], self._ptr(up),
                                                  self._ptr(lat), cam.data_ptr(), grav.data_ptr(), B, H, W,
                                                  self._ptr(r_up), self._ptr(r_lat),
                                                  torch.cuda.current_stream(device).cuda_stream)
        if rc != 0:
            raise _lib.GclmError(f"gclm_residual_fields failed ({rc})")
        out = {}
        if r_up is not None:
            out["up_residual"] = r_up
        if r_lat is not None:
            out["latitude_residual"] = r_lat
        return out

    def calculate_costs(self, residuals: Dict[str, torch.Tensor], data: Dict[str, torch.Tensor]):
        """(costs, weights): scaled Huber of the squared residual norms, times the confidences when present
        (reference: lm_optimizer.py:276-315).  Keys `up_cost` / `latitude_cost`, `up_weights` / `latitude_weights`."""
        costs, weights = {}, {}
        for key, conf_key, scale, ckey, wkey in (
                ("up_residual", "up_confidence", self.conf.up_loss_fn_scale, "up_cost", "up_weights"),
                ("latitude_residual", "latitude_confidence", self.conf.lat_loss_fn_scale, "latitude_cost", "latitude_weights")):
            if key not in residuals:
                continue
            r = _dev_f32(residuals[key], key)
            B, N, dim = r.shape
            conf = _dev_f32(data[conf_key], conf_key).reshape(B, N) if conf_key in data else None
            cost, weight = torch.empty((B, N), dtype=torch.float32, device=r.device), torch.empty((B, N), dtype=torch.float32, device=r.device)
            scale = self._SQUARED_LOSS_SCALE if self.conf.loss_fn == "squared_loss" else float(scale)
            with torch.cuda.device(r.device):
                rc = _lib.load().gclm_huber_costs(r.data_ptr(), B * N, dim, scale, self._ptr(conf), cost.data_ptr(),
                                                  weight.data_ptr(), None, torch.cuda.current_stream(r.device).cuda_stream)
            if rc != 0:
                raise _lib.GclmError(f"gclm_huber_costs failed ({rc})")
            costs[ckey], weights[wkey] = cost, weight
        return costs, weights

    def update_estimate(self, camera: BaseCamera, gravity: Gravity, delta: torch.Tensor):
        """Apply one LM step to (camera, gravity) along the columns planned by setup_optimization_and_priors
        (reference: lm_optimizer.py:518-549).  Inside a solve the per-image kernel does this; the host form composes
        Gravity.update / update_focal / update_dist for callers that drive their own loop."""
        zeros = lambda n: delta.new_zeros(delta.shape[:-1] + (n,))  # noqa: E731
        d_g = delta[..., list(self.gravity_delta_dims)] if self.estimate_gravity else zeros(2)
        new_gravity = gravity.update(d_g, spherical=self.conf.use_spherical_manifold)
        d_f = delta[..., list(self.focal_delta_dims)] if self.estimate_focal else zeros(1)
        new_camera = camera.update_focal(d_f, as_log=self.conf.use_log_focal)
        if self.camera_has_distortion and self.estimate_dist:
            new_camera = new_camera.update_dist(delta[..., list(self.dist_delta_dims)])
        return new_camera, new_gravity

    def _column_dims(self):
        """Columns of the full per-pixel Jacobian [d1, d2, focal, k...] that enter a system (lm_optimizer.py:336-343)."""
        dims = (0, 1) if self.estimate_gravity else ()
        if self.estimate_focal:
            dims += (2,)
        if self.camera_has_distortion:
            dims += tuple(range(3, 3 + self.camera_model.num_dist_params()))
        assert dims, "No parameters to optimize"
        return list(dims)

    def calculate_gradient_and_hessian(self, J: torch.Tensor, residuals: torch.Tensor, weights: torch.Tensor,
                                       shared_intrinsics: bool = False):
        """Grad = sum_px w J^T r (B,P), Hess = sum_px w J^T J (B,P,P) for materialised J (B,N,R,P_full), residuals
        (B,N,R), weights (B,N) (reference: lm_optimizer.py:317-385; contraction on the device, gclm_gradient_hessian).
        shared_intrinsics=True returns the reference's arrow-head layout: Grad (1, 2B+ni), Hess (1, 2B+ni, 2B+ni)."""
        for t in (J, residuals, weights):
            if not t.is_cuda:
                raise RuntimeError("calculate_gradient_and_hessian needs HIP device tensors (no CPU fallback)")
        Jc = J.detach().to(torch.float32)[..., self._column_dims()].contiguous()
        B, N, R, P = Jc.shape
        r = residuals.detach().to(torch.float32).reshape(B, N, R).contiguous()
        w = weights.detach().to(torch.float32).reshape(B, N).contiguous()
        Grad = torch.empty((B, P), dtype=torch.float32, device=Jc.device)
        Hess = torch.empty((B, P, P), dtype=torch.float32, device=Jc.device)
        with torch.cuda.device(Jc.device):
            rc = _lib.load().gclm_gradient_hessian(Jc.data_ptr(), r.data_ptr(), w.data_ptr(), B, N, R, P, 0, Grad.data_ptr(),
                                                   Hess.data_ptr(), torch.cuda.current_stream(Jc.device).cuda_stream)
        if rc != 0:
            raise _lib.GclmError(f"gclm_gradient_hessian failed ({rc})")
        if not shared_intrinsics:
            return Grad, Hess
        # arrow-head assembly: per-frame gravity blocks on the diagonal, summed intrinsics in the last rows/columns
        ni = P - 2
        n = 2 * B + ni
        G = torch.cat([Grad[:, :2].reshape(-1), Grad[:, 2:].sum(0)])[None]
        Hs = Hess.new_zeros((n, n))
        for b in range(B):
            Hs[2 * b:2 * b + 2, 2 * b:2 * b + 2] = Hess[b, :2, :2]
        Hs[:2 * B, 2 * B:] = Hess[:, :2, 2:].reshape(2 * B, ni)
        Hs[2 * B:, :2 * B] = Hess[:, 2:, :2].permute(1, 0, 2).reshape(ni, 2 * B)
        Hs[2 * B:, 2 * B:] = Hess[:, 2:, 2:].sum(0)
        return G, Hs[None]

    def setup_system(self, camera: BaseCamera, gravity: Gravity, residuals: Dict[str, torch.Tensor],
                     weights: Dict[str, torch.Tensor], as_rpf: bool = False, shared_intrinsics: bool = False):
        """(Grad, Hess) from given per-pixel residuals and weights (reference: lm_optimizer.py:387-461): the Jacobian
        fields come from gclm_jacobian_fields, the contraction from gclm_gradient_hessian.  The solve uses the fused
        sweep instead (`system`), which never materialises either."""
        from .perspective_fields import J_perspective_field
        J_up, J_lat = J_perspective_field(camera, gravity, spherical=self.conf.use_spherical_manifold and not as_rpf,
                                          log_focal=self.conf.use_log_focal and not as_rpf)
        Grad = Hess = None
        for J, rkey, wkey in ((J_up, "up_residual", "up_weights"), (J_lat, "latitude_residual", "latitude_weights")):
            if rkey not in residuals:
                continue
            Jn = J.reshape(J.shape[0], -1, J.shape[-2], J.shape[-1])
            g, h = self.calculate_gradient_and_hessian(Jn, residuals[rkey], weights[wkey], shared_intrinsics)
            Grad, Hess = (g, h) if Grad is None else (Grad + g, Hess + h)
        assert Grad is not None, "residuals hold neither an up nor a latitude entry"
        return Grad, Hess

    def estimate_uncertainty(self, camera_opt: BaseCamera, gravity_opt: Gravity, errors: Dict[str, torch.Tensor],
                             weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Covariance = inverse Hessian in (roll, pitch, focal[, dist]) and the derived standard deviations
        (reference: lm_optimizer.py:463-516).  `forward` computes the same quantities inside its finalize kernel."""
        from .misc import J_focal2fov
        _, Hess = self.setup_system(camera_opt, gravity_opt, errors, weights, as_rpf=True, shared_intrinsics=False)
        Cov = torch.linalg.inv(Hess)
        zero = Cov.new_zeros(Cov.shape[:-2])
        roll = pitch = grav = focal = fov = zero
        if self.estimate_gravity:
            roll, pitch = Cov[..., 0, 0], Cov[..., 1, 1]
            a, b, c = Cov[..., 0, 0], 0.5 * (Cov[..., 0, 1] + Cov[..., 1, 0]), Cov[..., 1, 1]
            grav = 0.5 * (a + c) + torch.sqrt((0.5 * (a - c)) ** 2 + b ** 2)       # largest eigenvalue of the 2x2 block
        if self.estimate_focal:
            i = self.focal_delta_dims[0]
            focal = Cov[..., i, i]
            fov = J_focal2fov(camera_opt.f[..., 1], camera_opt.size[..., 1]) ** 2 * focal
        return {"covariance": Cov, "roll_uncertainty": torch.sqrt(roll), "pitch_uncertainty": torch.sqrt(pitch),
                "gravity_uncertainty": torch.sqrt(grav), "focal_uncertainty": torch.sqrt(focal) / 2,
                "vfov_uncertainty": torch.sqrt(fov / 2)}
