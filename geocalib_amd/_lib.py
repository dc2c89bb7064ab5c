"""ctypes binding of libgeocalib_hip.so (C ABI: include/gclm.h).

There is no CPU fallback: if the shared library is missing or no HIP device is visible the
product path raises.  Build the library with `python __graft_entry__.py` (or `make -C
geocalib_amd/csrc`); it is kept in-tree at geocalib_amd/lib/libgeocalib_hip.so.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GCLM_LIB_PATH: measurement rigs only (same-box A/B of two builds of the library, scripts/ab_lib.sh)
LIB_PATH = os.environ.get("GCLM_LIB_PATH") or os.path.join(_HERE, "lib", "libgeocalib_hip.so")

CAMERA_MODEL_IDS = {"pinhole": 0, "simple_radial": 1, "radial": 2, "simple_divisional": 3}
INFO_STRIDE = 48
SHARED_PARTIAL_STRIDE = 32
COMM_ID_BYTES = 128
MAX_PARAMS = 5
ABI_VERSION = 610          # GCLM_VERSION of include/gclm.h this binding was written against
INFO = {"stop_at": 0, "initial_up_cost": 1, "initial_latitude_cost": 2, "initial_cost": 3,
        "final_up_cost": 4, "final_latitude_cost": 5, "final_cost": 6, "roll_uncertainty": 7,
        "pitch_uncertainty": 8, "gravity_uncertainty": 9, "focal_uncertainty": 10,
        "vfov_uncertainty": 11, "n_params": 12, "lambda": 13, "step_failures": 14, "covariance": 16}


class GclmConfig(C.Structure):
    """struct gclm_config (include/gclm.h)."""

    _fields_ = [("struct_size", C.c_int32), ("abi_version", C.c_int32), ("device", C.c_int32),
                ("camera_model", C.c_int32), ("shared_intrinsics", C.c_int32),
                ("group_size", C.c_int32), ("num_steps", C.c_int32), ("lambda0", C.c_float),
                ("fix_lambda", C.c_int32), ("early_stop", C.c_int32), ("atol", C.c_float),
                ("rtol", C.c_float), ("use_spherical_manifold", C.c_int32),
                ("use_log_focal", C.c_int32), ("up_loss_fn_scale", C.c_float),
                ("lat_loss_fn_scale", C.c_float), ("estimate_gravity", C.c_int32),
                ("estimate_focal", C.c_int32), ("estimate_dist", C.c_int32),
                ("compute_uncertainty", C.c_int32), ("heuristic_init", C.c_int32)]

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)

    @classmethod
    def default(cls, device: int = 0) -> "GclmConfig":
        """gclm_default_config: LMOptimizer.default_conf plus the library's struct_size / abi_version stamp."""
        cfg = cls()
        if load().gclm_default_config(C.byref(cfg)) != 0:
            raise GclmError("gclm_default_config failed")
        cfg.device = int(device)
        return cfg


class GclmError(RuntimeError):
    """A gclm_* entry point returned a non-zero code."""


_lib = None

_P = C.c_void_p
_SIGNATURES = {
    "gclm_version": (C.c_int, []),
    "gclm_default_config": (C.c_int, [C.POINTER(GclmConfig)]),
    "gclm_abi_config_size": (C.c_int, []),
    "gclm_create": (C.c_int, [C.POINTER(_P), C.POINTER(GclmConfig)]),
    "gclm_configure": (C.c_int, [_P, C.POINTER(GclmConfig)]),
    "gclm_destroy": (C.c_int, [_P]),
    "gclm_last_error": (C.c_char_p, [_P]),
    "gclm_workspace_bytes": (C.c_size_t, [_P]),
    "gclm_solve": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "gclm_calibrate": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P]),
    "gclm_system": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P]),
    "gclm_shared_begin": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P]),
    "gclm_shared_reduce": (C.c_int, [_P, C.c_int, _P, _P]),
    "gclm_shared_apply": (C.c_int, [_P, C.c_int, _P, _P]),
    "gclm_shared_finish": (C.c_int, [_P, _P, _P]),
    "gclm_upsample_fields": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "gclm_upsample_fields_multi": (C.c_int, [C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, _P]),
    "gclm_gradient_hessian": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "gclm_optimizer_step": (C.c_int, [_P, _P, _P, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P, _P]),
    "gclm_residual_fields": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "gclm_huber_costs": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_float, _P, _P, _P, _P, _P]),
    "gclm_jacobian_fields": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "gclm_pack_fields": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "gclm_synth_fields": (C.c_int, [C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float,
                                    _P, _P, _P, _P, _P, _P, _P]),
    "gclm_synth_fields_grouped": (C.c_int, [C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "gclm_comm_unique_id": (C.c_int, [_P]),
    "gclm_comm_create": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int, C.c_int]),
    "gclm_comm_destroy": (C.c_int, [_P]),
    "gclm_comm_last_error": (C.c_char_p, [_P]),
    "gclm_comm_versions": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gclm_comm_all_gather": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "gclm_comm_all_reduce_sum": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "gclm_comm_all_reduce_sum_i32": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "gclm_set_stop_comm": (C.c_int, [_P, _P]),
    "gclm_merge_stop_at": (C.c_int, [C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int), C.c_int, _P]),
    "gclm_set_sweep_iters": (C.c_int, [_P, C.c_int]),
    "gclm_set_slat_plane": (C.c_int, [_P, C.c_int]),
    "gclm_set_slat_plane_limit": (C.c_int, [_P, C.c_size_t]),
    "gclm_slat_plane_bytes": (C.c_size_t, [_P]),
    "gclm_release_workspace": (C.c_int, [_P]),
    "gclm_read_probe": (C.c_int, [C.POINTER(_P), C.c_int, C.c_size_t, _P]),
    "gclm_plan_cut": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gclm_set_row_pairs": (C.c_int, [_P, C.c_int]),
    "gclm_set_fused_steps": (C.c_int, [_P, C.c_int]),
    "gclm_set_paced_launches": (C.c_int, [_P, C.c_int]),
    "gclm_set_timing": (C.c_int, [_P, C.c_int]),
    "gclm_last_pass_timing": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load():
    """Load libgeocalib_hip.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py` "
                "or `make -C geocalib_amd/csrc`). geocalib_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        # a library of another ABI generation must not be driven with this binding's struct layout
        if lib.gclm_version() != ABI_VERSION or lib.gclm_abi_config_size() != C.sizeof(GclmConfig):
            raise ImportError(f"{LIB_PATH}: ABI {lib.gclm_version()} / sizeof(gclm_config) {lib.gclm_abi_config_size()}, "
                              f"this binding expects {ABI_VERSION} / {C.sizeof(GclmConfig)}: rebuild the library")
        _lib = lib
    return _lib


def last_error(handle=None) -> str:
    msg = load().gclm_last_error(handle)
    return msg.decode() if msg else ""


def check(rc: int, handle=None, what: str = "gclm"):
    if rc != 0:
        raise GclmError(f"{what} failed ({rc}): {last_error(handle)}")
