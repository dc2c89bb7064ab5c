"""Probe (GPU box): several BUILDS of the library on the SAME device tensors in ONE process.

    python scripts/variant_probe.py [--models pinhole,simple_radial] [--batch 1024] [--reps 3] name=path.so[@sweep_iters[@slat_plane]] [...]

(`slat_plane`: gclm_set_slat_plane mode -1 / 0 / 1 for libraries that have it (ABI >= 500) -- the same library can appear
twice under two names to compare the sin(latitude) scratch plane on and off on one allocation.)

Where a batch lands in physical memory moves the memory-bound sweep by up to 8 % from one process to the next,
so two builds can only be compared on one allocation: every library is loaded side by side (ctypes), each solves the same
tensors with 20 fixed steps, interleaved `reps` times, and reports its mean sweep time (HIP events inside the library).
Build the variants in the container (hipcc cross-compiles): scripts/build_variants.sh."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from geocalib_amd import _lib  # noqa: E402
from geocalib_amd.synth import synth_fields  # noqa: E402

args = sys.argv[1:]


def opt(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]
        return v
    return default


models = opt("--models", "pinhole,simple_radial").split(",")
B = int(opt("--batch", "1024"))
reps = int(opt("--reps", "3"))
allocations = int(opt("--allocations", "1"))       # repeat everything on N allocations of the fields (earlier ones stay alive: other pages)
H, W = 480, 640
libs = []
for a in args:
    name, path = a.split("=", 1)
    path, _, rest = path.partition("@")
    iters, _, slat = rest.partition("@")
    lib = C.CDLL(os.path.abspath(path))
    for fn, (res, at) in _lib._SIGNATURES.items():
        f = getattr(lib, fn, None)                     # (an older build of the library lacks this round's entry points)
        if f is not None:
            f.restype, f.argtypes = res, at
    assert lib.gclm_version() in (400, _lib.ABI_VERSION) and lib.gclm_abi_config_size() == C.sizeof(_lib.GclmConfig)
    libs.append((name, lib, int(iters or 0), int(slat) if slat else None))
dev = torch.device("cuda:0")
keep = []
for model in [m for m in models for _ in range(allocations)]:
    data, gtc, _ = synth_fields(model, B, H, W, dev, seed=1)
    if allocations > 1:
        keep.append(data)
        print(f"-- allocation {len(keep)}: up_field at {data['up_field'].data_ptr():#x}")
    up, lat, upc, latc = (data[k].contiguous() for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence"))
    handles, outs = {}, {}
    for name, lib, iters, slat in libs:
        cfg = _lib.GclmConfig()
        assert lib.gclm_default_config(C.byref(cfg)) == 0
        cfg.camera_model = _lib.CAMERA_MODEL_IDS[model]; cfg.num_steps = 20; cfg.early_stop = 0
        h = C.c_void_p()
        assert lib.gclm_create(C.byref(h), C.byref(cfg)) == 0, lib.gclm_last_error(None)
        assert lib.gclm_set_sweep_iters(h, iters) == 0
        if slat is not None:
            assert lib.gclm_set_slat_plane(h, slat) == 0
        handles[name] = h
        outs[name] = (torch.empty(B, 8, device=dev), torch.empty(B, 3, device=dev), torch.empty(B, _lib.INFO_STRIDE, device=dev))
    stream = torch.cuda.current_stream(dev).cuda_stream
    times = {name: [] for name, *_ in libs}
    solves = {name: [] for name, *_ in libs}
    for rep in range(reps + 1):
        for name, lib, *_ in libs:
            h = handles[name]; cam, grav, info = outs[name]
            lib.gclm_set_timing(h, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rc = lib.gclm_calibrate(h, up.data_ptr(), lat.data_ptr(), upc.data_ptr(), latc.data_ptr(), B, H, W, None, None, None,
                                        None, 0, cam.data_ptr(), grav.data_ptr(), info.data_ptr(), stream)
                assert rc == 0, lib.gclm_last_error(h)
            e1.record()
            torch.cuda.synchronize()
            n, ms = C.c_int(0), C.c_float(0)
            assert lib.gclm_last_pass_timing(h, C.byref(n), C.byref(ms)) == 0
            if rep:                                   # the first round is warm-up
                times[name].append(ms.value / n.value)
                solves[name].append(e0.elapsed_time(e1) / 3)
    ref = outs[libs[0][0]]
    for name, lib, *_ in libs:
        t = times[name]
        same = all(torch.equal(a, b) or torch.allclose(a, b, equal_nan=True, rtol=0, atol=0) for a, b in zip(outs[name], ref))
        print(f"{model:18s} {name:14s} sweep {sum(t)/len(t)*1e3:8.1f} us  (reps " + " ".join(f"{x*1e3:.1f}" for x in t) +
              f")  = {B*H*W*20/(sum(t)/len(t))/1e9:5.2f} TB/s   solve {sum(solves[name])/len(solves[name]):7.3f} ms = "
              f"{B/(sum(solves[name])/len(solves[name]))*1e3:7.0f} img/s   results identical to {libs[0][0]}: {same}", flush=True)
        lib.gclm_destroy(handles[name])
    del data, up, lat, upc, latc
