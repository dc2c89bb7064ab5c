"""GPU box: does the HIP virtual-memory-management API give control over the physical-placement lottery of the sweep?

    python scripts/probes/vmm_probe.py [--maps 10] [--model pinhole] [--batch 1024]

The memory-bound pinhole sweep runs 928 ... 1005 us on the same 6.3 GB depending on which physical pages back the four
input tensors (profiles/archive/r03_placement_probe2.log); virtual layout and padding do not explain it.  Here the five planes
of a B=1024 batch are backed by hipMemCreate physical handles mapped with hipMemAddressReserve / hipMemMap:

    malloc        control: one hipMalloc per tensor (what torch's allocator does for blocks this large)
    vmm-one       ONE physical handle for all 6.3 GB
    vmm-tensor    one handle per tensor (up | lat | up_conf | lat_conf), contiguous virtual range
    vmm-plane     one handle per PLANE (up_x and up_y are interleaved per image, so: per tensor, up split in two halves)
    vmm-2M/64M/1G handles of that size (2 MiB = the minimum granularity on gfx950)
    vmm-one-1Galign  as vmm-one, every tensor starting on a 1 GiB boundary of the virtual range

Every variant is mapped `--maps` times with ALL its mappings alive at once (so they are on different pages), the same
synthetic content is generated into each (gclm_synth_fields), and the sweep is timed with the library's HIP events over
3 solves of 20 LM steps.  Output: per variant the sweep times, their spread (max/min - 1) and mean."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from geocalib_amd import _lib  # noqa: E402

args = sys.argv[1:]


def opt(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]
        return v
    return default


MAPS = int(opt("--maps", "10"))
MODEL = opt("--model", "pinhole")
B = int(opt("--batch", "1024"))
ONLY = opt("--only", "")
VERBOSE = "--verbose" in args
H, W = 480, 640
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_build", "libvmm_probe.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                    os.path.join(here, "vmm_alloc.hip"), "-o", so], check=True)
vmm = C.CDLL(so)
vmm.vmm_granularity.restype = C.c_size_t
vmm.vmm_granularity.argtypes = [C.c_int, C.c_int]
vmm.vmm_alloc.restype = C.c_int
vmm.vmm_alloc.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p),
                          C.POINTER(C.c_void_p)]
vmm.vmm_free.argtypes = [C.c_void_p]
vmm.vmm_touch.argtypes = [C.c_void_p, C.c_size_t]
lib = _lib.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
stream = torch.cuda.current_stream(dev).cuda_stream
print(f"granularity: minimum {vmm.vmm_granularity(0, 0)} B, recommended {vmm.vmm_granularity(0, 1)} B", flush=True)

plane = B * H * W * 4
SEGS = [2 * plane, plane, plane, plane]               # up_field (B,2,H,W), latitude, up_confidence, latitude_confidence
MiB, GiB = 1 << 20, 1 << 30
VARIANTS = [("malloc", -1, 0, 0), ("vmm-one", 0, 0, 0), ("vmm-tensor", 1, 0, 0), ("vmm-2M", 2, 2 * MiB, 0),
            ("vmm-64M", 2, 64 * MiB, 0), ("vmm-1G", 2, GiB, 0), ("vmm-one-1Galign", 0, 0, GiB), ("malloc-again", -1, 0, 0)]
if ONLY:
    VARIANTS = [v for v in VARIANTS if v[0] in ONLY.split(",")]

cfg = _lib.GclmConfig.default(0)
cfg.camera_model = _lib.CAMERA_MODEL_IDS[MODEL]
cfg.num_steps, cfg.early_stop = 20, 0
h = C.c_void_p()
_lib.check(lib.gclm_create(C.byref(h), C.byref(cfg)), None, "gclm_create")
cam, grav = torch.empty(B, 8, device=dev), torch.empty(B, 3, device=dev)
info = torch.empty(B, _lib.INFO_STRIDE, device=dev)
gtc, gtg = torch.empty(B, 8, device=dev), torch.empty(B, 3, device=dev)


def sweep_us(ptrs, solves=3):
    up, lat, upc, latc = ptrs

    def solve():
        _lib.check(lib.gclm_calibrate(h, up, lat, upc, latc, B, H, W, None, None, None, None, 0, cam.data_ptr(), grav.data_ptr(),
                                      info.data_ptr(), stream), h, "gclm_calibrate")
    solve()
    torch.cuda.synchronize()
    lib.gclm_set_timing(h, 1)
    for _ in range(solves):
        solve()
    torch.cuda.synchronize()
    n, ms = C.c_int(0), C.c_float(0)
    lib.gclm_last_pass_timing(h, C.byref(n), C.byref(ms))
    lib.gclm_set_timing(h, 0)
    return ms.value / n.value * 1e3


summary = []
for name, mode, chunk, align in VARIANTS:
    blocks, times, retimed = [], [], []
    for m in range(MAPS):
        sizes = (C.c_size_t * 4)(*SEGS)
        ptrs = (C.c_void_p * 4)()
        blk = C.c_void_p()
        rc = vmm.vmm_alloc(0, 4, sizes, mode, chunk, align, ptrs, C.byref(blk))
        if rc != 0:
            print(f"{name}: vmm_alloc failed ({rc}) at mapping {m}", flush=True)
            break
        blocks.append((blk, list(ptrs)))
        if VERBOSE:
            print(f"{name}[{m}]: segments at " + " ".join(hex(p) for p in ptrs), flush=True)
            for p, nbytes in zip(ptrs, SEGS):
                assert vmm.vmm_touch(p, nbytes) == 0
            print(f"{name}[{m}]: memset of every segment ok", flush=True)
        _lib.check(lib.gclm_synth_fields(cfg.camera_model, 1, 0, B, H, W, 0.02, ptrs[0], ptrs[1], ptrs[2], ptrs[3],
                                         gtc.data_ptr(), gtg.data_ptr(), stream), None, "gclm_synth_fields")
        torch.cuda.synchronize()
        times.append(sweep_us(list(ptrs)))
        ferr = (cam[:, 3] / gtc[:, 3] - 1).abs().median().item()
        assert ferr < 5e-3, ferr
    for blk, ptrs in blocks:                  # reproducibility of each mapping
        retimed.append(sweep_us(ptrs, 2))
    for blk, _ in blocks:
        vmm.vmm_free(blk)
    if times:
        spread = max(times) / min(times) - 1
        summary.append((name, times, spread))
        print(f"{name:16s} sweep us: " + " ".join(f"{t:6.1f}" for t in times) +
              f"   min {min(times):6.1f} mean {sum(times) / len(times):6.1f} max {max(times):6.1f}  spread {spread * 100:4.1f} %", flush=True)
        print(f"{'':16s} re-timed: " + " ".join(f"{t:6.1f}" for t in retimed), flush=True)
print("\nsummary (spread = max/min - 1 over the mappings of a variant):")
for name, times, spread in summary:
    print(f"  {name:16s} min {min(times):6.1f}  mean {sum(times) / len(times):6.1f}  max {max(times):6.1f}  spread {spread * 100:4.1f} %")
