"""GPU box: achieved HBM bandwidth of the two kernels either side of the LM path:
gclm_pack_fields (CNN-head epilogue, reads 5 + writes 5 fp32 planes) and gclm_upsample_fields (_post_process bilinear).
usage: python scripts/probes/fields_probe.py [--json OUT]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd.fields import pack_fields, upsample_fields
dev = torch.device("cuda:0")
rows = []


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, H, W in ((256, 480, 640), (1024, 480, 640), (1, 480, 640)):
    g = torch.Generator(device=dev).manual_seed(1)
    up_raw = torch.randn((B, 2, H, W), device=dev, generator=g); lat_raw = torch.randn((B, 1, H, W), device=dev, generator=g)
    ulc = torch.randn((B, H, W), device=dev, generator=g); llc = torch.randn((B, H, W), device=dev, generator=g)
    ms = timeit(lambda: pack_fields(up_raw, lat_raw, ulc, llc, inplace=True))
    byt = B * H * W * 4 * 10
    rows.append({"kernel": "gclm_pack_fields (in place)", "B": B, "H": H, "W": W, "ms": round(ms, 4), "GB/s": round(byt / ms / 1e6, 1), "bytes": byt})
    print(rows[-1], flush=True)
    # the read + write streaming ceiling of the SAME four tensors in this process: x *= 1.0 in place (torch's vectorised
    # elementwise kernel, 4 launches; `one` is a device scalar, nothing to fold).  scripts/probes/pack_bench.hip holds the
    # hand-written copy with pack_fields' own launch shape (round 5's copy baseline had been optimised away)
    one = torch.ones((), device=dev)
    ms_c = timeit(lambda: (up_raw.mul_(one), lat_raw.mul_(one), ulc.mul_(one), llc.mul_(one)))
    rows.append({"kernel": "torch x *= 1.0 in place on the same tensors (read + write ceiling, 4 launches)", "B": B, "H": H, "W": W,
                 "ms": round(ms_c, 4), "GB/s": round(byt / ms_c / 1e6, 1), "bytes": byt})
    print(rows[-1], flush=True)
    # eager torch epilogue (what the reference's heads run), for scale
    def eager():
        u = torch.nn.functional.normalize(up_raw, dim=1); c1 = torch.sigmoid(ulc)
        l = torch.asin(torch.clamp(torch.tanh(lat_raw), -1 + 1e-5, 1 - 1e-5)); c2 = torch.sigmoid(llc)
        return u, c1, l, c2
    ms_e = timeit(eager, 5)
    rows.append({"kernel": "eager torch epilogue (8 kernels)", "B": B, "H": H, "W": W, "ms": round(ms_e, 4)})
    print(rows[-1], flush=True)
    del up_raw, lat_raw, ulc, llc
# x2 (rows of whole lines), x3.375 (ragged rows: phase-rotated stores), one image, x1.25 (no 4-float window: per-lane gathers)
for B, (h, w), (H, W) in ((256, (240, 320), (480, 640)), (64, (320, 480), (1080, 1620)), (1, (320, 480), (1080, 1620)),
                          (1, (240, 320), (480, 640)), (128, (320, 448), (400, 560))):
    src = torch.randn((B, 5, h, w), device=dev)
    ms = timeit(lambda: upsample_fields(src, (H, W)))
    byt = B * 5 * 4 * (h * w + H * W)
    rows.append({"kernel": "gclm_upsample_fields", "B": B, "src": [h, w], "dst": [H, W], "ms": round(ms, 4), "GB/s": round(byt / ms / 1e6, 1), "bytes": byt})
    print(rows[-1], flush=True)
    dstbuf = torch.empty((B, 5, H, W), device=dev)
    ms_f = timeit(lambda: dstbuf.fill_(1.0))
    rows.append({"kernel": "torch fill_ of the output (pure write, for scale)", "B": B, "dst": [H, W], "ms": round(ms_f, 4), "GB/s": round(dstbuf.numel() * 4 / ms_f / 1e6, 1)})
    print(rows[-1], flush=True)
    del dstbuf
    ms_t = timeit(lambda: torch.nn.functional.interpolate(src, (H, W), mode="bilinear", align_corners=False), 5)
    rows.append({"kernel": "torch F.interpolate bilinear", "B": B, "src": [h, w], "dst": [H, W], "ms": round(ms_t, 4), "GB/s": round(byt / ms_t / 1e6, 1)})
    print(rows[-1], flush=True)
if "--json" in sys.argv:
    out = sys.argv[sys.argv.index("--json") + 1]
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    extra = {}
    log = os.path.join(os.path.dirname(os.path.abspath(out)), "pack_bench.log")      # written by scripts/r06_collect.sh just before
    if os.path.exists(log):
        import re
        cur = None
        for ln in open(log):
            m = re.match(r"== B = (\d+)", ln)
            if m:
                cur = f"B{m.group(1)}"
            m = re.match(r"\s+(copy[^=]*?|nt both, grid 300 \(one pass\))\s+mean\s+([\d.]+) us .*= +([\d.]+) TB/s", ln)
            if m and cur:
                extra.setdefault(cur, {})[m.group(1).strip()] = {"mean_us": float(m.group(2)), "TB/s": float(m.group(3)), "frac_of_8": round(float(m.group(3)) / 8, 4)}
    json.dump({"pack_fields_read_write_ceiling": {"source": "scripts/probes/pack_bench.hip (same call): in-place copy x * 1.0f of the five planes with "
               "pack_fields' launch shape (grid 300 x B, one pass) vs the epilogue's arithmetic on the same shape", "rows": extra},
               "what": "kernels either side of the LM path; GB/s = algorithmic bytes (read + written once) / mean time over 20 calls (torch events; upsample includes its output allocation)", "rows": rows}, open(out, "w"), indent=1)
