"""The step before the LM path: CNN-head epilogue fused into one HIP pass (gclm_pack_fields).

Replaces the tail of the reference's UpDecoder / LatitudeDecoder (geocalib/geocalib.py:57,73-75):

    up_field            = F.normalize(up_raw, dim=1)
    up_confidence       = sigmoid(up_log_confidence)
    latitude_field      = asin(clamp(tanh(lat_raw), -1 + 1e-5, 1 - 1e-5))
    latitude_confidence = sigmoid(lat_log_confidence)

and writes the five planes in the layout LMOptimizer reads (eager PyTorch: 8 kernels, ~18 plane passes)."""
from typing import Dict, Optional

import torch

from . import _lib
from .lm_optimizer import _raw_stream


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on_device(device):
    """These entry points take no handle, so the launch goes to HIP's CURRENT device: switch only when the tensors live on
    another one (one process per GPU: never; the context manager costs 3-4 us of a 10 us single-image call)."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    return _NO_SWITCH if index == torch.cuda.current_device() else torch.cuda.device(device)


def pack_fields(up_raw: torch.Tensor, lat_raw: torch.Tensor, up_log_confidence: Optional[torch.Tensor] = None,
                lat_log_confidence: Optional[torch.Tensor] = None, inplace: bool = False) -> Dict[str, torch.Tensor]:
    """up_raw (B,2,H,W), lat_raw (B,1,H,W), log-confidences (B,H,W) or (B,1,H,W): raw head outputs on a HIP device.
    Returns the dict `LMOptimizer.forward` consumes."""
    for t in (up_raw, lat_raw):
        if not t.is_cuda:
            raise RuntimeError("geocalib_amd.pack_fields needs HIP device tensors (no CPU fallback)")
    B, _, H, W = lat_raw.shape
    assert up_raw.shape == (B, 2, H, W), up_raw.shape

    def prep(t):
        return None if t is None else t.detach().to(torch.float32).contiguous()

    up_raw, lat_raw, ulc, llc = prep(up_raw), prep(lat_raw), prep(up_log_confidence), prep(lat_log_confidence)
    for c in (ulc, llc):
        assert c is None or c.numel() == B * H * W, c.shape
    out_like = (lambda t: t) if inplace else torch.empty_like
    up, lat = out_like(up_raw), out_like(lat_raw)
    upc = None if ulc is None else out_like(ulc).view(B, H, W)
    latc = None if llc is None else out_like(llc).view(B, H, W)
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(lat_raw.device):
        rc = _lib.load().gclm_pack_fields(p(up_raw), p(ulc), p(lat_raw), p(llc), B, H, W, p(up), p(upc), p(lat), p(latc),
                                          torch.cuda.current_stream(lat_raw.device).cuda_stream)
    if rc != 0:
        raise _lib.GclmError(f"gclm_pack_fields failed ({rc})")
    out = {"up_field": up, "latitude_field": lat}
    if upc is not None:
        out["up_confidence"] = upc
    if latc is not None:
        out["latitude_confidence"] = latc
    return out


def upsample_fields(t: torch.Tensor, size) -> torch.Tensor:
    """Bilinear resize of the trailing (h, w) planes of `t` to `size` = (H, W), matching
    `torch.nn.functional.interpolate(t, size, mode="bilinear", align_corners=False)` (extractor.py:60-63)."""
    if not t.is_cuda:
        raise RuntimeError("geocalib_amd.upsample_fields needs a HIP device tensor (no CPU fallback)")
    H, W = int(size[0]), int(size[1])
    src = t.detach().to(torch.float32).contiguous()
    h, w = src.shape[-2:]
    dst = src.new_empty(src.shape[:-2] + (H, W))
    planes = src.numel() // (h * w)
    with _on_device(src.device):
        rc = _lib.load().gclm_upsample_fields(src.data_ptr(), planes, h, w, H, W, dst.data_ptr(), _raw_stream(src.device))
    if rc != 0:
        raise _lib.GclmError(f"gclm_upsample_fields failed ({rc})")
    return dst


def upsample_fields_multi(tensors, size):
    """`upsample_fields` for several tensors of equal (h, w) in ONE launch (gclm_upsample_fields_multi): the outputs are views
    of one allocation, each contiguous, with the leading shape of its source."""
    H, W = int(size[0]), int(size[1])
    srcs = []
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("geocalib_amd.upsample_fields_multi needs HIP device tensors (no CPU fallback)")
        srcs.append(t.detach().to(torch.float32).contiguous())
    h, w = srcs[0].shape[-2:]
    assert all(t.shape[-2:] == (h, w) and t.device == srcs[0].device for t in srcs) and 1 <= len(srcs) <= 8
    planes = [t.numel() // (h * w) for t in srcs]
    flat = srcs[0].new_empty((sum(planes), H, W))
    outs, lo = [], 0
    for t, n in zip(srcs, planes):
        outs.append(flat[lo:lo + n].view(t.shape[:-2] + (H, W)))
        lo += n
    C = _lib.C
    n = len(srcs)
    with _on_device(srcs[0].device):
        rc = _lib.load().gclm_upsample_fields_multi((C.c_void_p * n)(*[t.data_ptr() for t in srcs]),
                                                     (C.c_void_p * n)(*[o.data_ptr() for o in outs]), (C.c_int * n)(*planes), n, h, w,
                                                     H, W, _raw_stream(srcs[0].device))
    if rc != 0:
        raise _lib.GclmError(f"gclm_upsample_fields_multi failed ({rc})")
    return outs


def fastest_placement(allocate, solve, tries: int = 3, keep_first: bool = False):
    """Pick, out of `tries` allocations of the same field buffers, the one the sweep streams fastest.

    Where a batch of fields lands in PHYSICAL memory moves the memory-bound sweep by up to 8 % (928 ...
    1 005 us for the same 6.3 GB, exactly reproducible per allocation, a property of the combination of the planes'
    pages that neither the virtual layout nor the library controls) -- only another allocation changes it.  A serving
    loop allocates its field buffers ONCE (the CNN head writes into them every batch) and can afford to choose:

        allocate() -> dict of device tensors (a candidate; all candidates are alive at the same time, hence on
                      different pages);   solve(fields) -> anything (one calibration of the candidate, e.g. an LMOptimizer)

    Every candidate is solved three times (warm-up, then the better of two timed with HIP events on the current stream); returns
    (fields_of_the_fastest, [milliseconds of every candidate]).  The others are dropped.  `keep_first`: a third value,
    the FIRST candidate's fields (what a caller who allocates once gets) -- measurement rigs report both.

    Round 4 tried to control the placement instead of choosing it (HIP virtual-memory management: one physical handle per
    batch / per tensor / per 2 MiB .. 1 GiB chunk, 1 GiB-aligned tensors): same spread, same discrete levels
    (profiles/archive/r04_vmm_placement.log) -- choosing among allocations stays the only handle a caller has."""
    if tries <= 1:
        f = allocate()
        return (f, [], f) if keep_first else (f, [])
    cands, times = [], []
    for _ in range(tries):
        f = allocate()
        solve(f)
        best_ms = float("inf")
        for _ in range(2):            # the better of two timed solves: the candidates differ by 1-4 %, one solve's jitter is ~0.5 %
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            solve(f)
            e1.record()
            e1.synchronize()
            best_ms = min(best_ms, e0.elapsed_time(e1))
        cands.append(f)
        times.append(best_ms)
    best = min(range(tries), key=times.__getitem__)
    return (cands[best], times, cands[0]) if keep_first else (cands[best], times)
