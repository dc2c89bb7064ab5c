"""Time the REFERENCE's own CPU PyTorch path (build container only: needs /root/reference).

    python scripts/cpu_reference_torch.py [--images 16] [--chunk 8]

What is timed is upstream code, untouched: `geocalib.lm_optimizer.LMOptimizer` (lm_optimizer.py:141) in `.eval()`
mode under `torch.no_grad()`, float32 on the CPU, `torch.set_num_threads(os.cpu_count())`, num_steps = 20,
early_stop = False, on synthetic 640x480 perspective fields of the bench protocol (oracle/synth.py), in chunks of
B = 8 images (the reference materialises (B,N,2,P) Jacobians and many (B,N,2,2) temporaries: several GB at B = 8;
B = 1024 does not fit).  One untimed warm-up chunk, then `--images` images.

The result goes to profiles/cpu_reference_torch.json; bench.py attaches it to its JSON line as
`cpu_baseline.reference_torch` WITH ITS PROVENANCE (the hardware it was measured on is this container's host CPU, not
the GPU box's: /root/reference does not exist there).  It is a reported baseline, not a target."""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import, synth  # noqa: E402


def cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_reference_torch.json"))
    args = ap.parse_args()
    ref = ref_import.load()
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    H, W, steps = 480, 640, 20
    res = {"what": "geocalib.lm_optimizer.LMOptimizer (reference, /root/reference, unmodified), .eval(), torch.no_grad(), "
                   "CPU float32",
           "host": "build container (NOT the GPU box: the reference is not available there)",
           "cpu": cpu_name(), "cores": cores, "torch_threads": torch.get_num_threads(), "torch": torch.__version__,
           "height": H, "width": W, "lm_steps": steps, "early_stop": False, "chunk": args.chunk, "models": {}}
    for model in ("pinhole", "simple_radial"):
        opt = ref.lm_optimizer.LMOptimizer({"camera_model": model, "num_steps": steps, "early_stop": False}).eval()

        def run(indices):
            data, cams, _ = synth.make_fields(args.seed, indices, model, H, W)
            td = {k: torch.from_numpy(v) for k, v in data.items()}
            t0 = time.perf_counter()
            with torch.no_grad():
                out = opt(td)
            dt = time.perf_counter() - t0
            err = float(np.median(np.abs(out["camera"]._data[:, 3].numpy() / cams[:, 3] - 1)))
            return dt, err

        run(range(args.chunk))                     # warm-up (allocator, thread pool)
        total, errs = 0.0, []
        for lo in range(0, args.images, args.chunk):
            dt, err = run(range(1000 + lo, 1000 + min(lo + args.chunk, args.images)))
            total += dt
            errs.append(err)
        res["models"][model] = {"images_per_sec": round(args.images / total, 4), "images": args.images,
                                "seconds": round(total, 2), "median_focal_rel_err_vs_gt": max(errs)}
        print(model, res["models"][model], flush=True)
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
