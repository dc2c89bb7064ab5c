"""bench.py -- images/sec of complete LM calibrations on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: a full LM calibration (num_steps=20 sweeps +
final sweep with uncertainty) of `--batch` synthetic 640x480 images PER GPU (weak scaling: the
global batch is batch x N; BASELINE configs[1] at N=1, configs[2] = 8192 images at N=8), followed
for N>1 by the single RCCL all-gather of the packed results.  Inputs are generated on the device
before the timed region (counter-based generator keyed by (seed, global image index)).

    python bench.py                      # N=1
    python bench.py --gpus N             # N>1 without a launcher: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline:     achieved = algorithmic bytes of one sweep launch / mean sweep duration measured with
                HIP events on the launch stream inside the timed region; peak = 8 TB/s HBM3E.
  cpu_baseline: rank 0, at every N, after the last barrier, on a bounded sample of rank 0's timed batch, on THIS box's host
                cores.  Where the reference
                checkout exists on the box ($GEOCALIB_REFERENCE or /root/reference): the reference's OWN PyTorch CPU path
                (geocalib/lm_optimizer.py:141, .eval(), no_grad) is timed -- kind "reference", measured_on "this box" --
                with the CPU oracle (oracle/lm_oracle.c, a port) beside it as `port`.  Where it does not (the GPU box):
                the port is the baseline (kind "port", reference_on_this_box false) and `reference_torch` carries the
                reference's timing from the build container (profiles/cpu_reference_torch.json, hardware named).
  check:        the line's own correctness record.  `vs_oracle`: the HIP results of the first images of rank 0's TIMED batch
                against the CPU oracle's solve of the very same fields (oracle/lm_oracle.c, the restatement of the reference
                that tests/test_oracle.py pins to reference-generated goldens; the solve the cpu_baseline leg times anyway):
                max focal rel / max gravity abs / max final-cost rel distance next to north_star's 1e-4 gate.  The
                ground-truth errors beside it are the noise floor of the synthetic data, not a parity figure.
  secondary:    N=1 default run only: BASELINE configs[3] (simple_radial, B=1024) and configs[4]'s shape (shared intrinsics,
                64 groups x 16 frames) at 5 steps after 2 warm-ups each, same event-based sweep timing, ground-truth check and
                `check.vs_oracle` (64 images each); configs[3] also carries `slat_off`: the same solves on the same allocation
                with the sin(latitude) scratch plane switched off (gclm_set_slat_plane(h, 0)) -- the plane's effect on THIS box;
                and SURVEY 8(f)1's radial and simple_divisional at B=1024, each with `row_pairs_off`: the same solves with the
                one-row walk of the sweep (gclm_set_row_pairs(h, 0)) -- the row-pair walk's effect on THIS box.
  overlap:      N=1, independent intrinsics: the same batch solved as two halves on two side streams
                (LMOptimizer.overlap_streams = 2, the library's default for large batches); `value` stays the one-stream run.
The timed region (exactly --steps steps between barrier + synchronize) is run --repeats times; `value` and
`ms_per_step` are the MEDIAN region (a 0.2-0.4 s window has a few % of run-to-run variance), all regions are listed.
`value`, `ms_per_step` and `roofline` are measured on every rank's FIRST allocation of its input fields; at N = 1
`placement.best_of_n` carries, beside them, the same measurement on the fastest-streaming of --placement-tries allocations
(a serving loop may choose where its persistent field buffers live; the line of record does not).
`roofline.read_ceiling_frac`: what THIS allocation streams on THIS box with the sweep's own load and no arithmetic
(gclm_read_probe over the five timed planes, 5 launches), as a fraction of the 8 TB/s peak; `frac_of_read_ceiling` = frac / that.
For N > 1 the line also carries `multi_gpu.parity`: rank 0 re-generates the shard of ANOTHER rank from (seed, first index),
solves it alone and compares the gathered rows of that rank bit for bit (image / group sharding); for the frame split it solves
whole groups alone and records the distance of the split result (gate 1e-4).
For N > 1 the line also carries `multi_gpu`: ranks_seen (from the communicator), per_rank_ms (every rank's own
median step time) and collective_ms (device time inside the collectives per step, max over ranks); `roofline.per_rank_frac`
lists every rank's own sweep fraction (its own HIP events) and `roofline.frac` is their MINIMUM.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL across processes here
os.environ.setdefault("NCCL_DEBUG", "NONE")                # RCCL otherwise prints a 5-line banner on STDOUT; this file prints ONE line

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
PLANES = 5                       # up_x, up_y, latitude, up_confidence, latitude_confidence


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--camera-model", default="pinhole", choices=["pinhole", "simple_radial", "radial", "simple_divisional"])
    ap.add_argument("--lm-steps", type=int, default=20)
    ap.add_argument("--shared-group", type=int, default=0,
                    help="frames per shared-intrinsics group (BASELINE configs[4]: 16); 0 = independent intrinsics. "
                         "With N GPUs every group's frames are split over the ranks (one all-reduce per LM step)")
    ap.add_argument("--shared-by-group", action="store_true",
                    help="with --shared-group: every GPU owns WHOLE groups (no communication during the solve, one all-gather of "
                         "results) instead of a slice of every group's frames")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--streams", type=int, default=1,
                    help="independent images only: solve the per-GPU batch as this many contiguous parts on side streams "
                         "(LMOptimizer.overlap_streams): one part's update launches run under another part's sweep.  The "
                         "roofline block is then measured in a separate one-stream pass (overlapping launches have no "
                         "separable durations)")
    ap.add_argument("--placement-tries", type=int, default=4,
                    help="N = 1: allocations of the input fields tried for `placement.best_of_n`, AFTER the line of record (which "
                         "is always measured on the first allocation); geocalib_amd.fields.fastest_placement; 1 = none")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="images for the CPU baseline (-1: auto, 0: skip)")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="N=1 pinhole run: skip the `secondary` records (configs[3] simple_radial, configs[4] shared-16 shape)")
    ap.add_argument("--no-overlap", action="store_true", help="N=1: skip the `overlap` record (the batch as two halves on two streams)")
    ap.add_argument("--no-timing", action="store_true", help="skip the in-library HIP-event timing of the sweeps")
    ap.add_argument("--comm", default="torch", choices=["torch", "rccl"],
                    help="route of the two data-path collectives: torch.distributed (nccl = RCCL) or RCCL directly behind "
                         "the C ABI (gclm_comm_all_gather / gclm_comm_all_reduce_sum on the solve's stream)")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="with --shared-group: give every rank the per-rank shape of a run on this many GPUs (e.g. 8: "
                         "2 frames of each of batch/2 groups) -- configs[4]'s partition at shape on fewer GPUs")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (+ ranks sharing GPUs) only to exercise the N>1 path on one GPU")
    return ap.parse_args()


ORACLE_GATE = 1e-4     # north_star: "matching reference focal/gravity to <1e-4 rel"


def vs_oracle(hip, ref, what):
    """Distance of the HIP results from the CPU oracle's on the same images: `hip` / `ref` = dicts of numpy arrays with
    camera (n,8), gravity (n,3), final_cost (n,)."""
    import numpy as np
    n = ref["camera"].shape[0]
    fi = np.abs(hip["camera"][:n, 2:4] / ref["camera"][:, 2:4] - 1).max(1)           # per image
    gi = np.abs(hip["gravity"][:n] - ref["gravity"]).max(1)
    ci = np.abs(hip["final_cost"][:n] / ref["final_cost"] - 1)
    f, g, c = float(fi.max()), float(gi.max()), float(ci.max())
    d = [x for x in (f, g, c)]
    ok = (fi <= ORACLE_GATE) & (gi <= ORACLE_GATE) & (ci <= ORACLE_GATE)             # (a NaN compares false)
    return {"images": int(n), "max_focal_rel": f, "max_gravity_abs": g, "max_final_cost_rel": c, "gate": ORACLE_GATE,
            "within_gate": bool(all(x == x and x <= ORACLE_GATE for x in d)),        # (x == x: a NaN is not within any gate)
            "images_within_gate": int(ok.sum()), "median_focal_rel": float(np.median(fi)), "median_gravity_abs": float(np.median(gi)),
            "median_final_cost_rel": float(np.median(ci)),
            "against": "oracle/lm_oracle.c (float32), " + what}


def oracle_solve(model, lm_steps, data, group, cores, precision="f32"):
    """The checker's answer on `data` (numpy, float32): oracle/lm_oracle.c at float32, independent images, or -- `group` > 0 --
    shared-intrinsics groups of `group` frames, ONE group per call as the reference solves them (lm_optimizer.py:350-383)."""
    import numpy as np
    from oracle import lm_oracle
    conf = {"camera_model": model, "num_steps": lm_steps, "early_stop": False}
    n = next(iter(data.values())).shape[0]
    if group:
        outs = [lm_oracle.solve({k: v[lo:lo + group] for k, v in data.items()}, {**conf, "shared_intrinsics": True}, precision=precision,
                                num_threads=cores) for lo in range(0, n, group)]
    else:
        outs = [lm_oracle.solve(data, conf, precision=precision, num_threads=cores)]
    return {k: np.concatenate([o[k] for o in outs]) for k in ("camera", "gravity", "final_cost")}


def hip_rows(out, n):
    """camera / gravity / final_cost of the first n images of a result dict, on the host"""
    return {"camera": out["camera"]._data[:n].cpu().numpy(), "gravity": out["gravity"]._data[:n].cpu().numpy(),
            "final_cost": out["final_cost"][:n].cpu().numpy()}


def read_ceiling(lib, data, dev, launches=5):
    """What the five planes of `data` stream on this box with the sweep's own load and nothing else (gclm_read_probe): mean
    of `launches` launches after two warm-ups, HIP events on the launch stream.  None when the fields are not the five
    16-byte-aligned planes."""
    if not all(k in data for k in ("up_field", "latitude_field", "up_confidence", "latitude_confidence")):
        return None
    n = data["latitude_field"].numel()
    flat = data["up_field"].reshape(-1)
    planes = [flat[:n], flat[n:], data["latitude_field"].reshape(-1), data["up_confidence"].reshape(-1),
              data["latitude_confidence"].reshape(-1)]
    if n == 0 or n % 4 or any(q.numel() != n or q.data_ptr() % 16 for q in planes):
        return None
    arr = (C.c_void_p * 5)(*[q.data_ptr() for q in planes])
    stream = torch.cuda.current_stream(dev)
    for _ in range(2):
        lib.gclm_read_probe(arr, 5, n, stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(launches):
        if lib.gclm_read_probe(arr, 5, n, stream.cuda_stream) != 0:
            return None
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / launches
    return {"ms": ms, "frac": 5 * n * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "launches": launches}


def cpu_baseline(args, n_images, host_data, keep=None):
    """The CPU baseline on a bounded sample: the FIRST n images of the very batch rank 0's GPU was timed on, already copied
    to the host (numpy, float32).  Runs on rank 0 after the timed regions and the last barrier, at every N (the other
    ranks' processes are idle by then: the sample gets the box's granted cores to itself).

    The port (oracle/lm_oracle.c: a C restatement of the reference algorithm, OpenMP over images) is always timed.  Where
    the reference checkout exists on THIS box ($GEOCALIB_REFERENCE, default /root/reference) the reference's own PyTorch
    path is timed too, on the first `ref_images` of the same sample, and becomes the baseline of record
    (kind "reference", measured_on "this box") with the port beside it.

    --shared-group G: the sample is cut into groups of G frames, each solved as the reference solves a shared-intrinsics
    batch (ONE group per call, lm_optimizer.py:350-383), one call after the other.

    `keep` (a dict): receives the oracle's camera / gravity / final_cost of the sample -- the checker's answer on the timed
    fields, which the caller compares the HIP results with (`check.vs_oracle`)."""
    from oracle import lm_oracle, ref_import
    cores = lm_oracle.effective_cpus()        # what the cgroup grants, not what the box has
    lm_oracle.build()
    group = getattr(args, "shared_group", 0)
    if group:
        n_images = max(group, n_images // group * group)
    data = {k: v[:n_images] for k, v in host_data.items()}
    t0 = time.perf_counter()
    solved = oracle_solve(args.camera_model, args.lm_steps, data, group, cores)
    dt = time.perf_counter() - t0
    if keep is not None:
        keep.update(solved)
    port = {"value": round(n_images / dt, 3), "unit": "images/sec" if not group else "frames/sec",
            "cores": min(cores, group or n_images), "kind": "port",
            "sample": f"the first {n_images} images of rank 0's timed batch ({args.width}x{args.height}), {args.lm_steps} LM iters, "
                      + (f"as {n_images // group} shared-intrinsics groups of {group} frames, one call each, " if group else "")
                      + f"oracle/lm_oracle.c (float32, OpenMP over images), {dt:.1f} s"}
    if ref_import.available():
        try:
            ref = reference_on_this_box(args, data, cores)
            return {**ref, "reference_on_this_box": True, "port": port}
        except Exception as e:      # an incomplete checkout must not take the baseline down: say so and keep the port
            port["reference_error"] = repr(e)
    out = {**port, "reference_on_this_box": False}
    ref = reference_torch(args)
    if ref is not None:
        out["reference_torch"] = ref
    return out


def reference_on_this_box(args, data, cores, ref_images=8):
    """The reference's own LMOptimizer (geocalib/lm_optimizer.py:141; .eval(), torch.no_grad(), CPU float32,
    torch.set_num_threads(cores)) on the first `ref_images` images of the sample, as ONE batch (the reference materialises
    (B,N,2,P) Jacobians: several GB at B = 8 of 640x480), after one untimed single-image warm-up."""
    from oracle import ref_import
    ref = ref_import.load()
    group = getattr(args, "shared_group", 0)
    n = min(group or ref_images, next(iter(data.values())).shape[0])
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        opt = ref.lm_optimizer.LMOptimizer({"camera_model": args.camera_model, "num_steps": args.lm_steps,
                                            "early_stop": False, "shared_intrinsics": bool(group)}).eval()
        td = {k: torch.from_numpy(v[:n].copy()) for k, v in data.items()}
        with torch.no_grad():
            opt({k: v[:1] for k, v in td.items()})             # warm-up: thread pool, allocator
            t0 = time.perf_counter()
            opt(td)
            dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(prev)
    return {"value": round(n / dt, 4), "unit": "images/sec" if not group else "frames/sec", "cores": cores, "kind": "reference",
            "measured_on": "this box",
            "code": f"{ref_import.REFERENCE_ROOT}/geocalib/lm_optimizer.py:141 LMOptimizer, .eval(), torch.no_grad(), CPU float32, "
                    f"torch {torch.__version__}, {cores} threads",
            "sample": f"the first {n} images of rank 0's timed batch ({args.width}x{args.height}) as one "
                      f"{'shared-intrinsics group' if group else 'batch'}, {args.lm_steps} LM iters, {dt:.1f} s"}


def reference_torch(args):
    """The reference's own CPU PyTorch timing for this camera model, as measured by scripts/cpu_reference_torch.py in
    the build container (the only place /root/reference exists); None when the file or the model is missing."""
    path = os.path.join(ROOT, "profiles", "cpu_reference_torch.json")
    if not os.path.exists(path) or (args.height, args.width, args.lm_steps) != (480, 640, 20) or getattr(args, "shared_group", 0):
        return None
    with open(path) as fh:
        r = json.load(fh)
    m = r.get("models", {}).get(args.camera_model)
    if m is None:
        return None
    return {"value": m["images_per_sec"], "unit": "images/sec", "cores": r["cores"], "kind": "reference",
            "code": "geocalib/lm_optimizer.py:141 LMOptimizer, .eval(), torch.no_grad(), CPU float32",
            "measured_on": f"{r['host']}: {r['cpu']}, {r['cores']} threads, torch {r['torch']}",
            "sample": f"{m['images']} images {r['width']}x{r['height']} in chunks of {r['chunk']}, {r['lm_steps']} LM iters, "
                      f"{m['seconds']} s", "source": "profiles/cpu_reference_torch.json"}


def quick_case(lib, LMOptimizer, synth_fields, dev, model, B, H, W, lm_steps, seed, group, steps=5, warmup=2, oracle_images=64,
               slat_control=False, pairs_control=False):
    """One secondary record: `steps` solves after `warmup`, first allocation, one stream; sweep launches timed with the
    library's HIP events, result checked against the synthetic ground truth like the headline and -- the first
    `oracle_images` images -- against the CPU oracle's solve of the same fields.  `slat_control`: the same measurement again
    with the sin(latitude) scratch plane off, then on again (same allocation, same process: A / B / A).  `pairs_control`: the
    same for the row-pair walk of the sweep (LMOptimizer.row_pairs = False = gclm_set_row_pairs(h, 0), then the default again)."""
    conf = {"camera_model": model, "num_steps": lm_steps, "early_stop": False}
    if group:
        conf.update(shared_intrinsics=True, group_size=group)
    opt = LMOptimizer(conf).eval()
    opt.overlap_streams = 1
    data, gt_cam, gt_grav = synth_fields(model, B, H, W, dev, seed=seed, group_size=group or 1)
    h = opt._handle(dev)
    bytes_per_launch = B * H * W * PLANES * 4

    def measure():
        for _ in range(warmup):
            opt(data)
        torch.cuda.synchronize()
        lib.gclm_set_timing(h.ptr, 1)
        t0 = time.perf_counter()
        for _ in range(steps):
            res = opt(data)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n, ms = C.c_int(0), C.c_float(0)
        lib.gclm_last_pass_timing(h.ptr, C.byref(n), C.byref(ms))
        lib.gclm_set_timing(h.ptr, 0)
        avg = ms.value / max(n.value, 1)
        return res, dt, avg, n.value

    out, dt, avg_ms, n_timed = measure()
    f_err = (out["camera"]._data[:, 3] / gt_cam[:, 3] - 1).abs().median().item()
    g_err = (out["gravity"]._data - gt_grav).abs().max(1).values.median().item()
    assert f_err < 5e-3 and g_err < 5e-3, (model, group, f_err, g_err)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    value = B * steps / dt
    rec = {"workload": ((f"BASELINE configs[3]: batch={B}, {model}" if model == "simple_radial" else
                         f"SURVEY 8(f)1: batch={B}, {model}") if not group else
                        f"BASELINE configs[4] shape: shared intrinsics, {B // group} groups x {group} frames, {model}") +
                       f", synthetic {W}x{H}, {lm_steps} LM iters + final/uncertainty sweep, early_stop=False",
           "value": round(value, 1), "unit": "images/sec" if not group else "frames/sec", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 4),
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg_ms, 4), "launches_timed": n_timed,
                        "whole_job_frac": round(value * (lm_steps + 1) * H * W * PLANES * 4 / 1e9 / HBM_PEAK_GBS, 4)},
           "check": {"median_focal_rel_err_vs_gt": f_err, "median_gravity_abs_err_vs_gt": g_err},
           "workspace_bytes": int(lib.gclm_workspace_bytes(h.ptr)), "slat_plane_bytes": int(lib.gclm_slat_plane_bytes(h.ptr)),
           "placement": "first allocation (no choice among allocations)"}
    rc = read_ceiling(lib, data, dev)
    if rc is not None:
        rec["roofline"].update(read_ceiling_frac=round(rc["frac"], 4), frac_of_read_ceiling=round(achieved / HBM_PEAK_GBS / rc["frac"], 4))
    if slat_control:
        on_raw = [t.clone() for t in opt._last_raw]
        lib.gclm_set_slat_plane(h.ptr, 0)
        _, dt0, avg0, n0 = measure()
        same = all(bool(((a == b) | (a.isnan() & b.isnan())).all()) for a, b in zip(on_raw, opt._last_raw))
        lib.gclm_set_slat_plane(h.ptr, -1)
        _, dt2, avg2, _ = measure()
        rec["slat_off"] = {"value": round(B * steps / dt0, 1), "ms_per_step": round(dt0 / steps * 1e3, 4),
                           "frac": round(bytes_per_launch / (avg0 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg0, 4),
                           "launches_timed": n0, "bit_identical": bool(same),
                           "on_again": {"value": round(B * steps / dt2, 1), "ms_per_step": round(dt2 / steps * 1e3, 4),
                                        "frac": round(bytes_per_launch / (avg2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                           "what": "the same solves on the same allocation with gclm_set_slat_plane(h, 0): every sweep evaluates "
                                   "sin(latitude) per pixel instead of loading the library's scratch plane; then the default "
                                   "again (`on_again`).  `value` / `roofline` above are the default (plane on)"}
    if pairs_control:
        on_rows = hip_rows(out, B)
        opt.row_pairs = False
        out0, dt0, avg0, n0 = measure()
        off_rows = hip_rows(out0, B)
        opt.row_pairs = None
        _, dt2, avg2, _ = measure()
        import numpy as np
        d_f = np.abs(on_rows["camera"][:, 2:4] / off_rows["camera"][:, 2:4] - 1).max(1)
        d_g = np.abs(on_rows["gravity"] - off_rows["gravity"]).max(1)
        rec["row_pairs_off"] = {"value": round(B * steps / dt0, 1), "ms_per_step": round(dt0 / steps * 1e3, 4),
                                "frac": round(bytes_per_launch / (avg0 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg0, 4),
                                "launches_timed": n0,
                                "median_focal_rel_vs_default": float(np.median(d_f)), "median_gravity_abs_vs_default": float(np.median(d_g)),
                                "on_again": {"value": round(B * steps / dt2, 1), "ms_per_step": round(dt2 / steps * 1e3, 4),
                                             "frac": round(bytes_per_launch / (avg2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                                "what": "the same solves on the same allocation with gclm_set_row_pairs(h, 0): every lane walks ONE "
                                        "row per iteration and evaluates the radial terms per pixel; then the default again "
                                        "(`on_again`).  `value` / `roofline` above are the default (a lane takes row H - y along "
                                        "with row y and evaluates what depends on r^2 once for both).  Same per-pixel values, "
                                        "another order of additions: the two walks' results differ by float32 summation order, "
                                        "amplified by the solve (medians over the batch beside it)"}
    if oracle_images:
        try:
            from oracle.lm_oracle import effective_cpus
            n = min(B, oracle_images)
            if group:
                n = max(group, n // group * group)
            host = {k: v[:n].cpu().numpy() for k, v in data.items()}
            ref32 = oracle_solve(model, lm_steps, host, group, effective_cpus())
            rec["check"]["vs_oracle"] = vs_oracle(hip_rows(out, n), ref32,
                                                  f"the first {n} images of this case's timed batch" +
                                                  (f" as {n // group} shared-intrinsics groups of {group}" if group else ""))
            if model == "simple_divisional":
                # How sharp is the yardstick on these images?  The model's k column cancels in float32 (camera.py:913): the SAME
                # algorithm evaluated in float64 lands elsewhere on some images.  An image counts as explained when the HIP
                # result is within 1e-4 + 10 x that distance of the float32 oracle (the rule of the seeded fuzz, tests/).
                import numpy as np
                ref64 = oracle_solve(model, lm_steps, host, group, effective_cpus(), precision="f64")
                hip = hip_rows(out, n)
                def per_image(a, b):
                    return np.maximum.reduce([np.abs(a["camera"][:n, 2:4] / b["camera"][:n, 2:4] - 1).max(1),
                                              np.abs(a["gravity"][:n] - b["gravity"][:n]).max(1),
                                              np.abs(a["final_cost"][:n] / b["final_cost"][:n] - 1)])
                own, dist = per_image(ref32, ref64), per_image(hip, ref32)
                beyond = dist > ORACLE_GATE
                rec["check"]["vs_oracle"]["yardstick"] = {
                    "what": "oracle/lm_oracle.c float32 against its own float64 build on the same images (largest of focal rel / "
                            "gravity abs / final-cost rel per image): how far two evaluations of the reference ALGORITHM land apart",
                    "images_where_it_exceeds_gate": int((own > ORACLE_GATE).sum()), "max": float(own.max()), "median": float(np.median(own)),
                    "hip_beyond_gate": [{"image": int(i), "hip_vs_oracle32": float(dist[i]), "oracle32_vs_oracle64": float(own[i])}
                                        for i in np.nonzero(beyond)[0][:8]],
                    "images_within_gate_plus_10x_own": int((dist <= ORACLE_GATE + 10.0 * own).sum())}
        except Exception as e:          # the checker must never take the product measurement down
            rec["check"]["vs_oracle"] = {"error": repr(e)}
    del data, out
    torch.cuda.empty_cache()
    return rec


def rccl_versions(lib):
    """NCCL_VERSION_CODE libgeocalib_hip.so was compiled against, ncclGetVersion() of the librccl this process bound
    (inside a torch process: torch's own librccl.so, loaded first, same soname), and torch's view of the same."""
    comp, run = C.c_int(0), C.c_int(0)
    lib.gclm_comm_versions(C.byref(comp), C.byref(run))
    try:
        tv = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        tv = None
    return {"compiled": comp.value, "runtime": run.value, "torch": tv}


def ensure_built(local_rank: int) -> None:
    """The HIP library normally travels with the tree; on a bare checkout local rank 0 builds it (hipcc), the other
    ranks wait for the file.  There is no fallback: without the library the bench fails."""
    so = os.path.join(ROOT, "geocalib_amd", "lib", "libgeocalib_hip.so")
    if os.path.exists(so):
        return
    if local_rank == 0:
        import subprocess                  # in a child, stdout -> stderr: this process prints ONE JSON line
        subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, stdout=sys.stderr,
                       check=True)
        return
    deadline = time.time() + 900
    while not os.path.exists(so):
        if time.time() > deadline:
            raise SystemExit(f"{so} was not built")
        time.sleep(1.0)
    time.sleep(2.0)      # let the linker finish writing


_RESULT_FD = None


def claim_stdout():
    """stdout carries ONE line.  Libraries below us write there too (gloo: "[Gloo] Rank 0 is connected to ...", RCCL's
    banner), so file descriptor 1 is pointed at stderr for the life of the process and the JSON line goes to a private
    duplicate of the original stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj) -> None:
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def fail_line(args, message: str, code: int = 2):
    """ONE JSON line that says why there is no measurement (instead of a traceback), and a non-zero exit."""
    emit({"metric": "images/sec LM calibration (640x480, 20 iters)", "value": None, "unit": "images/sec",
          "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": message})
    raise SystemExit(code)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) under
    torch.distributed.run on 127.0.0.1 with a free port, same arguments.  The children inherit stdout, rank 0 prints
    the ONE JSON line; the launcher's own chatter goes to stderr."""
    import socket
    import subprocess
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible == 0:
        fail_line(args, "no HIP device is visible (bench.py has no CPU fallback for the product path)")
    if args.backend == "nccl" and visible < args.gpus:        # (gloo test rigs let ranks share a GPU)
        fail_line(args, f"--gpus {args.gpus} but only {visible} HIP device(s) visible: RCCL needs one GPU per rank")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, GCLM_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "1")               # (what torchrun would set, without its warning)
    # the children write their ONE line to our stdout (the private duplicate); whatever else they print goes to stderr
    raise SystemExit(subprocess.run(cmd, env=env, stdout=_RESULT_FD if _RESULT_FD is not None else None).returncode)


def main():
    args = parse()
    claim_stdout()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("GCLM_FORCE_COLLECTIVES") == "1"):
        self_launch(args)       # (GCLM_FORCE_COLLECTIVES=1 with --gpus 1: the N>1 code path through RCCL with one rank)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); measuring {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        if rank == 0:
            fail_line(args, "no HIP device is visible (bench.py has no CPU fallback for the product path)")
        raise SystemExit(2)
    local_dev = local_rank % torch.cuda.device_count()      # one rank per GPU; test rigs with fewer GPUs than ranks share them
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # GCLM_FORCE_COLLECTIVES=1 under torchrun with ONE rank: every collective of the N>1 path runs (through RCCL)
    distributed = world > 1 or (os.environ.get("GCLM_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    ensure_built(local_rank)
    from geocalib_amd import LMOptimizer, _lib
    from geocalib_amd.parallel import (CollectiveTimer, GatherPlan, RcclComm, SharedIntrinsicsSplit, calibrate_sharded,
                                       frame_split_layout)
    from geocalib_amd.fields import fastest_placement
    from geocalib_amd.synth import synth_fields

    lib = _lib.load()
    B, H, W = args.batch, args.height, args.width
    n_total = B * world
    gs = args.shared_group
    conf = {"camera_model": args.camera_model, "num_steps": args.lm_steps, "early_stop": False}
    comm = None
    if args.comm == "rccl":
        assert distributed and args.backend == "nccl", "--comm rccl needs a process group to hand out the unique id"
        comm = RcclComm.from_torch_group(local_dev)
    vworld = args.virtual_world or world
    placement = {"tries": max(args.placement_tries, 1), "solve_ms": None}
    makers = []                       # how this rank's fields are made (N = 1: more allocations are tried AFTER the line of record)

    def place(make, optimizer):
        """The input fields of this rank: the FIRST allocation -- what a caller who allocates once gets.  `value`,
        `ms_per_step` and `roofline` are measured on it, at every N (no choice among allocations enters the line of record)."""
        makers.append(make)
        return make()

    if gs == 0:
        # independent intrinsics: rank r owns the contiguous images [r*B, (r+1)*B)
        opt = LMOptimizer(conf).eval()
        opt.overlap_streams = 1            # the line of record is the one-stream solve (--streams / the `overlap` block: two)
        data, gt_cam, gt_grav = place(lambda: synth_fields(args.camera_model, B, H, W, dev, seed=args.seed, first_index=rank * B), opt)
        ctimer = CollectiveTimer()
        plan = GatherPlan(n_total, world, dev) if distributed else None      # exchange buffers live outside the timed loop

        def step():
            return calibrate_sharded(opt, data, n_total, comm=comm, plan=plan, timer=ctimer)
    elif args.shared_by_group:
        # shared intrinsics sharded by group: rank r owns the groups of the frames [r*B, (r+1)*B); no collective in the solve
        assert B % gs == 0, "the per-GPU batch must hold whole groups"
        opt = LMOptimizer({**conf, "shared_intrinsics": True, "group_size": gs}).eval()
        data, gt_cam, gt_grav = place(lambda: synth_fields(args.camera_model, B, H, W, dev, seed=args.seed, first_index=rank * B,
                                                           group_size=gs), opt)
        ctimer = CollectiveTimer()
        plan = GatherPlan(n_total, world, dev) if distributed else None

        def step():
            return calibrate_sharded(opt, data, n_total, comm=comm, plan=plan, timer=ctimer)
    else:
        # shared intrinsics: every rank holds gs/world frames of EVERY group (--virtual-world: the shape of a larger run)
        lay = frame_split_layout(B, gs, vworld, rank)
        fpg, n_groups = lay["fpg"], lay["n_groups"]
        # one handle solves what it holds: groups of fpg local frames (= the whole group when it is not split)
        opt = LMOptimizer({**conf, "shared_intrinsics": True, "group_size": fpg}).eval()
        data, gt_cam, gt_grav = place(lambda: synth_fields(args.camera_model, B, H, W, dev, seed=args.seed, first_index=lay["first_index"],
                                                           group_size=gs, run=lay["run"], run_stride=lay["run_stride"]), opt)
        ctimer = CollectiveTimer()
        if not distributed:
            def step():
                return opt(data)
        else:
            gof = torch.arange(B, device=dev, dtype=torch.int32) // fpg
            split = SharedIntrinsicsSplit(opt, n_groups, comm=comm, timer=ctimer)

            def step():
                return split(data, gof)

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    ctimer.total_ms()                                     # drop the warm-up's collective events
    handle = opt._handle(dev)
    overlapped = args.streams > 1 and gs == 0
    presweep = None
    if overlapped and not args.no_timing:
        # the sweep's launch duration, on ONE stream, over --steps untimed steps: the timed steps below overlap the parts
        lib.gclm_set_timing(handle.ptr, 1)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        n, ms = C.c_int(0), C.c_float(0)
        _lib.check(lib.gclm_last_pass_timing(handle.ptr, C.byref(n), C.byref(ms)), handle.ptr, "timing")
        presweep = (ms.value, n.value)
        lib.gclm_set_timing(handle.ptr, 0)
    if overlapped:
        opt.overlap_streams = args.streams
        for _ in range(2):
            out = step()
        torch.cuda.synchronize()
    elif not args.no_timing:
        lib.gclm_set_timing(handle.ptr, 1)

    regions = []                                          # seconds per timed region of exactly --steps steps
    for _ in range(max(args.repeats, 1)):
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        regions.append(time.perf_counter() - t0)
    own = sorted(regions)[len(regions) // 2]              # this rank's median region
    coll_ms = ctimer.total_ms() / (len(regions) * args.steps)
    sweep_ms, sweep_n = 0.0, 0
    if presweep is not None:
        sweep_ms, sweep_n = presweep
    elif not args.no_timing and not overlapped:   # HIP events recorded around every sweep launch of the timed regions
        n, ms = C.c_int(0), C.c_float(0)
        _lib.check(lib.gclm_last_pass_timing(handle.ptr, C.byref(n), C.byref(ms)), handle.ptr, "timing")
        sweep_ms, sweep_n = ms.value, n.value
    algo_bytes_per_launch = B * H * W * PLANES * 4
    if not args.no_timing:
        lib.gclm_set_timing(handle.ptr, 0)
    # the pure-read ceiling of THIS rank's allocation, right after the timed regions (same buffers, same clocks)
    own_rc = read_ceiling(lib, data, dev)
    own_rc_frac = own_rc["frac"] if own_rc is not None else 0.0

    def algo_frac(avg_launch_ms):
        """algorithmic bytes of one sweep launch / its duration, as a fraction of the HBM peak"""
        return algo_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    own_sweep_ms = sweep_ms / sweep_n if sweep_n else 0.0
    ranks_seen, per_rank_ms, coll_ms_max, per_rank_sweep_ms = world, [own / args.steps * 1e3], coll_ms, [own_sweep_ms]
    per_rank_rc = [own_rc_frac]
    if distributed:
        tdev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor(regions, device=tdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)           # every region: MAX over ranks, then the median region
        regions = t.tolist()
        ranks_seen = dist.get_world_size()
        mine = torch.tensor([own / args.steps * 1e3, coll_ms, own_sweep_ms, own_rc_frac], device=tdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(allr, mine)
        per_rank_ms = [round(x[0].item(), 4) for x in allr]
        coll_ms_max = max(x[1].item() for x in allr)
        per_rank_sweep_ms = [x[2].item() for x in allr]     # every rank's own mean sweep launch (its own HIP events)
        per_rank_rc = [x[3].item() for x in allr]           # ... and its own allocation's pure-read ceiling
    elapsed = sorted(regions)[len(regions) // 2]

    # sanity: the solve must have recovered the synthetic ground truth (a fast wrong answer is no answer)
    cam = out["camera"]._data
    lo = rank * B if cam.shape[0] == n_total else 0
    f_err = (cam[lo:lo + B, 3] / gt_cam[:, 3] - 1).abs().median().item()
    g_err = (out["gravity"]._data[lo:lo + B] - gt_grav).abs().max(1).values.median().item()
    if os.environ.get("GCLM_BENCH_NO_CHECK") != "1":      # only the -DGCLM_NOMATH=1 measurement build (memory ceiling) skips this
        assert f_err < 5e-3 and g_err < 5e-3, (f_err, g_err)

    def timed_regions(fn, repeats):
        """median seconds of `repeats` regions of --steps calls of fn (single process: no barrier)"""
        rs = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            rs.append(time.perf_counter() - t0)
        return sorted(rs)[len(rs) // 2]

    extras = {}
    solo = world == 1 and not distributed and gs == 0 and not overlapped
    if solo:
        if placement["tries"] > 1:
            # beside the line of record: what a serving loop that CHOOSES among allocations of its persistent field buffers
            # gets on this box (geocalib_amd.fields.fastest_placement) -- the first allocation is candidate 0
            free_bytes = torch.cuda.mem_get_info(dev)[0]
            tries = max(1, min(placement["tries"], 1 + int(0.8 * free_bytes // (B * H * W * PLANES * 4))))   # all alive at once
            pending = [data]
            chosen, ms, _ = fastest_placement(lambda: pending.pop() if pending else makers[0]()[0], opt, tries, keep_first=True)
            placement.update(tries=tries, solve_ms=[round(t, 3) for t in ms] or None, chosen=ms.index(min(ms)) if ms else 0)
            best = {"chosen": placement["chosen"], "value": round(B * args.steps / elapsed, 1),
                    "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                    "frac": round(algo_frac(sweep_ms / sweep_n), 4) if sweep_n else None}
            if placement["chosen"] != 0:
                if not args.no_timing:
                    lib.gclm_set_timing(handle.ptr, 1)
                opt(chosen)
                sec = timed_regions(lambda: opt(chosen), max(args.repeats, 1))
                best.update(value=round(B * args.steps / sec, 1), ms_per_step=round(sec / args.steps * 1e3, 4))
                if not args.no_timing:
                    n, ms_ = C.c_int(0), C.c_float(0)
                    _lib.check(lib.gclm_last_pass_timing(handle.ptr, C.byref(n), C.byref(ms_)), handle.ptr, "timing")
                    lib.gclm_set_timing(handle.ptr, 0)
                    best["frac"] = round(algo_frac(ms_.value / max(n.value, 1)), 4)
            placement["best_of_n"] = best
            del chosen
        if not args.no_overlap and B >= 512:
            # the library's default for a batch this large: two halves on two side streams, one part's update launches
            # under the other's sweep (LMOptimizer.overlap_streams); results must be the one-stream solve's, bit for bit
            one = [t.clone() for t in opt._last_raw]
            opt.overlap_streams = 2
            for _ in range(2):
                opt(data)
            same = all(bool(((a == b) | (a.isnan() & b.isnan())).all()) for a, b in zip(one, opt._last_raw))   # (NaN == NaN here)
            sec = timed_regions(lambda: opt(data), max(args.repeats, 1))
            opt.overlap_streams = 1
            v2 = B * args.steps / sec
            extras["overlap"] = {"streams": 2, "value": round(v2, 1), "ms_per_step": round(sec / args.steps * 1e3, 4),
                                 "bit_identical": bool(same),
                                 "whole_job_frac": round(v2 * (args.lm_steps + 1) * H * W * PLANES * 4 / 1e9 / HBM_PEAK_GBS, 4)}
    host_sample = hip_sample = None
    if rank == 0 and args.cpu_sample != 0:
        from oracle.lm_oracle import effective_cpus
        n_cpu = min(B, args.cpu_sample if args.cpu_sample > 0 else max(8, 16 * effective_cpus()))   # ~7-10 s of CPU work
        if gs:
            n_cpu = min(B, max(gs, n_cpu // gs * gs))      # whole groups of the sample (any gs frames: the work is the same)
        host_sample = {k: v[:n_cpu].cpu().numpy() for k, v in data.items()}
        # rank 0's rows come first in a gathered result; the oracle's solve of the sample is the same sub-problem as the HIP
        # solve wherever rank 0 holds whole groups (always for independent images; not for a frame split / --virtual-world)
        whole_groups = gs == 0 or args.shared_by_group or (not distributed and vworld == 1)
        hip_sample = hip_rows(out, n_cpu) if whole_groups else None
    if solo and args.secondary and args.camera_model == "pinhole":
        del data, out
        torch.cuda.empty_cache()
        extras["secondary"] = {
            f"simple_radial_B{B}": quick_case(lib, LMOptimizer, synth_fields, dev, "simple_radial", B, H, W, args.lm_steps, args.seed, 0,
                                              oracle_images=64 if args.cpu_sample != 0 else 0, slat_control=True),
            "shared16_pinhole": quick_case(lib, LMOptimizer, synth_fields, dev, "pinhole", B, H, W, args.lm_steps, args.seed, 16,
                                           oracle_images=64 if args.cpu_sample != 0 else 0),
            f"radial_B{B}": quick_case(lib, LMOptimizer, synth_fields, dev, "radial", B, H, W, args.lm_steps, args.seed, 0,
                                       oracle_images=64 if args.cpu_sample != 0 else 0, pairs_control=True),
            f"simple_divisional_B{B}": quick_case(lib, LMOptimizer, synth_fields, dev, "simple_divisional", B, H, W, args.lm_steps, args.seed, 0,
                                                  oracle_images=64 if args.cpu_sample != 0 else 0, pairs_control=True),
        }

    def same_bits(a, b):
        return bool(((a == b) | (a.isnan() & b.isnan())).all())

    def gathered_parity():
        """Image / group sharding: rank 0 re-generates the shard of ANOTHER rank from (seed, first index) -- the generator is
        index-keyed -- solves it alone and compares the rows the all-gather delivered for that rank, bit for bit: proves that
        the collective returned every rank's rows in rank order and unscrambled (the gathered rows of rank r ARE a one-GPU
        solve of the images [rB, (r+1)B) by construction)."""
        r = world - 1
        d2, _, _ = synth_fields(args.camera_model, B, H, W, dev, seed=args.seed, first_index=r * B, group_size=gs or 1)
        alone = opt(d2)
        torch.cuda.synchronize()
        keys = [k for k in alone if torch.is_tensor(alone[k]) or hasattr(alone[k], "_data")]
        diff = []
        for k in keys:
            a = alone[k]._data if hasattr(alone[k], "_data") else alone[k]
            g = out[k]._data if hasattr(out[k], "_data") else out[k]
            if not same_bits(a, g[r * B:(r + 1) * B]):
                diff.append(k)
        return {"kind": "gathered rows of one rank vs a one-GPU solve of the same images on rank 0", "rank_checked": r,
                "images": B, "keys_compared": len(keys), "bit_identical": not diff, "keys_differing": diff}

    def split_parity():
        """Frame split (one all-reduce per LM step): rank 0 solves up to 4 WHOLE groups alone (shared intrinsics, one launch
        sequence, no collective) and records how far its own frames of those groups came out in the split run -- the first
        observation of the collective's summation order (expected ~1e-6, gate 1e-4)."""
        if vworld != world:
            return {"kind": "frame split vs whole groups solved on rank 0", "skipped": f"--virtual-world {vworld}: the {world} rank(s) "
                    "of this run hold only part of every group, the sum over them is not the whole group's system"}
        ng = min(4, n_groups)
        d2, _, _ = synth_fields(args.camera_model, ng * gs, H, W, dev, seed=args.seed, first_index=0, group_size=gs)
        opt2 = LMOptimizer({**conf, "shared_intrinsics": True, "group_size": gs}).eval()
        alone = opt2(d2)
        torch.cuda.synchronize()
        loc = torch.arange(ng * fpg, device=dev)
        idx = (loc // fpg) * gs + lay["first_index"] + loc % fpg           # this rank's frames inside the whole groups
        f = (out["camera"]._data[:ng * fpg, 2:4] / alone["camera"]._data[idx, 2:4] - 1).abs().max().item()
        g = (out["gravity"]._data[:ng * fpg] - alone["gravity"]._data[idx]).abs().max().item()
        c = (out["final_cost"][:ng * fpg] / alone["final_cost"][idx] - 1).abs().max().item()
        return {"kind": "frame split vs whole groups solved on rank 0", "groups_checked": ng, "frames_compared": ng * fpg,
                "max_focal_rel": f, "max_gravity_abs": g, "max_final_cost_rel": c, "gate": ORACLE_GATE,
                "within_gate": bool(all(x == x and x <= ORACLE_GATE for x in (f, g, c)))}

    if "best_of_n" not in placement:
        placement["tries"] = 1            # (N > 1, shared intrinsics, --streams: the first allocation and nothing else)
    if rank == 0:
        value = n_total * args.steps / elapsed
        result = {
            "metric": "images/sec LM calibration (640x480, 20 iters)", "value": round(value, 1),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "repeats": len(regions), "ms_per_step_repeats": [round(r / args.steps * 1e3, 4) for r in regions],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{1 if world == 1 else 2}]: batch={B}/GPU ({n_total} total) "
                                    if gs == 0 else
                                    f"BASELINE configs[4] shape: shared intrinsics, {n_total // gs} groups x {gs} frames "
                                    f"({B} frames/GPU), " if vworld == world or args.shared_by_group else
                                    f"BASELINE configs[4] per-rank shape of a {vworld}-GPU run: {B * vworld // gs} groups x {gs} frames, "
                                    f"this rank holds {gs // vworld} frames of each ({B} frames/GPU), ") +
                                   f"synthetic {W}x{H} perspective fields, {args.camera_model}, "
                                   f"{args.lm_steps} LM iters + final/uncertainty sweep, early_stop=False",
                       "shared_group": gs,
                       "camera_model": args.camera_model, "global_batch": n_total, "per_gpu_batch": B,
                       "height": H, "width": W, "lm_steps": args.lm_steps, "planes": PLANES,
                       "streams": args.streams if overlapped else 1,
                       "parallelism": ("single GPU" if world == 1 else
                                       f"image-sharded x{world}, one all-gather of results" if gs == 0 else
                                       f"group-sharded x{world}, one all-gather of results" if args.shared_by_group else
                                       f"frames of every group split x{world}, one all-reduce per LM step")},
            "check": {"median_focal_rel_err_vs_gt": f_err, "median_gravity_abs_err_vs_gt": g_err},
            "placement": {**placement, "what": "`value`, `ms_per_step` and `roofline` are measured on every rank's FIRST allocation of "
                          "its input fields (no choice among allocations).  N = 1 only, after the line of record: `tries` - 1 more "
                          "allocations are made and `best_of_n` is the same measurement on the allocation whose solve ran fastest "
                          "(geocalib_amd.fields.fastest_placement, DESIGN.md 9.1; candidate 0 = the first allocation)"},
        }
        result.update(extras)
        if distributed:
            result["multi_gpu"] = {
                "ranks_seen": ranks_seen, "per_rank_ms": per_rank_ms,
                "collective": ("all_gather of packed result rows" if gs == 0 or args.shared_by_group
                               else "all_reduce(sum) of per-group Schur partials"),
                "collectives_per_step": 1 if gs == 0 or args.shared_by_group else args.lm_steps,
                "collective_bytes": (n_total * 4 * (8 + 3 + _lib.INFO_STRIDE) if gs == 0 or args.shared_by_group
                                     else n_groups * 4 * _lib.SHARED_PARTIAL_STRIDE),
                "collective_ms": round(coll_ms_max, 4), "backend": args.backend,
                "comm": ("gclm_comm_* (RCCL behind the C ABI, on the solve's stream)" if comm is not None
                         else "torch.distributed"),
                "virtual_world": vworld if vworld != world else None,
                "launched_by": "bench.py itself (torch.distributed.run child)" if os.environ.get("GCLM_BENCH_SELF_LAUNCHED") == "1"
                               else "an external launcher",
                "rccl": rccl_versions(lib)}
            try:
                result["multi_gpu"]["parity"] = gathered_parity() if gs == 0 or args.shared_by_group else split_parity()
            except Exception as e:      # (a failing check is reported, it does not take the measurement down)
                result["multi_gpu"]["parity"] = {"error": repr(e)}
        if sweep_n and all(t > 0 for t in per_rank_sweep_ms):
            # every rank's own mean sweep launch; the line's `frac` / `achieved` are the SLOWEST rank's (N = 1: the only one)
            per_rank_frac = [round(algo_frac(t), 4) for t in per_rank_sweep_ms]
            avg_ms = max(per_rank_sweep_ms)
            achieved = algo_bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic, tsrc = None, ""
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as fh:
                    t = json.load(fh)
                key = f"{args.camera_model}_B{B}_{W}x{H}"
                traffic = t.get(key, {}).get("hbm_bytes_per_launch")
                tsrc = t.get(key, {}).get("source", "")
            result["roofline"] = {
                "bound": "hbm", "kernel": "gclm::sweep_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": (f"profiles/pmc_traffic.json ({tsrc}): rocprofv3 --pmc passes of this command on ANOTHER "
                                   "box and allocation than this run's, committed; not re-measured here (PMC needs rocprofv3 "
                                   "around the process)") if traffic is not None else None,
                "algorithmic_bytes_per_launch": algo_bytes_per_launch, "avg_launch_ms": round(avg_ms, 4),
                "launches_timed": sweep_n, "per_rank_frac": per_rank_frac,
                "per_rank_avg_launch_ms": [round(t, 4) for t in per_rank_sweep_ms],
                "measured_in": (f"a separate pass of {args.steps} steps on ONE stream before the timed regions: the timed steps "
                                f"solve {args.streams} parts of the batch concurrently, whose launch durations overlap and are "
                                "not separable" if overlapped else "the timed regions"),
                "whole_job_frac": round(value / world * (args.lm_steps + 1) * H * W * PLANES * 4 / 1e9 / HBM_PEAK_GBS, 4)}
            if all(x > 0 for x in per_rank_rc):
                # what the SAME buffers stream with the sweep's load and no arithmetic (gclm_read_probe, 5 launches right
                # after the timed regions): how much of `frac` is the allocation / box, how much the kernel
                result["roofline"].update(
                    read_ceiling_frac=round(per_rank_rc[per_rank_sweep_ms.index(avg_ms)], 4),
                    frac_of_read_ceiling=round(achieved / HBM_PEAK_GBS / per_rank_rc[per_rank_sweep_ms.index(avg_ms)], 4),
                    per_rank_read_ceiling_frac=[round(x, 4) for x in per_rank_rc],
                    read_ceiling="gclm_read_probe over this rank's five timed planes: non-temporal 16 B / lane loads, four per "
                                 "plane in flight per thread, no arithmetic; mean of 5 launches after the timed regions")
        if host_sample is not None:      # (the baselines time independent solves: configs[1] / [3])
            kept = {}
            try:
                result["cpu_baseline"] = cpu_baseline(args, next(iter(host_sample.values())).shape[0], host_sample, keep=kept)
            except Exception as e:  # the checker must never take the product measurement down
                result["cpu_baseline"] = {"value": None, "error": repr(e)}
            if hip_sample is not None and kept:
                n_ref = kept["camera"].shape[0]
                result["check"]["vs_oracle"] = vs_oracle(
                    hip_sample, kept, f"the first {n_ref} images of rank 0's timed batch"
                    + (f" as {n_ref // gs} shared-intrinsics groups of {gs}" if gs else "") + " (the cpu_baseline sample)")
            elif hip_sample is None:
                result["check"]["vs_oracle"] = None       # (a frame split: rank 0's frames are no sub-problem; see multi_gpu.parity)
        emit(result)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
