"""CPU, world_size 2 (gloo): the multi-process plumbing of geocalib_amd.parallel.

(1) the single all-gather of packed per-image result rows (uneven shards, rank order);
(2) the algebra behind the shared-intrinsics split (BASELINE config 5): per-rank Schur partials summed
    with ONE all-reduce reproduce the reference's dense arrow-head Cholesky step.  The per-frame
    systems come from the oracle, the dense solve is restated from lm_optimizer.py:350-383,:109-137,
    and step 0 is also compared with the reference's recorded delta (tests/golden/golden_trace.npz).
The GPU kernels that produce / consume these partials are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, conf_for, data_for


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(fn, world, *args):
    port = _free_port()
    mp.spawn(_entry, args=(fn, world, port, args), nprocs=world, join=True)


def _entry(rank, fn, world, port, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def _gather_worker(rank, world, n_total):
    from geocalib_amd.parallel import ROW, all_gather_rows, shard_range, unpack_rows
    lo, hi = shard_range(n_total, rank, world)
    rows = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.ones(1, ROW) + torch.arange(ROW) * 1e-3
    full = all_gather_rows(rows, n_total)
    assert full.shape == (n_total, ROW)
    expect = torch.arange(n_total, dtype=torch.float32)[:, None] + torch.arange(ROW) * 1e-3
    assert torch.allclose(full, expect)
    cam, grav, info = unpack_rows(full)
    assert cam.shape[1] == 8 and grav.shape[1] == 3 and info.shape[1] == 48


def _gather_plan_worker(rank, world, n_total):
    """The bench's N>1 loop: exchange buffers allocated once (GatherPlan), the collective timed (CollectiveTimer)."""
    from geocalib_amd.parallel import ROW, CollectiveTimer, GatherPlan, all_gather_rows, shard_range
    plan, timer = GatherPlan(n_total, world, torch.device("cpu")), CollectiveTimer()
    buf_ptr, out_ptr = plan.buf.data_ptr(), plan.out.data_ptr()
    lo, hi = shard_range(n_total, rank, world)
    for it in range(3):
        rows = (torch.arange(lo, hi, dtype=torch.float32)[:, None] + it) * torch.ones(1, ROW)
        full = all_gather_rows(rows, n_total, plan=plan, timer=timer)
        assert torch.equal(full[:, 0], torch.arange(n_total, dtype=torch.float32) + it)
    assert timer.calls == 3 and (plan.buf.data_ptr(), plan.out.data_ptr()) == (buf_ptr, out_ptr)


def test_gather_plan_reuses_its_buffers_world2():
    _run(_gather_plan_worker, 2, 8)
    _run(_gather_plan_worker, 2, 7)


def _early_stop_worker(rank, world):
    """Without an RCCL communicator (gloo rigs, CPU tensors) calibrate_sharded refuses the batch-global early stop -- it
    is only shard-invariant when the per-step counters can be summed over the ranks (gclm_set_stop_comm) -- before
    touching any tensor."""
    import pytest
    from geocalib_amd import LMOptimizer
    from geocalib_amd.parallel import calibrate_sharded
    with pytest.raises(ValueError, match="early_stop=False"):
        calibrate_sharded(LMOptimizer({"camera_model": "pinhole"}), {"latitude_field": torch.zeros(4, 1, 8, 8)}, 8)
    # a shard beyond one C call would be solved in slices, each with its own sequence of stop collectives: ranks with
    # different slice counts would deadlock and the stop would be per slice -- refused up front as well
    big = torch.zeros(1, 1, 2, 2).expand(LMOptimizer._MAX_CALL + 1, 1, 2, 2)             # a view: no memory behind it
    with pytest.raises(ValueError, match="65535 images per rank"):
        calibrate_sharded(LMOptimizer({"camera_model": "pinhole"}), {"latitude_field": big}, 2 * big.shape[0])


def test_calibrate_sharded_refuses_early_stop_world2():
    _run(_early_stop_worker, 2)


def test_shard_ranges_cover_batch():
    from geocalib_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 1024, 8191):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_rows_world8_uneven_8191():
    """The world size the driver's scaling run uses, with configs[2]'s total minus one image (shards of 1024 and 1023
    rows, padded to 1024 for the ONE collective): an N=8-only indexing slip must not be hardware's to find."""
    _run(_gather_worker, 8, 8191)


def test_all_gather_rows_world8_even_8192():
    _run(_gather_worker, 8, 8192)


def test_frame_split_layout_covers_every_frame_once():
    """bench.py's index arithmetic for configs[4] (every rank holds group_size / world frames of EVERY group): over all
    ranks every global frame is produced exactly once, the local group id equals global index // group_size, and the
    per-rank shape of the BASELINE run is 512 groups x 2 frames."""
    import pytest
    from geocalib_amd.parallel import frame_split_layout, global_frame_index
    for gs, world, B in ((16, 8, 1024), (16, 2, 64), (16, 1, 32), (8, 4, 6), (16, 16, 5)):
        seen = []
        for rank in range(world):
            lay = frame_split_layout(B, gs, world, rank)
            assert lay["fpg"] == gs // world and lay["n_groups"] == B // lay["fpg"]
            b = torch.arange(B)
            g = global_frame_index(b, lay)
            assert torch.equal(g // gs, b // lay["fpg"])                 # group_of_frame of bench.py == the global group
            assert torch.equal(g % gs, rank * lay["fpg"] + b % lay["fpg"])
            seen.append(g)
        allg = torch.cat(seen).sort().values
        assert torch.equal(allg, torch.arange(B * world))                # a partition of the global batch
    assert frame_split_layout(1024, 16, 8, 3) == {"fpg": 2, "n_groups": 512, "first_index": 6, "run": 2, "run_stride": 16}
    with pytest.raises(ValueError):
        frame_split_layout(1024, 16, 3, 0)
    with pytest.raises(ValueError):
        frame_split_layout(1023, 16, 8, 0)


def _by_group_worker(rank, world):
    """calibrate_sharded with shared intrinsics says which partition it needs BEFORE any tensor is touched."""
    import pytest
    from geocalib_amd import LMOptimizer
    from geocalib_amd.parallel import calibrate_sharded
    opt = LMOptimizer({"camera_model": "pinhole", "shared_intrinsics": True, "group_size": 4, "early_stop": False})
    local = {"latitude_field": torch.zeros(4, 1, 8, 8)}
    with pytest.raises(ValueError, match="do not divide"):
        calibrate_sharded(opt, local, 12)             # 3 groups over 2 ranks
    with pytest.raises(ValueError, match="whole groups"):
        calibrate_sharded(opt, {"latitude_field": torch.zeros(3, 1, 8, 8)}, 8)


def test_calibrate_sharded_validates_group_partition_world2():
    _run(_by_group_worker, 2)


def test_all_gather_rows_world2_uneven():
    _run(_gather_worker, 2, 5)


def test_all_gather_rows_world2_even():
    _run(_gather_worker, 2, 8)


def _damped(H, lam):
    return H + np.diag(np.maximum(np.diag(H) * lam, 1e-6))


def _dense_arrowhead_delta(Hs, Gs, lam, ni):
    """Reference path: assemble the (2B+ni) system (lm_optimizer.py:350-383), damp, Cholesky (:109-137)."""
    B = Hs.shape[0]
    n = 2 * B + ni
    A, g = np.zeros((n, n)), np.zeros(n)
    for b in range(B):
        A[2 * b:2 * b + 2, 2 * b:2 * b + 2] = Hs[b, :2, :2]
        A[2 * b:2 * b + 2, 2 * B:] = Hs[b, :2, 2:]
        A[2 * B:, 2 * b:2 * b + 2] = Hs[b, 2:, :2]
        A[2 * B:, 2 * B:] += Hs[b, 2:, 2:]
        g[2 * b:2 * b + 2] = Gs[b, :2]
        g[2 * B:] += Gs[b, 2:]
    L = np.linalg.cholesky(_damped(A, lam))
    return np.linalg.solve(L.T, np.linalg.solve(L, g))


def _schur_worker(rank, world, setname, q):
    from oracle import lm_oracle
    from geocalib_amd.parallel import shard_range
    data, conf = data_for(setname, "bench"), conf_for(setname, "bench")
    B = data["latitude_field"].shape[0]
    ni = 1 if "pinhole" in setname else 2
    init = lm_oracle.solve(data, {**conf, "num_steps": 0})           # trivial estimate
    sysm = lm_oracle.system(data, init["camera"], init["gravity"], conf, precision="f64")
    Hs, Gs, lam = sysm["H"], sysm["G"], 0.1
    lo, hi = shard_range(B, rank, world)
    # local Schur partials, layout of gclm_update.hip::shared_step_kernel (stride 32, 3x3 blocks)
    part = np.zeros(32)
    for b in range(lo, hi):
        Dinv = np.linalg.inv(_damped(Hs[b, :2, :2], lam))
        E = Hs[b, :2, 2:]
        S, C = np.zeros((3, 3)), np.zeros((3, 3))
        S[:ni, :ni] = E.T @ Dinv @ E
        C[:ni, :ni] = Hs[b, 2:, 2:]
        part[0:9] += S.ravel()
        part[9:9 + ni] += E.T @ Dinv @ Gs[b, :2]
        part[12:21] += C.ravel()
        part[21:21 + ni] += Gs[b, 2:]
        part[24] += 1
    t = torch.from_numpy(part)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                          # the ONE collective per step
    part = t.numpy()
    assert part[24] == B
    C = part[12:21].reshape(3, 3)[:ni, :ni]
    S = C - part[0:9].reshape(3, 3)[:ni, :ni] + np.diag(np.maximum(np.diag(C) * lam, 1e-6))
    dI = np.linalg.solve(S, part[21:21 + ni] - part[9:9 + ni])
    dG = np.stack([np.linalg.inv(_damped(Hs[b, :2, :2], lam)) @ (Gs[b, :2] - Hs[b, :2, 2:] @ dI) for b in range(lo, hi)])
    dense = _dense_arrowhead_delta(Hs, Gs, lam, ni)
    assert np.allclose(dI, dense[2 * B:], rtol=1e-9, atol=1e-12)
    assert np.allclose(dG.ravel(), dense[2 * lo:2 * hi], rtol=1e-9, atol=1e-12)
    if rank == 0:
        q.put((dI, dense))


def _check_schur(setname):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    _run(_schur_worker, 2, setname, q)
    dI, dense = q.get()
    tr = np.load(os.path.join(GOLDEN, "golden_trace.npz"))
    ref_delta = tr[f"{setname}/delta"][0][0]                          # the reference's own step-0 delta
    assert np.allclose(dense, ref_delta, rtol=2e-3, atol=2e-5)
    assert np.allclose(dI, ref_delta[-len(dI):], rtol=2e-3, atol=2e-5)


def test_schur_split_equals_dense_arrowhead_pinhole():
    _check_schur("shared_pinhole")


def test_schur_split_equals_dense_arrowhead_simple_radial():
    _check_schur("shared_simple_radial")
