"""Multi-GPU calibration: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Independent intrinsics (BASELINE config 3): images are independent, so the batch is sharded by
contiguous image ranges, every rank solves its shard with ZERO communication, and the packed
result rows (camera 8 + gravity 3 + infos 48 floats = 236 B/image) are exchanged with ONE
all-gather.  The payload is KBs: latency-bound, so everything is packed into a single collective.

Shared intrinsics sharded by GROUP (every rank holds whole groups): the same -- no communication during the solve, one
all-gather (`calibrate_sharded` with `group_size`); this is the partition to prefer whenever it is possible.

Shared intrinsics with a group's frames split across ranks (BASELINE config 5): per LM step every
rank reduces its frames to per-group Schur partials (32 floats/group, csrc/gclm_update.hip), ONE
all-reduce(sum) over all groups, then every rank solves the tiny Schur systems redundantly and
updates its own frames.  The reference has no counterpart (its LM is single-process): parity is
checked on the gathered results against the single-process oracle.
"""
import os
from typing import Dict, Tuple

import torch
import torch.distributed as dist

from . import _lib
from .camera import BaseCamera
from .gravity import Gravity
from .lm_optimizer import LMOptimizer, _dev_f32, _unit_gravity, get_trivial_estimation

ROW = 8 + 3 + _lib.INFO_STRIDE   # packed floats per image


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `n` items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frame_split_layout(frames_per_rank: int, group_size: int, world: int, rank: int) -> Dict[str, int]:
    """BASELINE configs[4]'s partition ("512 groups x 16 frames ... across 8 GPUs"): every rank holds
    `group_size // world` frames of EVERY group.  Returns, for `rank`: fpg (frames per group on this rank), n_groups,
    and the strided index runs of the generator / loader -- local frame b is global frame
    first_index + (b // run) * run_stride + b % run, its group is b // fpg (= global index // group_size).
    bench.py and the world-8 CPU test (tests/test_dist.py) share this arithmetic."""
    if group_size <= 0 or world <= 0 or group_size % world:
        raise ValueError(f"group_size {group_size} must be a positive multiple of the number of ranks {world}")
    fpg = group_size // world
    if frames_per_rank % fpg:
        raise ValueError(f"{frames_per_rank} frames per rank do not hold whole runs of {fpg} frames per group")
    return {"fpg": fpg, "n_groups": frames_per_rank // fpg, "first_index": rank * fpg, "run": fpg, "run_stride": group_size}


def global_frame_index(b, layout: Dict[str, int]):
    """Global frame index of local frame(s) `b` under `frame_split_layout` (int or tensor)."""
    return layout["first_index"] + (b // layout["run"]) * layout["run_stride"] + b % layout["run"]


def pack_rows(cam: torch.Tensor, grav: torch.Tensor, info: torch.Tensor) -> torch.Tensor:
    return torch.cat([cam, grav, info], dim=1).contiguous()


def collectives_on(group=None) -> bool:
    """True when results must go through the collectives: a process group with more than one rank -- or with a
    single rank and GCLM_FORCE_COLLECTIVES=1, which lets a one-GPU box exercise the RCCL code path end to end."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("GCLM_FORCE_COLLECTIVES") == "1"


def unpack_rows(rows: torch.Tensor):
    return rows[:, :8], rows[:, 8:11], rows[:, 11:11 + _lib.INFO_STRIDE]


class CollectiveTimer:
    """Device time spent inside the collectives of the N>1 paths: an event pair on the current stream around every
    collective call (torch's nccl collectives run on their own stream, but a blocking call makes the current stream
    wait for them, so the pair brackets the exchange as the solve sees it).  Read with total_ms() after a device
    synchronisation; bench.py reports it as `collective_ms` so that a scaling curve can be attributed."""

    def __init__(self):
        self.pairs, self.calls = [], 0

    def __call__(self, fn, device):
        if device.type != "cuda":
            self.calls += 1
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(device))
        out = fn()
        e1.record(torch.cuda.current_stream(device))
        self.pairs.append((e0, e1))
        self.calls += 1
        return out

    def total_ms(self) -> float:
        ms = sum(a.elapsed_time(b) for a, b in self.pairs)
        self.pairs = []
        return ms


class GatherPlan:
    """Send / receive buffers of the result all-gather, allocated ONCE for a (n_total, world, device): nothing is
    allocated inside a timed loop, and the padded send buffer is zeroed once.  With equal shards the gathered rows ARE
    the receive buffer: results returned by a call that was given a plan alias it until the next call with that plan."""

    def __init__(self, n_total: int, world: int, device, dtype=torch.float32):
        self.n_total, self.world = n_total, world
        self.cap = (n_total + world - 1) // world
        self.buf = torch.zeros((self.cap, ROW), dtype=dtype, device=device)
        self.out = torch.empty((world * self.cap, ROW), dtype=dtype, device=device)
        self.host = None          # gloo test rigs only


def all_gather_rows(rows: torch.Tensor, n_total: int, group=None, plan: GatherPlan = None,
                    timer: CollectiveTimer = None) -> torch.Tensor:
    """ONE all-gather of the per-image result rows of every rank -> (n_total, ROW), rank order.
    Shards may differ by one row: rows are padded to the largest shard for the collective."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if plan is None or plan.n_total != n_total or plan.world != world or plan.buf.device != rows.device:
        plan = GatherPlan(n_total, world, rows.device, rows.dtype)
    cap, buf, out = plan.cap, plan.buf, plan.out
    buf[: rows.shape[0]] = rows

    def exchange():
        if rows.is_cuda and dist.get_backend(group) == "gloo":     # test rigs: gloo has no device collectives
            host = out.cpu()
            dist.all_gather_into_tensor(host, buf.cpu(), group=group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, buf, group=group)

    timer(exchange, rows.device) if timer is not None else exchange()
    if n_total == world * cap:
        return out                                   # equal shards: the receive buffer IS the result
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r * cap: r * cap + (hi - lo)])
    assert parts[rank].shape[0] == rows.shape[0]
    return torch.cat(parts, 0)


class RcclComm:
    """Direct RCCL communicator behind the C ABI (gclm_comm_*): the torch-free route to the two collectives.

    The 128-byte unique id is created on rank 0 and must reach the other ranks out of band; `from_torch_group`
    ships it over an existing torch.distributed group (any backend), `unique_id()` + the constructor let the
    caller use a file or an environment variable instead."""

    def __init__(self, unique_id: bytes, nranks: int, rank: int, device: int):
        assert len(unique_id) == _lib.COMM_ID_BYTES
        self.nranks, self.rank, self.device = nranks, rank, device
        self._ptr = _lib.C.c_void_p()
        buf = _lib.C.create_string_buffer(unique_id, _lib.COMM_ID_BYTES)
        rc = _lib.load().gclm_comm_create(_lib.C.byref(self._ptr), buf, nranks, rank, device)
        if rc != 0:
            raise _lib.GclmError(f"gclm_comm_create failed ({rc}): {_lib.load().gclm_comm_last_error(None).decode()}")

    @staticmethod
    def unique_id() -> bytes:
        buf = _lib.C.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = _lib.load().gclm_comm_unique_id(buf)
        if rc != 0:
            raise _lib.GclmError(f"gclm_comm_unique_id failed ({rc})")
        return buf.raw

    @classmethod
    def from_torch_group(cls, device: int, group=None) -> "RcclComm":
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], world, rank, device)

    def _check(self, rc, what):
        if rc != 0:
            raise _lib.GclmError(f"{what} failed ({rc}): {_lib.load().gclm_comm_last_error(self._ptr).decode()}")

    def all_gather(self, send: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """(n, k) float32 rows of every rank -> (nranks * n, k), rank order; ONE collective on the current stream.
        `out`: a receive buffer kept by the caller (nothing is allocated inside a timed loop)."""
        send = send.contiguous()
        shape = (self.nranks * send.shape[0],) + tuple(send.shape[1:])
        recv = out if out is not None else send.new_empty(shape)
        assert recv.is_contiguous() and tuple(recv.shape) == shape and recv.dtype == send.dtype
        s = torch.cuda.current_stream(send.device).cuda_stream
        self._check(_lib.load().gclm_comm_all_gather(self._ptr, send.data_ptr(), recv.data_ptr(), send.numel(), s),
                    "gclm_comm_all_gather")
        return recv

    def all_reduce_sum_(self, buf: torch.Tensor) -> torch.Tensor:
        assert buf.is_contiguous() and buf.dtype == torch.float32
        s = torch.cuda.current_stream(buf.device).cuda_stream
        self._check(_lib.load().gclm_comm_all_reduce_sum(self._ptr, buf.data_ptr(), buf.numel(), s),
                    "gclm_comm_all_reduce_sum")
        return buf

    def __del__(self):
        try:
            if self._ptr:
                _lib.load().gclm_comm_destroy(self._ptr)
        except Exception:
            pass


def infos_from_rows(opt: LMOptimizer, rows: torch.Tensor, has_up: bool) -> Dict[str, torch.Tensor]:
    cam, grav, info = unpack_rows(rows)
    # the rows hold the solve's own unit vectors: wrapped as they are (re-normalising would move their last bit)
    out = {"camera": opt.camera_model(cam.contiguous()), "gravity": _unit_gravity(grav.contiguous())}
    out.update(opt._unpack_info(info, has_up))
    return out


def calibrate_sharded(opt: LMOptimizer, local_data: Dict[str, torch.Tensor], n_total: int,
                      group=None, comm: "RcclComm" = None, plan: GatherPlan = None,
                      timer: CollectiveTimer = None) -> Dict[str, torch.Tensor]:
    """Solve this rank's shard (independent intrinsics, or whole shared-intrinsics groups) and all-gather everybody's
    results.

    `local_data` holds the fields of the images [shard_range(n_total, rank, world)) of the global
    batch; the returned dict covers all `n_total` images on every rank.  `plan` (GatherPlan) keeps the exchange
    buffers across calls, `timer` (CollectiveTimer) accumulates the device time of the collective."""
    if opt.shared_intrinsics:
        # shared intrinsics shard by GROUP: every rank holds whole groups (frames of a group contiguous), solves them with
        # no communication and joins the single all-gather.  Frames of ONE group spread over ranks: SharedIntrinsicsSplit.
        gs = opt.conf.group_size
        n_local = next(iter(local_data.values())).shape[0]
        world_ = comm.nranks if comm is not None else (dist.get_world_size(group) if collectives_on(group) else 1)
        if not gs or n_local % gs or n_total % gs:
            raise ValueError("calibrate_sharded with shared intrinsics needs `group_size` and whole groups per rank; "
                             "use SharedIntrinsicsSplit when the frames of a group live on several ranks")
        if (n_total // gs) % world_:
            # shard_range would cut a group in two: say so here, not as a shape error inside the gather
            raise ValueError(f"{n_total // gs} groups do not divide over {world_} ranks: give every rank the same number "
                             "of whole groups (or use SharedIntrinsicsSplit)")
    forced = os.environ.get("GCLM_FORCE_COLLECTIVES") == "1"
    multi = comm is not None and (comm.nranks > 1 or forced) or collectives_on(group)
    stop_handle = None
    if not multi:
        return opt(local_data)            # one rank, no forced collective: nothing to pack, nothing to exchange
    if opt.conf.early_stop:
        if next(iter(local_data.values())).shape[0] > opt._MAX_CALL:
            # a shard beyond one C call is solved in slices (LMOptimizer._calibrate_chunked), each with its own sequence of
            # stop all-reduces: ranks with different slice counts would issue mismatched collectives, and the stop would
            # be per slice, not the batch's
            raise ValueError(f"calibrate_sharded with early_stop=True is limited to {opt._MAX_CALL} images per rank "
                             "(one C call, one sequence of stop collectives): use early_stop=False or more ranks")
        # The reference's early stop is ONE decision over the whole batch (lm_optimizer.py:90-92, 619); every rank taking
        # it over its own shard would make the gathered result depend on the world size.  With an
        # RCCL communicator the per-step "cost still moved" counters are summed over the ranks on the solve's stream
        # (gclm_set_stop_comm: one 4-byte all-reduce per LM step), so the stop IS the whole batch's.  Without one (gloo
        # test rigs) only a fixed number of steps is shard-invariant.
        stop_comm = comm
        if stop_comm is None and dist.get_backend(group) == "nccl" and next(iter(local_data.values())).is_cuda:
            stop_comm = getattr(opt, "_stop_comm", None)          # made once per optimiser (the results still travel by torch)
            if stop_comm is None:
                dev_index = next(iter(local_data.values())).device.index
                stop_comm = opt._stop_comm = RcclComm.from_torch_group(
                    dev_index if dev_index is not None else torch.cuda.current_device(), group)
        if stop_comm is None:
            raise ValueError("calibrate_sharded needs early_stop=False (a fixed number of steps) here: the batch-global early "
                             "stop is only shard-invariant with an RCCL communicator (`comm`, or the nccl backend)")
        stop_handle = opt._handle(next(iter(local_data.values())).device)
        _lib.check(_lib.load().gclm_set_stop_comm(stop_handle.ptr, stop_comm._ptr), stop_handle.ptr, "gclm_set_stop_comm")
    try:
        opt(local_data)
    finally:
        if stop_handle is not None:
            _lib.load().gclm_set_stop_comm(stop_handle.ptr, None)
    rows = pack_rows(*opt._last_raw)
    if comm is not None and (comm.nranks > 1 or forced):       # direct RCCL route (gclm_comm_all_gather; equal shards)
        assert n_total % comm.nranks == 0, "the direct RCCL route expects equal shards"
        recv = plan.out if plan is not None and plan.out.shape[0] == comm.nranks * rows.shape[0] else None
        gathered = (timer(lambda: comm.all_gather(rows, recv), rows.device) if timer is not None
                    else comm.all_gather(rows, recv))
        return infos_from_rows(opt, gathered, "up_field" in local_data)
    rows = all_gather_rows(rows, n_total, group, plan, timer)
    return infos_from_rows(opt, rows, "up_field" in local_data)


class SharedIntrinsicsSplit:
    """Shared-intrinsics LM where each group's frames are spread over the ranks of `group`.

    Every rank holds `local_data` (its frames, sorted by group id) and `group_of_frame`
    (int32, non-decreasing, values in [0, num_groups)).  Per step: local sweep + Schur partials
    (gclm_shared_reduce) -> ONE all-reduce(sum) of (num_groups, 32) floats -> solve + update
    (gclm_shared_apply)."""

    def __init__(self, opt: LMOptimizer, num_groups: int, group=None, comm: "RcclComm" = None,
                 timer: CollectiveTimer = None):
        assert opt.shared_intrinsics, "optimizer must be configured with shared_intrinsics=True"
        assert not opt.conf.early_stop, "split shared intrinsics runs a fixed number of steps"
        self.opt, self.num_groups, self.group, self.comm, self.timer = opt, num_groups, group, comm, timer
        self._partials = None       # (num_groups, 32) exchange buffer, kept across calls

    def __call__(self, local_data: Dict[str, torch.Tensor], group_of_frame: torch.Tensor):
        opt, lib = self.opt, _lib.load()
        with torch.no_grad():
            cam0, grav0 = get_trivial_estimation(local_data, opt.camera_model)
            opt.setup_optimization_and_priors(local_data, shared_intrinsics=True)
            up, lat, upc, latc, (B, H, W) = opt._fields(local_data)
            dev = lat.device
            h = opt._handle(dev)
            cam = _dev_f32(cam0._data, "camera").clone()
            grav = _dev_f32(grav0._data, "gravity").clone()
            gof = group_of_frame.to(device=dev, dtype=torch.int32).contiguous()
            if self._partials is None or self._partials.device != dev:
                self._partials = torch.zeros((self.num_groups, _lib.SHARED_PARTIAL_STRIDE), dtype=torch.float32, device=dev)
            partials = self._partials
            info = torch.empty((B, _lib.INFO_STRIDE), dtype=torch.float32, device=dev)
            multi = collectives_on(self.group)

            def exchange():
                if self.comm is not None:
                    self.comm.all_reduce_sum_(partials)
                elif multi and dist.get_backend(self.group) == "gloo":   # test rigs only
                    host = partials.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    partials.copy_(host)
                elif multi:
                    dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=self.group)

            with torch.cuda.device(dev):
                s = torch.cuda.current_stream(dev).cuda_stream
                P = opt._ptr
                _lib.check(lib.gclm_shared_begin(h.ptr, P(up), P(lat), P(upc), P(latc), B, H, W, cam.data_ptr(),
                                                 grav.data_ptr(), gof.data_ptr(), self.num_groups, s), h.ptr,
                           "gclm_shared_begin")
                for step in range(opt.num_steps):
                    _lib.check(lib.gclm_shared_reduce(h.ptr, step, partials.data_ptr(), s), h.ptr, "gclm_shared_reduce")
                    if self.comm is not None or multi:
                        self.timer(exchange, dev) if self.timer is not None else exchange()
                    _lib.check(lib.gclm_shared_apply(h.ptr, step, partials.data_ptr(), s), h.ptr, "gclm_shared_apply")
                _lib.check(lib.gclm_shared_finish(h.ptr, info.data_ptr(), s), h.ptr, "gclm_shared_finish")
        out = {"camera": cam0.__class__(cam), "gravity": Gravity(grav)}
        out.update(opt._unpack_info(info, up is not None))
        return out


__all__ = ["shard_range", "frame_split_layout", "global_frame_index", "pack_rows", "unpack_rows", "all_gather_rows", "calibrate_sharded", "GatherPlan",
           "CollectiveTimer", "SharedIntrinsicsSplit", "RcclComm", "ROW", "BaseCamera"]
