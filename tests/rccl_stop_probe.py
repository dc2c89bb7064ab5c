"""GPU box helper of tests/test_gpu_parity.py::test_sharded_early_stop_at_the_baseline_size_through_rccl (not a test).

One rank, GCLM_FORCE_COLLECTIVES=1: `calibrate_sharded(early_stop=True)` at BASELINE configs[1]'s size (B images of
640x480, default 30-step conf) with the batch-global stop travelling through `gclm_set_stop_comm`
  (a) on the direct route (`comm=RcclComm`, results by gclm_comm_all_gather) and
  (b) on the torch route (process group "nccl" = RCCL: the stop communicator is derived from the group with
      RcclComm.from_torch_group, the results travel by torch.distributed.all_gather_into_tensor),
against the plain `LMOptimizer.forward` of the same batch.  With one rank every all-reduce is the identity, so all three
must agree bit for bit, `stop_at` included -- what this exercises is first contact: communicator creation on both
routes, ~stop_at 4-byte all-reduces enqueued between the kernels of one solve, the all-gather at the per-rank shape of
the 8-GPU runs.  Prints ONE JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geocalib_amd import LMOptimizer, _lib  # noqa: E402
from geocalib_amd.parallel import RcclComm, calibrate_sharded  # noqa: E402
from geocalib_amd.synth import synth_fields  # noqa: E402


def main():
    model, B, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    os.environ["GCLM_FORCE_COLLECTIVES"] = "1"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    data, _, _ = synth_fields(model, B, 480, 640, dev, seed=2024)
    conf = {"camera_model": model}                                      # default conf: 30 steps, early_stop=True
    keys = ("camera", "gravity", "final_cost", "stop_at")

    def bits(res):
        return {k: (res[k]._data if k in ("camera", "gravity") else res[k]).detach().cpu() for k in keys}

    plain = bits(LMOptimizer(conf).eval()(data))
    comm = RcclComm(RcclComm.unique_id(), 1, 0, 0)
    direct = bits(calibrate_sharded(LMOptimizer(conf).eval(), data, B, comm=comm))
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
    opt = LMOptimizer(conf).eval()
    routed = bits(calibrate_sharded(opt, data, B))
    torch.cuda.synchronize()
    used_stop_comm = getattr(opt, "_stop_comm", None) is not None
    comp, run = _lib.C.c_int(0), _lib.C.c_int(0)
    _lib.load().gclm_comm_versions(_lib.C.byref(comp), _lib.C.byref(run))
    out = {"model": model, "B": B, "stop_at": {"plain": plain["stop_at"][0].item(), "gclm_comm": direct["stop_at"][0].item(),
                                                "torch_nccl": routed["stop_at"][0].item()},
           "identical_to_plain": {"gclm_comm": all(torch.equal(direct[k], plain[k]) for k in keys),
                                  "torch_nccl": all(torch.equal(routed[k], plain[k]) for k in keys)},
           "torch_route_made_a_stop_communicator": used_stop_comm, "rccl": {"compiled": comp.value, "runtime": run.value}}
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
