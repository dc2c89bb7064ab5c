"""GPU box: does the physical placement of the inputs change the sweep time?  usage: placement_probe.py <prealloc GiB> [models]
Allocates and frees <prealloc> GiB first (so that the inputs land elsewhere / in differently fragmented memory), then
runs scripts/probes/sweep_probe.py's measurement."""
import sys, os, runpy
import torch
gib = int(sys.argv[1])
if gib > 0:
    x = torch.empty(gib << 30, dtype=torch.uint8, device="cuda"); x.fill_(1); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
sys.argv = ["sweep_probe.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_probe.py"), run_name="__main__")
