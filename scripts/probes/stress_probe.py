import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from geocalib_amd import LMOptimizer
LMOptimizer.overlap_streams = 1      # these probes time single launches (the library default would split a large batch over two streams)
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
def run(model, B, H, W, steps=20):
    d, gtc, gtg = synth_fields(model, B, H, W, dev, seed=3)
    opt = LMOptimizer({"camera_model": model, "num_steps": steps, "early_stop": False}).eval()
    out = opt(d); torch.cuda.synchronize()
    t = time.perf_counter(); out = opt(d); torch.cuda.synchronize(); dt = time.perf_counter() - t
    f = (out["camera"]._data[:, 3] / gtc[:, 3] - 1).abs()
    print(f"{model} B={B} {W}x{H}: {dt*1e3:.2f} ms, {B/dt:.0f} img/s, {B*H*W*20*(steps+1)/dt/1e12:.2f} TB/s; median f err {f.median().item():.1e}, max {f.max().item():.1e}, fails {out['step_failures'].sum().item()}", flush=True)
run("pinhole", 8192, 480, 640)
run("pinhole", 8, 2048, 2048)
run("simple_radial", 2, 4096, 4096)
run("pinhole", 65535, 16, 16)
try:
    run("pinhole", 65536, 16, 16)
except Exception as e:
    print("B=65536:", type(e).__name__, str(e)[:120])
run("pinhole", 3, 480, 642)   # W % 4 != 0 -> scalar path
