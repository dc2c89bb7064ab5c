"""Probe (GPU box): do two half-batches on two streams hide each other's update launches?  

One batch of 1024 images solved (a) in one call on one stream, (b) as two halves on two streams of equal priority,
(c) as two halves on a high- and a low-priority stream, (d) as four quarters on four streams.  Whole-solve wall time, median of 15."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import LMOptimizer
LMOptimizer.overlap_streams = 1      # these probes time single launches (the library default would split a large batch over two streams)
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "pinhole"
B, H, W = 1024, 480, 640
data, gtc, _ = synth_fields(model, B, H, W, dev, seed=1)
opt = LMOptimizer({"camera_model": model, "num_steps": 20, "early_stop": False}).eval()


def parts(n):
    step = B // n
    return [{k: v[i * step:(i + 1) * step] for k, v in data.items()} for i in range(n)]


def run(streams, chunks):
    torch.cuda.synchronize()
    t = time.perf_counter()
    outs = []
    for s, d in zip(streams, chunks):
        with torch.cuda.stream(s):
            outs.append(opt(d))
    torch.cuda.synchronize()
    return time.perf_counter() - t, outs


lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
cases = {
    "one call, one stream": ([torch.cuda.current_stream()], parts(1)),
    "two halves, two streams, equal priority": ([torch.cuda.Stream(), torch.cuda.Stream()], parts(2)),
    "two halves, high + low priority": ([torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)], parts(2)),
    "four quarters, four streams": ([torch.cuda.Stream() for _ in range(4)], parts(4)),
    "four quarters, priorities -1,-1,0,0": ([torch.cuda.Stream(priority=p) for p in (-1, -1, 0, 0)], parts(4)),
}
ref = None
for rep in range(2):
    for name, (streams, chunks) in cases.items():
        for _ in range(3): run(streams, chunks)
        ts = sorted(run(streams, chunks)[0] for _ in range(15))
        _, outs = run(streams, chunks)
        cam = torch.cat([o["camera"]._data for o in outs])
        if ref is None: ref = cam
        print(f"{model:14s} {name:42s}: {ts[7]*1e3:7.3f} ms  (min {ts[0]*1e3:.3f})  = {B/ts[7]:8.0f} img/s   identical to one call: {torch.equal(cam, ref)}", flush=True)
