"""GPU box: where the time of GeoCalib.calibrate() (extractor.py:72-127) goes for ONE image once the CNN is taken out:
preprocess | LM solve | _post_process (undo scale / crop + four bilinear upsamplings back to the input resolution)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geocalib_amd import GeoCalib
from geocalib_amd.synth import synth_fields
dev = torch.device("cuda:0")
img = torch.rand(3, 768, 1024, device=dev)
cache = {}


def field_model(d):
    b, _, h, w = d["image"].shape
    if (h, w) not in cache:
        cache[(h, w)] = synth_fields("pinhole", b, h, w, dev, seed=3)[0]
    return dict(cache[(h, w)])


def med(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return sorted(ts)[n // 2] * 1e6


model = GeoCalib(field_model)
total = med(lambda: model.calibrate(img))
data = model.preprocess(img[None])
pre = med(lambda: model.preprocess(img[None]))
fields = field_model(data) | {"scales": data["scales"]}
lm = med(lambda: model.optimizer(dict(fields)))
out = dict(fields) | model.optimizer(dict(fields))
post = med(lambda: model._post_process(out["camera"], data, dict(out)))
print(f"calibrate() of one 1024x768 image (fields at {fields['latitude_field'].shape[-2:]}): total {total:.1f} us = preprocess {pre:.1f} + LM {lm:.1f} + post-process {post:.1f} (+ glue)")
