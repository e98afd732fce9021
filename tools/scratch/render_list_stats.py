"""Per-tile list statistics of the render_bench scene (human-shaped cloud): how many tiles are busy, how long their lists are."""
import os, sys, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mygauhuman_amd.diff_gaussian_rasterization import _C
from mygauhuman_amd.gaussian_renderer import render
from tools.train_demo import build

model, cam, _ = build(200_000, 6890, 1024, 1024)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
bg = torch.zeros(3, device="cuda")
captured = {}
orig = _C.rasterize_gaussians_async
def spy(*a, **k):
    out = orig(*a, **k)
    captured["out"] = out
    captured["P"] = a[1].shape[0]
    return out
_C.rasterize_gaussians_async = spy
with torch.no_grad():
    o = render(1, cam, model, pipe, bg)
torch.cuda.synchronize()
out = captured["out"]
cap, geom, binb, img = out[0], out[5], out[6], out[7]
ranges = _C.query_state("RANGES", captured["P"], cap, 1024, 1024, geom, binb, img).cpu().numpy().reshape(-1, 2)
n = (ranges[:, 1].astype(np.int64) - ranges[:, 0]).clip(0)
busy = n[n > 0]
print("tiles", len(n), "busy", len(busy), "instances", int(n.sum()), "mean list", busy.mean(), "median", np.median(busy), "p90", np.percentile(busy, 90), "max", busy.max())
print("histogram of list lengths:", np.histogram(busy, bins=[1, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 1 << 20])[0])
nc = o["radii"]
print("visible", int((nc > 0).sum()))
