"""Which tensors of one render() frame receive MORE THAN ONE gradient (autograd then adds them with an extra kernel each)?
Walks the autograd graph from the fused loss and counts the consumers of every (node, output).   python tools/autograd_fanout.py"""
import collections
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tools import render_bench  # noqa: E402
from mygauhuman_amd.diff_gaussian_rasterization._C import Phase1Loss  # noqa: E402
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402

model, cam, bg = render_bench.scene()
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, separate_feature_passes=False,
                             sync_free_raster=True)
fused = Phase1Loss(*bench._phase1_targets(1024, 1024, "cuda"))
o = render(1, cam, model, pipe, bg, fused_loss=fused)
loss = o["loss"]
seen, uses, stack = set(), collections.Counter(), [loss.grad_fn]
while stack:
    fn = stack.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    for nxt, idx in fn.next_functions:
        if nxt is not None:
            uses[(nxt, idx)] += 1
            stack.append(nxt)
for (fn, idx), n in sorted(uses.items(), key=lambda kv: -kv[1]):
    if n > 1:
        var = getattr(fn, "variable", None)
        print(n, "gradients ->", type(fn).__name__, "output", idx, "" if var is None else f"leaf {tuple(var.shape)}")
print("nodes:", len(seen))
