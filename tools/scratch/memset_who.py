import os, sys, types
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import render_bench
from mygauhuman_amd.gaussian_renderer import render
model, cam, bg = render_bench.scene()
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, separate_feature_passes=False, sync_free_raster=True)
def step():
    for p in model.parameters():
        p.grad = None
    o = render(1, cam, model, pipe, bg)
    sum(o[k].mean() for k in render_bench.PHASE1_KEYS).backward()
for _ in range(10): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
evs = prof.events()
# memset device events and the CPU op that launched them (by correlation: nearest enclosing cpu op in time)
cpu_ops = [e for e in evs if e.device_type.name == "CPU"]
for e in evs:
    if e.device_type.name != "CPU" and ("memset" in e.name.lower() or "fillBuffer" in e.name):
        print("DEV", e.name, round(e.device_time if hasattr(e,'device_time') else e.cuda_time, 1), "us")
for e in cpu_ops:
    if e.name in ("aten::zero_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "hipMemsetAsync", "aten::fill_"):
        print("CPU", e.name, e.input_shapes, [s for s in (e.stack or [])[:6]])
