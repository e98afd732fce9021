"""Which torch ops (and of what shapes) are left in one render() training frame -- the producers of the small fill / add / copy
kernels that sit between the HIP kernels.  GPU box:  python tools/render_glue_census.py"""
import os
import sys
import types

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd import human_synth  # noqa: E402
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402

KEYS = ("render", "render_alpha", "normal", "render_axis")


def main(P=200_000):
    model, body = human_synth.build(P, 6890, "cuda", seed=0, motion=os.environ.get("MOTION") == "1")
    cam = human_synth.view_camera(body, 1024, 1024, 0, device="cuda")
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    bg = torch.zeros(3, device="cuda")
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        o = render(1, cam, model, pipe, bg)
        sum(o[k].mean() for k in KEYS).backward()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    n = 5
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=bool(os.environ.get('STACK'))) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dev_us = getattr(e, "self_device_time_total", 0.0) or getattr(e, "self_cuda_time_total", 0.0)
        if dev_us > 0 and e.key.startswith("aten::"):
            rows.append((dev_us / n, e.count / n, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    print(f"{'us/frame':>9} {'calls/frame':>11}  op  input shapes")
    tot = 0.0
    for us, c, k, sh in rows:
        tot += us
        print(f"{us:9.2f} {c:11.1f}  {k:28s} {sh}")
    print(f"torch ops with device time: {tot:.1f} us/frame")
    if os.environ.get("STACK"):   # who issues the fills / copies / adds: innermost repo frame of each call
        seen = {}
        for e in prof.events():
            if e.name in ("aten::fill_", "aten::copy_", "aten::add_", "aten::zero_", "aten::eye", "aten::gt", "aten::sub") and e.stack:
                mine = [f for f in e.stack if "mygauhuman_amd" in f or "tools/" in f]
                key = (e.name, str(e.input_shapes)[:60], mine[0] if mine else (e.stack[0] if e.stack else "?"))
                seen[key] = seen.get(key, 0) + 1
        for (name, sh, where), c in sorted(seen.items(), key=lambda kv: -kv[1]):
            print(f"{c / n:5.1f}/frame {name:12s} {sh:62s} {where}")


if __name__ == "__main__":
    main()
