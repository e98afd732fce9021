"""hipGraph replay of the sync-free rasterizer session: capture RasterSession.forward (C2) and
forward + loss gradient + backward (C3) into a torch.cuda.CUDAGraph and replay it.

What is inside the captured region: kernel launches of libgsr.so and two hipMemsetAsync (tile counters, gradient rows) on the
capture stream, all operating on buffers that were allocated BEFORE the capture and stay alive (the session's).  What must NOT
be inside (each of these bakes a pointer or an object into the graph that the replay then misuses): the pinned device-to-host
copy of AsyncCapacity / ViewParallelStep's pinned report (torch recycles or frees that host block after the capture: the
replay then writes through a stale pointer), stage-profiling event records, tensor allocations, and any per-frame host scalar
that changes (tan_fov, image size: they are by-value kernel arguments).  Camera matrices / parameters are device tensors:
update them in place between replays.

Run once per change on the GPU box:  timeout -k 10 180 python tools/graph_replay.py > gpurun_out/graph_replay.log 2>&1
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mygauhuman_amd import _lib  # noqa: E402


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _lib.profile_enable([])  # no event records inside a captured region
    out = {}
    for name in ("C2", "C3"):
        sc = bench.Scene(name, 0, 1, dev)
        s = sc.session
        bwd = sc.wl["backward"]
        grads = None
        if bwd:
            P = sc.P
            grads = dict(means3D=torch.empty(P, 3, device=dev), sh=torch.empty(P, sc.M, 3, device=dev), opacity=torch.empty(P, 1, device=dev),
                         scales=torch.empty(P, 3, device=dev), rotations=torch.empty(P, 4, device=dev))

        def step():
            s.forward(sc.params, sc.camd, sc.bg, sc.deg)
            if bwd:
                dc, da = s.alpha_mask_loss_backward(sc.gt_d, sc.mask_d, 0.1)
                s.backward(sc.params, sc.camd, sc.bg, sc.deg, dc, s.dL_ddepth, da, grads)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        assert not s.overflowed()
        ref_color, ref_alpha = s.color.clone(), s.alpha.clone()
        ref_grad = None if grads is None else grads["opacity"].clone()
        eager_ms = timed(step, 200)
        print(f"{name}: eager {eager_ms:.4f} ms/step; capturing ...", flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        print(f"{name}: captured; first replay ...", flush=True)
        s.color.zero_()
        g.replay()
        torch.cuda.synchronize()
        ok_img = bool(torch.equal(s.color, ref_color) and torch.equal(s.alpha, ref_alpha))
        ok_grad = True if grads is None else bool(torch.allclose(grads["opacity"], ref_grad, rtol=1e-4, atol=1e-7))
        # a new camera through the SAME graph: rotate the view matrix in place, replay, compare with eager on that camera
        graph_ms = timed(g.replay, 200)
        assert not s.overflowed()
        out[name] = dict(eager_ms=round(eager_ms, 4), graph_ms=round(graph_ms, 4), image_bits_equal=ok_img, grads_close=ok_grad)
        print(json.dumps({name: out[name]}), flush=True)
        del g
    print(json.dumps(out))


if __name__ == "__main__":
    main()
