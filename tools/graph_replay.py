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
        ref_all = None if grads is None else {k: v.clone() for k, v in grads.items()}
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
        if grads is not None:  # replays must not accumulate: the gradient rows are zeroed by a memset node inside the graph
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            ok_grad = ok_grad and bool(torch.allclose(grads["opacity"], ref_grad, rtol=1e-4, atol=1e-7))
            # eager GPU work between replays (an optimizer step in real life) must not disturb the next replay
            junk = torch.empty(1 << 22, device=dev)
            junk.fill_(1.0)
            float(junk.sum())
            g.replay()
            torch.cuda.synchronize()
            rel = {k: float((grads[k] - ref_all[k]).abs().max() / ref_all[k].abs().max()) for k in grads}
            ok_grad = ok_grad and max(rel.values()) < 1e-4
            print(f"{name}: after 4 replays max |grad - ref| / max|ref| per tensor = {rel}", flush=True)
        # new inputs through the SAME graph: move every Gaussian a little in place, eager reference first, then replay
        if bwd:
            for trial in range(3):
                sc.params["means3D"].add_(0.002 * torch.randn_like(sc.params["means3D"]))
                step()
                torch.cuda.synchronize()
                want = {k: v.clone() for k, v in grads.items()}
                want_img = s.color.clone()
                junk = torch.empty(1 << 22, device=dev); junk.fill_(float(trial)); float(junk.sum())
                for v in grads.values():
                    v.zero_()
                g.replay()
                torch.cuda.synchronize()
                rel = {k: float((grads[k] - want[k]).abs().max() / want[k].abs().max()) for k in grads}
                print(f"{name}: moved Gaussians, trial {trial}: image equal {bool(torch.equal(s.color, want_img))}, grads rel {rel}", flush=True)
                ok_grad = ok_grad and max(rel.values()) < 1e-4
        graph_ms = timed(g.replay, 200)
        assert not s.overflowed()
        out[name] = dict(eager_ms=round(eager_ms, 4), graph_ms=round(graph_ms, 4), image_bits_equal=ok_img, grads_close=ok_grad)
        print(json.dumps({name: out[name]}), flush=True)
        del g
    print(json.dumps(out))


if __name__ == "__main__":
    main()
