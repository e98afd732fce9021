#!/usr/bin/env python3
"""Temporal nearest-vertex cache on the bench scene: misses per frame for static points and for points that move like an optimizer
step, and the kernel times of the three launches (GPU box)."""
import os
import sys
import types

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mygauhuman_amd import human_synth, lbs  # noqa: E402

P = int(os.environ.get("P", 200_000))
model, body = human_synth.build(P, 6890, "cuda", seed=0)
cam = human_synth.view_camera(body, 1024, 1024, 0, n_views=8, device="cuda")
xyz = model.get_xyz.detach()
nrm = torch.nn.functional.normalize(model._normal.detach())
verts = cam.big_pose_world_vertex


def deform():
    return lbs.coarse_deform_c2source(model.SMPL_NEUTRAL, xyz[None], cam.smpl_param, cam.big_pose_smpl_param, verts[None], normals=nrm[None],
                                      lean=True)


def timed(n=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        deform()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def graph_time(n=200):
    """GPU time of one deform call as a hipGraph replay (no host overhead in the number)."""
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(3):
            deform()
    torch.cuda.current_stream().wait_stream(s_)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        deform()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for cached in (False, True, False, True):
        lbs.NN_TEMPORAL_CACHE = cached
        deform()
        print(f"cache={cached}: {graph_time():.1f} us of GPU time per coarse_deform_c2source call as a graph replay (static points)", flush=True)
    for cached in (False, True):
        lbs.NN_TEMPORAL_CACHE = cached
        deform()
        print(f"cache={cached}: {timed():.1f} us per coarse_deform_c2source call (static points)", flush=True)
    print("static points: (misses last frame, searches so far) =", lbs._GRIDS.nn_cache_stats(verts, P))
    ext = float((xyz.max(0).values - xyz.min(0).values).max())
    for step in (1e-5, 1e-4, 1e-3):
        tot = 0
        for it in range(20):
            xyz.add_(torch.randn_like(xyz) * (step * ext))
            deform()
            tot += lbs._GRIDS.nn_cache_stats(verts, P)[0]
        print(f"random steps of {step:g} x extent ({step * ext:.2e}): mean misses per frame {tot / 20:.0f} of {P}; {timed():.1f} us per call", flush=True)
