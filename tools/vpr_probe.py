# usage (GPU box, repo root): python tools/vpr_probe.py -> ms/step of ViewParallelRender with the fused loss / the temporal cache on and off
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.getcwd())
import torch
import bench
from mygauhuman_amd import human_synth, parallel, lbs
from mygauhuman_amd.diff_gaussian_rasterization._C import Phase1Loss
dev = torch.device("cuda", 0)
wl = bench.RENDER_WL
for motion in (True,):
    model, body = human_synth.build(wl["P"], wl["V"], dev, seed=0, motion=motion)
    cam = human_synth.view_camera(body, wl["W"], wl["H"], 0, n_views=8, device=dev)
    bg = torch.zeros(3, device=dev)
    spec = Phase1Loss(*bench._phase1_targets(wl["W"], wl["H"], dev))
    for fused, cache in ((True, True), (False, False), (True, False), (False, True), (True, True), (False, False)):
        if True:
            lbs.NN_TEMPORAL_CACHE = cache
            step = parallel.ViewParallelRender(model, bench._render_pipe(), bg)
            if fused:
                one = lambda: step(1, cam, lambda o: o["loss"], fused_loss=spec)
            else:
                one = lambda: step(1, cam, bench._phase1_loss)
            for _ in range(10):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                one()
            torch.cuda.synchronize()
            print(f"motion={motion} fused={fused} nn_cache={cache}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms/step", flush=True)
