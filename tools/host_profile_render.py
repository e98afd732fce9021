"""Where does the HOST spend an eager render() training step?  cProfile over 200 steps (GPU box).   python tools/host_profile_render.py [motion]"""
import cProfile
import os
import pstats
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mygauhuman_amd import human_synth  # noqa: E402
from mygauhuman_amd.diff_gaussian_rasterization._C import Phase1Loss  # noqa: E402
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402

motion = len(sys.argv) > 1 and sys.argv[1] == "motion"
dev = torch.device("cuda", 0)
wl = bench.RENDER_WL
model, body = human_synth.build(wl["P"], wl["V"], dev, seed=0, motion=motion, decoder="reference_size")
cam = human_synth.view_camera(body, wl["W"], wl["H"], 0, n_views=8, device=dev)
bg, pipe = torch.zeros(3, device=dev), bench._render_pipe()
spec = Phase1Loss(*bench._phase1_targets(wl["W"], wl["H"], dev))
params = list(model.parameters()) + ([p for m in (model.pose_decoder, model.lweight_offset_decoder) for p in m.parameters()] if motion else [])


def step():
    for p in params:
        p.grad = None
    o = render(1, cam, model, pipe, bg, fused_loss=spec)
    o["loss"].backward()


for _ in range(30):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumtime").print_stats(45)
