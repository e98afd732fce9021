import sys, os, types
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tests import util
from tests.test_gpu_render import _human_scene
from oracle import oracle as orc
orc.build()
from mygauhuman_amd.gaussian_renderer import render
from mygauhuman_amd.graph import GraphedFrame
s = _human_scene(orc, seed=11)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
bg = util.to_dev(np.array([0.2, 0.3, 0.1], np.float32))
names = [str(i) + str(tuple(p.shape)) for i, p in enumerate(s.model.parameters())]
params = [p for p in s.model.parameters()]
keys = ("render", "render_alpha", "normal", "render_axis")
def step():
    o = render(1, s.cam, s.model, pipe, bg)
    sum(o[k].mean() for k in keys).backward()
    return o
def eager():
    for p in params: p.grad = None
    o = step()
    return o["render"].detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in params]
frame = GraphedFrame(step, warmup=3, zero_grads=params)
def report(tag, ge):
    for n, p, g in zip(names, params, ge):
        if g is None: 
            print(tag, n, "eager None, graph", None if p.grad is None else tuple(p.grad.shape)); continue
        sc = float(g.abs().max()) + 1e-20
        print(tag, n, tuple(g.shape), "rel err", float((p.grad - g).abs().max()) / sc, "ptr", p.grad.data_ptr())
def short(tag, ge):
    errs = []
    for n, p, g in zip(names, params, ge):
        if g is None or float(g.abs().max()) == 0: continue
        sc = float(g.abs().max())
        errs.append(float((p.grad - g).abs().max()) / sc)
    print(tag, "max rel err", max(errs), flush=True)
out = frame.replay(); torch.cuda.synchronize()
s.cam.smpl_param["poses"].add_(0.05 * torch.randn_like(s.cam.smpl_param["poses"]))
torch.cuda.synchronize()
img_e, ge = eager()
torch.cuda.synchronize()
out = frame.replay(); torch.cuda.synchronize(); short("new pose, 1st replay", ge)
out = frame.replay(); torch.cuda.synchronize(); short("new pose, 2nd replay", ge)
out = frame.replay(); torch.cuda.synchronize(); short("new pose, 3rd replay", ge)
# camera change instead of pose
s.cam.world_view_transform.add_(1e-3)
torch.cuda.synchronize()
img_e, ge = eager(); torch.cuda.synchronize()
out = frame.replay(); torch.cuda.synchronize(); short("new camera, 1st replay", ge)
out = frame.replay(); torch.cuda.synchronize(); short("new camera, 2nd replay", ge)
