"""Stress run: 1M Gaussians, SH3, 4096x4096, forward + loss gradient + backward through the sync-free session (finite outputs, no overflow)."""
import sys, numpy as np, torch, math, time
sys.path.insert(0, '/root/repo')
from mygauhuman_amd import synthetic, cameras, parallel
P, W, H = 1_000_000, 4096, 4096
g = synthetic.uniform_gaussians(P, seed=0, sh_degree=3, log_scale_mean=math.log(0.004))
cam = cameras.make_camera(W, H, 50.0)
to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
params = dict(means3D=to(g["means3D"]), shs=to(g["shs"]), opacities=to(g["opacities"]), scales=to(g["scales"]), rotations=to(g["rotations"]))
camd = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
bg = torch.zeros(3, device="cuda")
gt, mask = synthetic.loss_targets(W, H, seed=0)
step = parallel.ViewParallelStep(params, 3, camd, bg)
for _ in range(3): step(camd, bg, to(gt), to(mask), reduce=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): color, alpha, radii = step(camd, bg, to(gt), to(mask), reduce=False)
torch.cuda.synchronize()
print("1M Gaussians, 4096^2: %.2f ms/step, R=%d, overflow=%s, finite=%s, visible=%d" % ((time.perf_counter()-t0)/10*1e3, step.session.num_rendered(), step.session.overflowed(), bool(torch.isfinite(color).all() and all(torch.isfinite(v).all() for v in step.grads.values())), int((radii>0).sum())))
