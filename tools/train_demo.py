"""One full training iteration the way train.py:212-412 composes it (phase before PBR): render() -> L1 + mask L2 + normal / axis L1
+ SSIM(image) + SSIM(normal) -> backward -> densification statistics -> Adam step, on a synthetic articulated scene
(200k Gaussians, 1024^2), with a densify-and-prune every 100 iterations.  LPIPS (a VGG network) is left out.
Prints ms per iteration for the fused path and for the reference's structure on the same library (seven passes, torch glue)."""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd import cameras, densify, loss_utils  # noqa: E402
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402
from mygauhuman_amd.scene_model import HumanGaussianModel  # noqa: E402
from tools.render_bench import PARENTS  # noqa: E402


def build(P, V, W, H, seed=0):
    rng = np.random.default_rng(seed)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    vt = rng.uniform(-1, 1, (V, 3)).astype(np.float32) * np.array([0.45, 0.9, 0.15], np.float32)
    smpl = dict(v_template=d(vt), shapedirs=d(rng.normal(0, 0.01, (V, 3, 10)).astype(np.float32)),
                posedirs=d(rng.normal(0, 0.001, (V, 3, 207)).astype(np.float32)),
                J_regressor=d((lambda j: j / j.sum(1, keepdims=True))(rng.uniform(0, 1, (24, V)).astype(np.float32))),
                weights=d((lambda w: w / w.sum(1, keepdims=True))(rng.uniform(0, 1, (V, 24)).astype(np.float32) ** 4)),
                kintree_table=torch.from_numpy(np.stack([PARENTS, np.arange(24)])).cuda())
    pts = (vt[rng.integers(0, V, P)] + rng.normal(0, 0.01, (P, 3))).astype(np.float32)
    g = dict(means3D=pts, scales=np.exp(rng.normal(np.log(0.006), 0.3, (P, 3))).astype(np.float32),
             rotations=rng.normal(0, 1, (P, 4)).astype(np.float32),
             opacities=(1 / (1 + np.exp(-rng.normal(0, 1.5, (P, 1))))).astype(np.float32),
             shs=np.concatenate([rng.normal(0, 1, (P, 1, 3)), rng.normal(0, 0.1, (P, 15, 3))], 1).astype(np.float32))
    model = HumanGaussianModel.from_arrays(g, 3, smpl=smpl)
    cam_np = cameras.look_at_camera(W, H, eye=[0.0, 0.0, -2.4], target=[0.0, 0.0, 0.0], fov_deg=50.0)
    sp = dict(poses=d(rng.normal(0, 0.15, (1, 72)).astype(np.float32)), shapes=d(rng.normal(0, 0.5, (1, 10)).astype(np.float32)),
              R=d(np.eye(3, dtype=np.float32)), Th=d(np.zeros((1, 3), np.float32)))
    bp = dict(poses=d(np.zeros((1, 72), np.float32)), shapes=d(np.zeros((1, 10), np.float32)), R=d(np.eye(3, dtype=np.float32)),
              Th=d(np.zeros((1, 3), np.float32)))
    cam = cameras.ViewCamera(cam_np, "cuda", sp, bp, d(vt))
    return model, cam, d(vt)


def main(P=200_000, V=6890, W=1024, H=1024, iters=120):
    for sep, sync_free in ((False, True), (False, False), (False, True), (False, False), (True, False)):
        torch.manual_seed(0)
        model, cam, verts = build(P, V, W, H)
        densify.training_setup(model, dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=0.05, scaling=5e-3, rotation=1e-3,
                                          normal=1e-3, albedo=0.05, roughness=0.05))
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, separate_feature_passes=sep,
                                     sync_free_raster=sync_free)
        # the "reference structure" variant also runs the torch op chain for the per-frame attributes
        import mygauhuman_amd.gaussian_renderer as gr
        from mygauhuman_amd.attributes import frame_attributes
        from tests.torch_reference import frame_attributes_torch, ssim_torch
        from tests.util import GetterOnlyModel
        gr.frame_attributes = frame_attributes_torch if sep else frame_attributes
        bg = torch.zeros(3, device="cuda")
        gt = torch.rand((3, H, W), device="cuda")
        gt_n = torch.rand((3, H, W), device="cuda")
        mask = (torch.rand((1, H, W), device="cuda") > 0.5).float()
        ssim = ssim_torch if sep else loss_utils.ssim

        def iteration(it):
            # reference structure: the model is read through its property getters (torch ops), like the reference's class
            o = render(it, cam, GetterOnlyModel(model) if sep else model, pipe, bg)
            img, alpha, normal, axis = o["render"], o["render_alpha"], o["normal"], o["render_axis"]
            loss = (loss_utils.l1_loss(img, gt) + 0.1 * loss_utils.l2_loss(alpha, mask) + loss_utils.l1_loss(normal, gt_n) +
                    loss_utils.l1_loss(axis, gt_n) + 0.01 * (2.0 - ssim(img[None], gt[None]) - ssim(normal[None], gt_n[None])))
            loss.backward()
            with torch.no_grad():
                vis, radii = o["visibility_filter"], o["radii"]
                if sep:   # the reference's statements (train.py:403-404), boolean-mask indexing
                    model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis].float())
                    g = o["viewspace_points"].grad
                    model.xyz_gradient_accum[vis] += torch.norm(g[vis, :2], dim=-1, keepdim=True)
                    model.denom[vis] += 1
                else:
                    densify.update_max_radii(model, radii, vis)
                    densify.add_densification_stats(model, o["viewspace_points"], vis)
                if it % 100 == 0:
                    densify.densify_and_prune(model, 2e-4, 0.005, 2.0, 20, t_vertices=verts)
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)

        for it in range(1, 16):
            iteration(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(16, 16 + iters):
            iteration(it)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters * 1e3
        print(f"training iteration ({'reference structure: seven passes + torch glue + conv2d SSIM' if sep else 'fused path, sync_free_raster=' + str(sync_free)}), "
              f"P={model.get_xyz.shape[0]} after densification, {W}x{H}: {dt:.2f} ms/iteration ({1e3 / dt:.0f} it/s)", flush=True)


if __name__ == "__main__":
    main()
