"""Time gaussian_renderer.render() forward+backward: fused multi-feature pass vs the reference's seven passes."""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
from mygauhuman_amd import cameras  # noqa: E402
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402
from mygauhuman_amd.scene_model import HumanGaussianModel  # noqa: E402

ALL_KEYS = ("render", "normal", "albedo", "occlusion", "roughness", "world_normal", "render_axis")
PHASE1_KEYS = ("render", "render_alpha", "normal", "render_axis")  # the images train.py:256-286 puts in the loss before the PBR phase
PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int64)


def scene(P=200_000, V=6890, W=1024, H=1024):
    """The body scene of this tool: (model, camera, background)."""
    rng = np.random.default_rng(0)
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
    big = np.zeros(72, np.float32)
    sp = dict(poses=d(rng.normal(0, 0.15, (1, 72)).astype(np.float32)), shapes=d(rng.normal(0, 0.5, (1, 10)).astype(np.float32)),
              R=d(np.eye(3, dtype=np.float32)), Th=d(np.zeros((1, 3), np.float32)))
    bp = dict(poses=d(big[None]), shapes=d(np.zeros((1, 10), np.float32)), R=d(np.eye(3, dtype=np.float32)), Th=d(np.zeros((1, 3), np.float32)))
    cam = cameras.ViewCamera(cam_np, "cuda", sp, bp, d(vt))
    bg = torch.zeros(3, device="cuda")
    return model, cam, bg


def main(P=200_000, V=6890, W=1024, H=1024):
    model, cam, bg = scene(P, V, W, H)
    for sep, keys in ((False, PHASE1_KEYS if os.environ.get("KEYS") in ("phase1", "fused") else ALL_KEYS),) if (os.environ.get("PROFILE") or os.environ.get("ONLY")) else (((False, PHASE1_KEYS), (False, ALL_KEYS), (False, PHASE1_KEYS), (False, ALL_KEYS)) if os.environ.get("ORDER") else ((False, ALL_KEYS), (False, PHASE1_KEYS), (True, ALL_KEYS))):
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, separate_feature_passes=sep, sync_free_raster=os.environ.get("SYNC_FREE", "1") != "0")

        fused = None
        if os.environ.get("KEYS") == "fused":   # the phase-1 TRAINING loss of train.py:261-265, fused with the rasterizer (round 4)
            import bench
            from mygauhuman_amd.diff_gaussian_rasterization._C import Phase1Loss
            fused = Phase1Loss(*bench._phase1_targets(W, H, "cuda"))

        def step():
            for p in model.parameters():
                p.grad = None
            o = render(1, cam, model, pipe, bg, fused_loss=fused)
            loss = o["loss"] if fused is not None else sum(o[k].mean() for k in keys)
            loss.backward()
            return o
        for _ in range(30):  # first steps after a change of the loss composition load new torch kernels / grow the allocator
            o = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        if os.environ.get("PROFILE"):
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60), flush=True)
            print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60), flush=True)
        print(f"render() fwd+bwd, P={P}, {W}x{H}, visible={int((o['radii'] > 0).sum())}: "
              f"{'seven passes' if sep else 'fused'}, loss over {len(keys)} images: {dt:.2f} ms/frame", flush=True)


if __name__ == "__main__":
    main()
