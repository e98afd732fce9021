"""Seeded synthetic ARTICULATED scene ("S-human") for the render() workloads of bench.py, the tools and the
view-parallel tests: an SMPL-shaped body model (random template / blend shapes / regressor / skinning weights with the
standard 24-joint tree -- the real SMPL_NEUTRAL.pkl needs a registration download and is not in the reference tree), P canonical
Gaussians scattered around its vertices, ring cameras at 2.4 m looking at the body (BASELINE configs[3]: "8 ZJU-MoCap views per
step"), one target pose per view, and -- for motion_offset_flag models -- two small MLPs with the call surface of the
reference's decoders (nets/mlp_delta_body_pose.py: pose_decoder(posevec)["Rs"] [1,23,3,3]; nets/mlp_delta_weight_lbs.py:
lweight_offset_decoder(xyz[1,P,3]) -> [1,24,P]).  Everything is generated on the CPU from numpy's default_rng: all ranks and
all devices see the same bits."""
import math

import numpy as np
import torch

from . import cameras
from .lbs import batch_rodrigues
from .scene_model import HumanGaussianModel

PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int64)


class PoseRefiner(torch.nn.Module):
    """posevec [1,69] -> {"Rs": [1,23,3,3]}: a small MLP whose output (23 axis-angle corrections, initialised near zero) goes
    through rodrigues, like BodyPoseRefiner."""

    def __init__(self, width=128, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = torch.nn.Parameter(torch.randn((69, width), generator=g) / math.sqrt(69.0))
        self.b1 = torch.nn.Parameter(torch.zeros(width))
        self.w2 = torch.nn.Parameter(1e-2 * torch.randn((width, 69), generator=g) / math.sqrt(width))
        self.b2 = torch.nn.Parameter(torch.zeros(69))

    def forward(self, posevec):
        h = torch.relu(posevec.reshape(1, 69) @ self.w1 + self.b1)
        rv = (h @ self.w2 + self.b2).reshape(23, 3)
        return {"Rs": batch_rodrigues(rv).reshape(1, 23, 3, 3)}


class LbsOffsetDecoder(torch.nn.Module):
    """xyz [1,P,3] -> skinning-weight logit offsets [1,24,P] (LBSOffsetDecoder's call surface).  A STAND-IN, not the reference's
    network: an AFFINE map of the position -- offsets[j] = b[j] + A[:, j] . xyz, 96 parameters, three broadcast multiply-adds.
    The reference runs its real decoder EVERY frame when motion_offset_flag is set (gaussian_renderer/__init__.py:100-106 calls
    pc.lweight_offset_decoder(means3D)): a 63-d positional embedding through four 128-wide Conv1d layers on every Gaussian
    (nets/mlp_delta_weight_lbs.py), i.e. ~2 x (63 x 128 + 3 x 128 x 128 + 128 x 24) = 120 kFLOP per Gaussian per frame forward,
    a 200k x 128 GEMM chain that is likely the largest single cost of the reference's own step.  That network is outside SURVEY.md
    section 8 (render() calls whatever `pc.lweight_offset_decoder` is), so every figure measured with this stand-in -- `bench.py
    --workload render`, the view-parallel payload -- OMITS it, and the bench line says so in config.workload.  (A stand-in MLP with
    a 3- or 24-wide side spends the frame in rocBLAS' skinny-GEMM kernels: a 3-64-24 MLP on 200k points measured 1.2 ms of two
    rocBLAS launches per frame, more than the whole render() frame -- which is why the stand-in is affine.)"""

    def __init__(self, seed=1):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.A = torch.nn.Parameter(0.05 * torch.randn((3, 24), generator=g))
        self.b = torch.nn.Parameter(torch.zeros(24))

    def forward(self, xyz):
        x = xyz[0]
        out = self.b + x[:, 0:1] * self.A[0] + x[:, 1:2] * self.A[1] + x[:, 2:3] * self.A[2]   # [P, 24]
        return out.t()[None]


def body_arrays(V=6890, seed=0):
    rng = np.random.default_rng(seed)
    vt = rng.uniform(-1, 1, (V, 3)).astype(np.float32) * np.array([0.45, 0.9, 0.15], np.float32)
    J = rng.uniform(0, 1, (24, V)).astype(np.float32)
    w = rng.uniform(0, 1, (V, 24)).astype(np.float32) ** 4
    return dict(v_template=vt, shapedirs=rng.normal(0, 0.01, (V, 3, 10)).astype(np.float32),
                posedirs=rng.normal(0, 0.001, (V, 3, 207)).astype(np.float32), J_regressor=J / J.sum(1, keepdims=True),
                weights=(w / w.sum(1, keepdims=True)).astype(np.float32))


def gaussian_arrays(body, P, seed=0, scale=0.006):
    rng = np.random.default_rng(seed + 1)
    vt = body["v_template"]
    pts = (vt[rng.integers(0, vt.shape[0], P)] + rng.normal(0, 0.01, (P, 3))).astype(np.float32)
    return dict(means3D=pts, scales=np.exp(rng.normal(np.log(scale), 0.3, (P, 3))).astype(np.float32),
                rotations=rng.normal(0, 1, (P, 4)).astype(np.float32),
                opacities=(1 / (1 + np.exp(-rng.normal(0, 1.5, (P, 1))))).astype(np.float32),
                shs=np.concatenate([rng.normal(0, 1, (P, 1, 3)), rng.normal(0, 0.1, (P, 15, 3))], 1).astype(np.float32))


def view_camera(body, W, H, view, n_views=8, device="cuda", radius=2.4, fov_deg=50.0, pose_scale=0.15):
    """Ring camera `view` of n_views (view 0 looks along +z from z = -radius, like tools/render_bench.py) with ITS OWN target pose
    and shape (seeded by the view index) and the shared big pose."""
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
    cam_np = cameras.ring_camera(W, H, view % n_views, n_views, radius=radius, fov_deg=fov_deg)
    rng = np.random.default_rng(1000 + view)
    sp = dict(poses=d(rng.normal(0, pose_scale, (1, 72))), shapes=d(rng.normal(0, 0.5, (1, 10))), R=d(np.eye(3)),
              Th=d(np.zeros((1, 3))))
    bp = dict(poses=d(np.zeros((1, 72))), shapes=d(np.zeros((1, 10))), R=d(np.eye(3)), Th=d(np.zeros((1, 3))))
    cam = cameras.ViewCamera(cam_np, device, sp, bp, d(body["v_template"]))
    cam.cam_np = cam_np
    return cam


def build(P, V=6890, device="cuda", seed=0, motion=False, sh_degree=3, decoder="affine"):
    """(model, body arrays).  model.SMPL_NEUTRAL holds the body tables as device tensors; motion=True attaches the two decoders
    (decoder = "affine": the 96-parameter stand-in; "reference_size": nets.FusedLBSOffsetDecoder -- the reference network's layers,
    random init -- on the fused kernels; "reference_size_torch": the same module in torch ops)."""
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    body = body_arrays(V, seed)
    smpl = {k: d(v) for k, v in body.items()}
    smpl["kintree_table"] = torch.from_numpy(np.stack([PARENTS, np.arange(24)])).to(device)
    model = HumanGaussianModel.from_arrays(gaussian_arrays(body, P, seed), sh_degree, smpl=smpl, motion_offset_flag=motion,
                                           device=device, seed=seed)
    if motion:
        model.pose_decoder = PoseRefiner().to(device)
        if decoder in ("reference_size", "reference_size_torch"):
            from .nets import FusedLBSOffsetDecoder
            torch.manual_seed(seed + 7)
            net = FusedLBSOffsetDecoder().to(device)
            with torch.no_grad():
                net.bw_fc.weight.mul_(0.05)      # small offsets around the SMPL weights, like a network early in training
            net.use_fused = decoder == "reference_size"
            model.lweight_offset_decoder = net
        else:
            model.lweight_offset_decoder = LbsOffsetDecoder().to(device)
    return model, body
