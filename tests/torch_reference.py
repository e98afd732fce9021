"""Plain-PyTorch formulations of the per-frame glue of the hot path, used by the tests (and by tools/ that time the
reference's structure) as the fp32 / fp64 CHECKERS of the HIP kernels.  They are not product code: nothing under
mygauhuman_amd/ imports this module.

  pose chain      rodrigues -> optional refinement product -> kinematic chain -> rest-pose removal
                  (what scene/gaussian_model.py:894-980 computes; checks csrc/pose.hip)
  frame_attributes_torch   covariance T R S S^T R^T T^T, SH colour, the six feature colour sets
                  (gaussian_renderer/__init__.py:128-198; checks csrc/attributes.hip)
  ssim_torch      grouped-conv2d SSIM (utils/loss_utils.py:36-66; checks csrc/ssim.hip)
"""
from math import exp

import torch
import torch.nn.functional as F

from mygauhuman_amd import covariance
from mygauhuman_amd.lbs import batch_rodrigues, parents_host  # noqa: F401  (re-exported for the tests)
from mygauhuman_amd.sh_utils import eval_sh


# ------------------------------------------------------------------------------------------ SMPL pose chain
def rigid_chain(rot_mats, joints, parents):
    """rot_mats [B,J,3,3], joints [B,J,3], parents [J] -> per-joint 3x4 transforms with the rest pose removed:
    G_j = G_parent(j) [R_j | t_j - t_parent(j)],  A_j = [G_j.R | G_j.t - G_j.R t_j]."""
    B, J = joints.shape[:2]
    Rs, ts = [rot_mats[:, 0]], [joints[:, 0]]
    for j in range(1, J):
        p = int(parents[j])
        Rs.append(Rs[p] @ rot_mats[:, j])
        ts.append(ts[p] + (Rs[p] @ (joints[:, j] - joints[:, p])[..., None])[..., 0])
    R, t = torch.stack(Rs, 1), torch.stack(ts, 1)
    t = t - (R @ joints[..., None])[..., 0]
    A = torch.zeros((B, J, 4, 4), dtype=rot_mats.dtype, device=rot_mats.device)
    A[..., :3, :3], A[..., :3, 3], A[..., 3, 3] = R, t, 1.0
    return A


def pose_transforms_torch(smpl, params, rot_mats=None, correct_Rs=None):
    """(A [B,24,4,4], R, Th, joints) of an SMPL parameter dict -- the checker of lbs.smpl_pose_transforms."""
    betas = params["shapes"]
    v_shaped = smpl["v_template"][None] + torch.einsum("vcl,bl->bvc", smpl["shapedirs"][..., :betas.shape[-1]].to(betas.dtype), betas)
    if rot_mats is None:
        rot_mats = batch_rodrigues(params["poses"].reshape(-1, 3)).view(params["poses"].shape[0], -1, 3, 3)
        if correct_Rs is not None:
            rot_mats = torch.cat([rot_mats[:, :1], rot_mats[:, 1:] @ correct_Rs.reshape(rot_mats.shape[0], -1, 3, 3)], dim=1)
    joints = torch.einsum("jv,bvc->bjc", smpl["J_regressor"].to(v_shaped.dtype), v_shaped)
    return rigid_chain(rot_mats, joints, list(parents_host(smpl))), params["R"], params["Th"], joints


def smpl_pose_transforms_torch(smpl, params, correct_Rs=None):
    """Same return tuple as lbs.smpl_pose_transforms -- (A [1,24,4,4], rot_mats [1,24,3,3], joints [1,24,3]) -- through torch ops."""
    rot = batch_rodrigues(params["poses"].reshape(-1, 3)).view(1, -1, 3, 3)
    if correct_Rs is not None:
        rot = torch.cat([rot[:, :1], rot[:, 1:] @ correct_Rs.reshape(1, -1, 3, 3)], dim=1)
    A, _, _, joints = pose_transforms_torch(smpl, params, rot_mats=rot)
    return A, rot, joints


# ------------------------------------------------------------------------------------------ per-frame attributes
def _view_colour(v, viewmatrix):
    t = covariance.transformVector3x3(v, viewmatrix)
    return torch.stack([t[:, 0], -t[:, 1], t[:, 2]], dim=1) * 0.5 + 0.5


def frame_attributes_torch(means3D, transforms, world_normals, scales, scale_modifier, rot_cov, rot_axis, albedo, roughness,
                           occlusion, shs, sh_degree, campos, viewmatrix):
    dir_pp = means3D - campos.reshape(1, 3)
    dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    axis, _ = covariance.flip_align_view(covariance.get_minimum_axis(scales, rot_axis), dirn)
    axis = axis / axis.norm(dim=1, keepdim=True)
    world_axis = covariance.bmm3(transforms, axis[..., None]).squeeze(-1)
    world_axis = world_axis / world_axis.norm(dim=1, keepdim=True)
    wn = world_normals / world_normals.norm(dim=1, keepdim=True)
    cov3D = covariance.build_covariance_from_scaling_rotation(scales, scale_modifier, rot_cov, transforms)
    colors = None
    if shs is not None:
        colors = torch.clamp_min(eval_sh(sh_degree, shs.transpose(1, 2), dirn) + 0.5, 0.0)
    rough3 = roughness.mean(dim=1)[:, None].repeat(1, 3)
    features = torch.cat([_view_colour(wn, viewmatrix), wn * 0.5 + 0.5, albedo, occlusion, rough3,
                          _view_colour(world_axis, viewmatrix)], dim=1)
    return cov3D, colors, features


# ------------------------------------------------------------------------------------------ SSIM
def _window_2d(window_size, sigma=1.5):
    centre = window_size // 2
    taps = torch.tensor([exp(-((i - centre) ** 2) / (2.0 * sigma * sigma)) for i in range(window_size)], dtype=torch.float32)
    taps = taps / taps.sum()
    return torch.outer(taps, taps)


def ssim_torch(img1, img2, window_size=11, size_average=True):
    C = img1.size(-3)
    win = _window_2d(window_size).to(device=img1.device, dtype=img1.dtype).expand(C, 1, window_size, window_size).contiguous()
    blur = lambda t: F.conv2d(t, win, padding=window_size // 2, groups=C)  # noqa: E731
    m1, m2 = blur(img1), blur(img2)
    m11, m22, m12 = m1 * m1, m2 * m2, m1 * m2
    v1, v2, v12 = blur(img1 * img1) - m11, blur(img2 * img2) - m22, blur(img1 * img2) - m12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * m12 + c1) * (2 * v12 + c2)) / ((m11 + m22 + c1) * (v1 + v2 + c2))
    return smap.mean() if size_average else smap.mean(1).mean(1).mean(1)
