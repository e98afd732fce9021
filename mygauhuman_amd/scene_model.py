"""Minimal articulated-Gaussian model: exactly the attributes gaussian_renderer.render() reads from the reference's
GaussianModel (scene/gaussian_model.py:152-209), so render() can run without the reference's dataset / optimizer /
densification machinery (out of scope this round).  A real GaussianModel instance from the reference
works with render() as well -- only the attribute names matter."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import covariance
from .sh_utils import RGB2SH


class HumanGaussianModel:
    def __init__(self, sh_degree, smpl=None, motion_offset_flag=False, device="cuda"):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        self.motion_offset_flag = motion_offset_flag
        self.device = torch.device(device)
        self.SMPL_NEUTRAL = smpl  # dict of device tensors: v_template, shapedirs, posedirs, J_regressor, weights, kintree_table
        self.pose_decoder = None
        self.lweight_offset_decoder = None
        e = torch.empty(0, device=self.device)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self._normal = self._albedo = self._roughness = e
        self.optimizer = None  # densify.training_setup() creates the Adam groups and the densification statistics

    @classmethod
    def from_points(cls, points, colors, sh_degree, dist2, smpl=None, motion_offset_flag=False, device="cuda"):
        """create_from_pcd (scene/gaussian_model.py:215-248): scales from distCUDA2, identity rotations, opacity 0.1."""
        m = cls(sh_degree, smpl, motion_offset_flag, device)
        dev = m.device
        P = points.shape[0]
        M = (sh_degree + 1) ** 2
        feats = torch.zeros((P, 3, M), device=dev)
        feats[:, :3, 0] = RGB2SH(colors.to(dev).float())
        scales = torch.log(torch.sqrt(torch.clamp_min(dist2, 0.0000001)))[..., None].repeat(1, 3)
        rots = torch.zeros((P, 4), device=dev)
        rots[:, 0] = 1
        opac = torch.full((P, 1), math.log(0.1 / 0.9), device=dev)
        req = lambda t: t.clone().detach().float().requires_grad_(True)  # noqa: E731
        m._xyz = req(points.to(dev))
        m._features_dc = req(feats[:, :, 0:1].transpose(1, 2).contiguous())
        m._features_rest = req(feats[:, :, 1:].transpose(1, 2).contiguous())
        m._scaling, m._rotation, m._opacity = req(scales), req(rots), req(opac)
        m._normal = req(F.normalize(torch.randn((P, 3), device=dev), dim=1))
        m._albedo = req(torch.zeros((P, 3), device=dev))
        m._roughness = req(torch.ones((P, 1), device=dev))  # [P,1] like the reference (:241)
        return m

    @classmethod
    def from_arrays(cls, g, sh_degree, smpl=None, motion_offset_flag=False, device="cuda", seed=0):
        """From a synthetic.uniform_gaussians() dict (activated values -> raw parameters)."""
        m = cls(sh_degree, smpl, motion_offset_flag, device)
        dev = m.device
        rng = np.random.default_rng(seed)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).float().requires_grad_(True)  # noqa: E731
        P = g["means3D"].shape[0]
        m._xyz = t(g["means3D"])
        m._features_dc = t(g["shs"][:, 0:1, :])
        m._features_rest = t(g["shs"][:, 1:, :])
        m._scaling = t(np.log(g["scales"]))
        m._rotation = t(g["rotations"])
        op = np.clip(g["opacities"], 1e-6, 1 - 1e-6)
        m._opacity = t(np.log(op / (1 - op)))
        m._normal = t(rng.normal(0, 1, (P, 3)).astype(np.float32))
        m._albedo = t(rng.normal(0, 1, (P, 3)).astype(np.float32))
        m._roughness = t(rng.normal(0, 1, (P, 1)).astype(np.float32))
        return m

    # ---- the accessors render() uses (same names as the reference)
    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return F.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_normal(self):
        return self._normal / self._normal.norm(dim=1, keepdim=True)

    def get_minimum_axis(self, dir_pp_normalized=None):
        axis = covariance.get_minimum_axis(self.get_scaling, self.get_rotation)
        normal_axis, _ = covariance.flip_align_view(axis, dir_pp_normalized)
        return normal_axis / normal_axis.norm(dim=1, keepdim=True)

    @property
    def get_albedo(self):
        return torch.sigmoid(self._albedo)

    @property
    def get_roughness(self):  # the reference returns the albedo activation here too (scene/gaussian_model.py:197-199)
        return torch.sigmoid(self._albedo)

    def frame_activations(self):
        """All of the above activations in one kernel (mygauhuman_amd.activations): render() asks for this when the model
        offers it and reads the property getters otherwise (the reference's own GaussianModel)."""
        from .activations import frame_activations
        from types import SimpleNamespace
        opacity, albedo, scaling, rotation, normal, occlusion = frame_activations(self._opacity, self._albedo, self._scaling,
                                                                                  self._rotation, self._normal)
        return SimpleNamespace(opacity=opacity, albedo=albedo, roughness=albedo, scaling=scaling, rotation=rotation, normal=normal,
                               occlusion=occlusion)

    def get_covariance(self, scaling_modifier=1, transform=None):
        return covariance.build_covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation, transform)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity, self._normal,
                self._albedo]
