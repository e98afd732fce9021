"""gaussian_renderer.render() on the MI355X hot path -- same signature and result-dict keys as the reference
(gaussian_renderer/__init__.py:53-295; callers train.py:224,484 and render.py:200).

What changes underneath:
  * LBS deform of the canonical Gaussians      -> one HIP kernel (+ one in backward), mygauhuman_amd.lbs
  * covariance / SH colour / feature colours   -> one HIP kernel (+ one in backward), mygauhuman_amd.attributes
  * rasterisation                               -> mygauhuman_amd.diff_gaussian_rasterization (HIP)
  * the reference rasterises SEVEN times per frame with identical geometry and different colours (:203-272); here the
    seven images come out of ONE fused pass (one preprocess + binning, a 21-channel blend, one backward) with the same
    outputs and gradients; `pipe.separate_feature_passes = True` restores the seven separate calls;
  * no per-frame host read of num_rendered (CR/rasterizer_impl.cu:283): the fused pass runs through the sync-free entry with a
    generous binning capacity and a deferred, never-silent overflow check (diff_gaussian_rasterization._C.AsyncCapacity: examined
    when the frame's backward runs, at later frames, and by AsyncCapacity.check_all() / `with AsyncCapacity.frames():` for
    forward-only loops); `pipe.sync_free_raster = False` restores the reference's blocking read.
`pc` is any object exposing the reference GaussianModel accessors (scene_model.HumanGaussianModel or the reference's own
class); `viewpoint_camera` exposes FoVx, FoVy, image_height, image_width, world_view_transform, full_proj_transform,
camera_center, smpl_param, big_pose_smpl_param, big_pose_world_vertex (scene/cameras.py:17-74) and optionally
`occlusion`.  The post-30k-iteration occlusion baking (baking.py, nvdiffrast) is outside the hot path: pass
`viewpoint_camera.occlusion` if you have it, otherwise the opacity-derived placeholder of the reference's first 30k
iterations is used (:141).
"""
import math
import os

import torch

from .. import gradlink
from .. import lbs as _lbs
from ..attributes import frame_attributes
from ..covariance import bmm3
from ..diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

_ZERO_POINTS = {}


def _screenspace_leaf(xyz):
    """The gradient holder of the 2D means (:62-66: `zeros_like(...) + 0` and retain_grad()): a zero LEAF that requires grad.  Its
    values are never written by anybody, so every frame gets a fresh leaf over ONE shared zero buffer per (shape, dtype, device)
    instead of a fill kernel per frame."""
    key = (tuple(xyz.shape), xyz.dtype, str(xyz.device))
    z = _ZERO_POINTS.get(key)
    if z is None or (xyz.is_cuda and torch.cuda.is_current_stream_capturing() and not getattr(z, "_gsr_persistent", False)):
        z = torch.zeros_like(xyz, requires_grad=False)
        if not (xyz.is_cuda and torch.cuda.is_current_stream_capturing()):
            z._gsr_persistent = True
            if len(_ZERO_POINTS) > 16:
                _ZERO_POINTS.clear()
            _ZERO_POINTS[key] = z
    return z.detach().requires_grad_(True)


RESULT_KEYS = ("render", "render_depth", "render_alpha", "viewspace_points", "visibility_filter", "radii", "transforms",
               "translation", "correct_Rs", "normal", "albedo", "occlusion", "roughness", "world_normal", "render_axis")


def _deform(pc, means3D, normal, cam, lbs_weights=None, correct_Rs=None, return_transl=False):
    smpl = getattr(pc, "SMPL_NEUTRAL", None)
    if isinstance(smpl, dict) and "kintree_table" in smpl:
        return _lbs.coarse_deform_c2source(smpl, means3D[None], cam.smpl_param, cam.big_pose_smpl_param,
                                           cam.big_pose_world_vertex[None], lbs_weights=lbs_weights, correct_Rs=correct_Rs,
                                           return_transl=return_transl, normals=normal[None], lean=True)
    raise RuntimeError("render(): pc.SMPL_NEUTRAL (device tensors incl. kintree_table) is required for the LBS deform")


def _features_of(pc):
    """The SH coefficients for the attribute kernel: the model's two parameter tensors as they are (no torch.cat) when it has
    them in the reference's layout, get_features otherwise."""
    dc, rest = getattr(pc, "_features_dc", None), getattr(pc, "_features_rest", None)
    if (dc is not None and rest is not None and dc.dim() == 3 and rest.dim() == 3 and dc.shape[1] == 1 and rest.shape[1] == 15
            and dc.is_contiguous() and rest.is_contiguous()):
        return (dc, rest)
    return pc.get_features


# (gradlink.py) the position / quaternion gradients of a frame meet inside the attribute / activation kernels instead of in three
# accumulation kernels of autograd; GSR_GRAD_LINK=0 keeps autograd's own accumulation
GRAD_LINK = os.environ.get("GSR_GRAD_LINK", "1") != "0"


def render(iteration, viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None,
           return_smpl_rot=False, transforms=None, translation=None, envmap=None, fused_loss=None):
    """Render the scene. Background tensor (bg_color) must be on the GPU.
    fused_loss (extension, default None = the reference's behaviour): a diff_gaussian_rasterization._C.Phase1Loss -- the loss
    train.py:261-265 forms from this result (bound-masked L1 on image / normal / axis, 0.1 L2 on alpha) evaluated FUSED with the
    rasterizer: the result then carries "loss" (a 0-dim tensor; add the other terms to it and call backward())."""
    with gradlink.frame_link(GRAD_LINK and not getattr(pipe, "separate_feature_passes", False)):
        return _render_frame(iteration, viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, return_smpl_rot,
                             transforms, translation, envmap, fused_loss)


def _render_frame(iteration, viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, return_smpl_rot, transforms,
                  translation, envmap, fused_loss):
    dev = pc.get_xyz.device
    # the gradient holder of the 2D means (:62-66: `zeros_like(...) + 0` and retain_grad()): a zero LEAF that requires grad
    # receives the same .grad without the extra add kernel
    screenspace_points = _screenspace_leaf(pc.get_xyz)

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
    raster_settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    # the parameter activations: one fused kernel when the model offers it (scene_model.HumanGaussianModel.frame_activations),
    # the reference's property getters otherwise
    act = pc.frame_activations() if hasattr(pc, "frame_activations") else None
    means3D = pc.get_xyz
    normal = act.normal if act is not None else pc.get_normal
    correct_Rs = None
    if not pc.motion_offset_flag:
        _, means3D, _, transforms, _, world_normal = _deform(pc, means3D, normal, viewpoint_camera)
    elif transforms is None:
        dst_posevec = viewpoint_camera.smpl_param["poses"][:, 3:]
        correct_Rs = pc.pose_decoder(dst_posevec)["Rs"]
        lbs_weights = pc.lweight_offset_decoder(means3D[None].detach()).permute(0, 2, 1)
        _, means3D, _, transforms, translation, world_normal = _deform(pc, means3D, normal, viewpoint_camera, lbs_weights,
                                                                       correct_Rs, return_smpl_rot)
    else:  # cached per-pose transforms (render.py:169-195)
        means3D = bmm3(transforms, means3D[..., None]).squeeze(-1) + translation
        world_normal = bmm3(transforms, normal[..., None]).squeeze(-1)

    means3D = means3D.reshape(-1, 3)
    means2D = screenspace_points
    opacity = act.opacity if act is not None else pc.get_opacity
    albedo = act.albedo if act is not None else pc.get_albedo
    roughness = act.roughness if act is not None else pc.get_roughness
    scaling = act.scaling if act is not None else pc.get_scaling
    rotation_n = act.rotation if act is not None else pc.get_rotation
    occlusion = getattr(viewpoint_camera, "occlusion", None)
    if iteration > 30000 and occlusion is not None:
        occlusion = occlusion.detach()
        if envmap is not None:
            occ = torch.clamp(occlusion, min=0, max=1) * envmap.permute(1, 2, 0)
            _occlusion = occ.sum(dim=(1, 2)).repeat(1, 3).clamp(min=0.0, max=1.0)
        else:
            _occlusion = occlusion.sum(dim=(1, 2))
    else:
        _occlusion = act.occlusion if act is not None else opacity.repeat(1, 3)

    # covariance in the posed frame, view-dependent colour and the six feature colour sets (:120-198): one HIP kernel
    # (mygauhuman_amd.attributes)
    sh_python = override_color is None and pipe.convert_SHs_python
    cov3D_precomp, colors_precomp, features = frame_attributes(
        means3D, transforms.reshape(-1, 3, 3), world_normal.reshape(-1, 3), scaling, scaling_modifier, pc._rotation,
        rotation_n, albedo, roughness, _occlusion, _features_of(pc) if sh_python else None, pc.active_sh_degree,
        viewpoint_camera.camera_center, viewpoint_camera.world_view_transform)

    scales = rotations = shs = None
    if not pipe.compute_cov3D_python:
        cov3D_precomp = None
        scales, rotations = scaling, rotation_n
    if override_color is not None:
        colors_precomp = override_color
    elif not sh_python:
        shs = pc.get_features

    def raster(colors, use_shs=None):
        return rasterizer(means3D=means3D, means2D=means2D, shs=use_shs, colors_precomp=colors, opacities=opacity,
                          scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)

    if getattr(pipe, "separate_feature_passes", False):
        # the reference's structure: seven rasterizer calls with identical geometry (:203-272).  The feature passes
        # always use precomputed colours; with in-kernel SHs the reference would raise there (shs AND colors_precomp)
        rendered_image, radii, depth, alpha = raster(colors_precomp, shs)
        (rendered_normal, rendered_world_normal, rendered_albedo, rendered_occlusion, rendered_roughness,
         rendered_axis) = [raster(features[:, 3 * k:3 * k + 3])[0] for k in range(6)]
    else:
        # fused: one preprocess + binning + a 21-channel blend (and one backward) give the same seven images
        res = rasterizer.forward_multi(
            means3D=means3D, means2D=means2D, opacities=opacity, extra_colors=features, shs=shs, colors_precomp=colors_precomp,
            scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, sync_free=getattr(pipe, "sync_free_raster", True),
            loss_spec=fused_loss)
        rendered_image, radii, depth, alpha, feats = res[:5]
        fused_value = res[5] if fused_loss is not None else None
        rendered_normal, rendered_world_normal, rendered_albedo, rendered_occlusion, rendered_roughness, rendered_axis = feats
    if fused_loss is not None and getattr(pipe, "separate_feature_passes", False):
        raise RuntimeError("render(fused_loss=...): the fused loss rides on the fused multi-feature pass (pipe.separate_feature_passes = False)")

    extra_out = {} if fused_loss is None else {"loss": fused_value}
    return {**extra_out, "render": rendered_image, "render_depth": depth, "render_alpha": alpha, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii, "transforms": transforms, "translation": translation,
            "correct_Rs": correct_Rs, "normal": rendered_normal, "albedo": rendered_albedo, "occlusion": rendered_occlusion,
            "roughness": rendered_roughness, "world_normal": rendered_world_normal, "render_axis": rendered_axis}
