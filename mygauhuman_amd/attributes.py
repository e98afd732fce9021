"""Per-frame Gaussian attributes of gaussian_renderer.render() (gaussian_renderer/__init__.py:128-198): the 3D covariance
in the posed frame, the view-dependent colour and the six feature colour sets (normal, world normal, albedo, occlusion,
roughness, minimum axis).

    frame_attributes(...)        one HIP kernel forward + one backward (csrc/attributes.hip); tensors must live on the GPU
(the chain of torch ops it replaces is kept with the tests as its checker: tests/torch_reference.py).
Returns (cov3D [P,6], colors [P,3] or None, features [P,18]) with
features = cat(normal, world_normal, albedo, occlusion, roughness.mean x3, axis) -- the `extra` operand of
diff_gaussian_rasterization.rasterize_gaussians_multi.
"""
import contextlib

import torch

from . import gradlink
from ._lib import check, lib, ptr

# view-parallel compact SH exchange (parallel.ViewParallelRender): while a sink is installed, the backward of the attribute
# kernel hands (forward colours, dL_dcolours, posed positions) to it -- the three things the rank-one SH gradient of a view is
# made of -- and skips the SH gradient itself when autograd does not ask for it (detached SH tensors)
_SH_SINK = None


@contextlib.contextmanager
def sh_gradient_sink(sink):
    global _SH_SINK
    prev, _SH_SINK = _SH_SINK, sink
    try:
        yield sink
    finally:
        _SH_SINK = prev


class _FrameAttributes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, transforms, world_normals, scales, rot_cov, rot_axis, albedo, roughness, occlusion, shs, campos,
                viewmatrix, scale_modifier, sh_degree, shs_rest=None):
        if not means3D.is_cuda:
            raise RuntimeError("frame_attributes: tensors must live on a HIP device (no CPU path)")
        dev, f32 = means3D.device, torch.float32
        P = means3D.shape[0]
        c = lambda t: None if t is None else t.detach().contiguous().float()  # noqa: E731
        ins = [c(means3D), c(transforms).reshape(P, 9), c(world_normals), c(scales), c(rot_cov), c(rot_axis), c(albedo),
               c(roughness), c(occlusion), c(shs), c(campos).reshape(3), c(viewmatrix).reshape(16)]
        rest = c(shs_rest)
        M = 0 if shs is None else int(shs.shape[1]) + (0 if rest is None else int(rest.shape[1]))
        cov3D = torch.empty((P, 6), dtype=f32, device=dev)
        colors = torch.empty((P, 3), dtype=f32, device=dev) if shs is not None else None
        features = torch.empty((P, 18), dtype=f32, device=dev)
        with torch.cuda.device(dev):
            check(lib.gsr_frame_attributes_forward_split(
                P, int(sh_degree), M, ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), ptr(ins[3]), float(scale_modifier), ptr(ins[4]),
                ptr(ins[5]), ptr(ins[6]), ptr(ins[7]), ptr(ins[8]), ptr(ins[9]), ptr(rest), ptr(ins[10]), ptr(ins[11]), ptr(cov3D),
                ptr(colors), ptr(features), torch.cuda.current_stream(dev).cuda_stream), "gsr_frame_attributes_forward")
        ctx.has_shs = shs is not None
        ctx.has_rest = rest is not None
        ctx.sink = _SH_SINK if shs is not None else None
        ctx.save_for_backward(*([t for t in ins if t is not None] + ([rest] if rest is not None else [])
                                + ([colors] if ctx.sink is not None else [])))
        ctx.meta = (float(scale_modifier), int(sh_degree), M, transforms.shape)
        # albedo and roughness as ONE tensor (get_roughness reads _albedo): the backward writes the sum of both gradients once
        ctx.rough_is_albedo = (albedo.data_ptr() == roughness.data_ptr() and albedo.shape == roughness.shape
                               and albedo.stride() == roughness.stride())
        # (gradlink) positions: the rasterizer of this frame may park its dL_dmeans3D for this backward to add in-kernel;
        # raw quaternion: this backward parks its gradient for the activations' backward when rot_axis is their normalised output
        link = gradlink.current()
        ctx.link, ctx.park_rot = link, False
        if link is not None:
            if ctx.needs_input_grad[0] and means3D.is_contiguous() and means3D.dtype == f32:
                link.attr_means_ptr = means3D.data_ptr()
            ctx.park_rot = (link.act_rot_in_ptr is not None and rot_cov.data_ptr() == link.act_rot_in_ptr
                            and rot_axis.data_ptr() == link.act_rot_out_ptr and ctx.needs_input_grad[4] and ctx.needs_input_grad[5])
        return cov3D, (colors if colors is not None else torch.empty(0, device=dev)), features

    @staticmethod
    def backward(ctx, g_cov, g_colors, g_features):
        saved = list(ctx.saved_tensors)
        colors_fwd = saved.pop() if ctx.sink is not None else None
        rest = saved.pop() if ctx.has_rest else None
        if not ctx.has_shs:
            saved.insert(9, None)
        means3D, transforms, wn, scales, rot_cov, rot_axis, albedo, roughness, occlusion, shs, campos, view = saved
        mod, D, M, t_shape = ctx.meta
        dev, f32 = means3D.device, torch.float32
        P = means3D.shape[0]
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        g_colors = c(g_colors) if (ctx.has_shs and g_colors is not None and g_colors.numel()) else None
        new = lambda *s: torch.empty(s, dtype=f32, device=dev)  # noqa: E731
        d_means, d_T, d_wn, d_scales, d_rc, d_ra = new(P, 3), new(P, 9), new(P, 3), new(P, 3), new(P, 4), new(P, 4)
        d_alb, d_rough, d_occ = new(P, 3), new(P, 3), new(P, 3)
        want_sh = ctx.has_shs and (ctx.needs_input_grad[9] or (rest is not None and ctx.needs_input_grad[14]) or M != 16)
        d_shs = new(P, M - (rest.shape[1] if rest is not None else 0), 3) if want_sh else None
        d_rest = new(P, rest.shape[1], 3) if (rest is not None and want_sh) else None
        if ctx.sink is not None and g_colors is not None:
            ctx.sink.collect(colors_fwd, g_colors, means3D)
        if ctx.rough_is_albedo:
            d_rough = d_alb   # one pointer: the kernel writes the sum; autograd gets it once (and None for the second use)
        acc_means = None
        if ctx.link is not None and ctx.link.means_grad is not None:
            acc_means, ctx.link.means_grad = ctx.link.means_grad, None
        with torch.cuda.device(dev):
            check(lib.gsr_frame_attributes_backward_acc(
                P, D, M, ptr(means3D), ptr(transforms), ptr(wn), ptr(scales), mod, ptr(rot_cov), ptr(rot_axis), ptr(albedo),
                ptr(roughness), ptr(occlusion), ptr(shs), ptr(rest), ptr(campos), ptr(view), ptr(c(g_cov)), ptr(g_colors),
                ptr(c(g_features)), ptr(d_means), ptr(d_T), ptr(d_wn), ptr(d_scales), ptr(d_rc), ptr(d_ra), ptr(d_alb),
                ptr(d_rough), ptr(d_occ), ptr(d_shs), ptr(d_rest), ptr(acc_means), torch.cuda.current_stream(dev).cuda_stream),
                "gsr_frame_attributes_backward")
        if ctx.park_rot:
            ctx.link.rot_grad, d_rc = d_rc, None
        return (d_means, d_T.view(t_shape), d_wn, d_scales, d_rc, d_ra, d_alb, None if ctx.rough_is_albedo else d_rough, d_occ, d_shs,
                None, None, None, None, d_rest)


def frame_attributes(means3D, transforms, world_normals, scales, scale_modifier, rot_cov, rot_axis, albedo, roughness, occlusion,
                     shs, sh_degree, campos, viewmatrix):
    """HIP path.  means3D [P,3], transforms [P,3,3], world_normals [P,3] (un-normalised), scales [P,3] (activated),
    rot_cov / rot_axis [P,4], albedo / roughness / occlusion [P,3], shs [P,M,3], None, or the model's two parameter tensors
    (features_dc [P,1,3], features_rest [P,15,3]) as a tuple: they are then read in place, without the torch.cat of
    get_features."""
    rest = None
    if isinstance(shs, (tuple, list)):
        shs, rest = shs
        if shs.shape[1] != 1 or rest.shape[1] != 15:
            shs, rest = torch.cat((shs, rest), dim=1), None
    cov3D, colors, features = _FrameAttributes.apply(means3D, transforms, world_normals, scales, rot_cov, rot_axis, albedo,
                                                     roughness, occlusion, shs, campos, viewmatrix, scale_modifier, sh_degree, rest)
    return cov3D, (colors if shs is not None else None), features
