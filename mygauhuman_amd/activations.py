"""The parameter activations of one frame in one HIP kernel each way (csrc/activations.hip): what the reference reads through
the GaussianModel property getters (scene/gaussian_model.py:157-199) plus render()'s occlusion placeholder
(gaussian_renderer/__init__.py:141).

    opacity, albedo, scaling, rotation, normal, occlusion = frame_activations(_opacity, _albedo, _scaling, _rotation, _normal)

opacity [P,1] = sigmoid, albedo [P,3] = sigmoid (get_roughness reads _albedo too, :197-199: use `albedo` for it), scaling [P,3] =
exp, rotation [P,4] = F.normalize, normal [P,3] = x / |x|, occlusion [P,3] = opacity.repeat(1, 3).  Tensors must live on the GPU.
"""
import torch

from . import gradlink
from ._lib import check, lib, ptr


class _FrameActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, opacity_raw, albedo_raw, scaling_raw, rotation_raw, normal_raw):
        if not opacity_raw.is_cuda:
            raise RuntimeError("frame_activations: tensors must live on a HIP device (no CPU path)")
        dev, f32 = opacity_raw.device, torch.float32
        P = opacity_raw.shape[0]
        c = lambda t: t.detach().contiguous().float()  # noqa: E731
        raw = [c(opacity_raw), c(albedo_raw), c(scaling_raw), c(rotation_raw), c(normal_raw)]
        new = lambda *s: torch.empty(s, dtype=f32, device=dev)  # noqa: E731
        opacity, albedo, scaling, rotation, normal, occlusion = new(P, 1), new(P, 3), new(P, 3), new(P, 4), new(P, 3), new(P, 3)
        with torch.cuda.device(dev):
            check(lib.gsr_model_activations_forward(P, *[ptr(t) for t in raw], ptr(opacity), ptr(albedo), ptr(scaling), ptr(rotation),
                                                    ptr(normal), ptr(occlusion), torch.cuda.current_stream(dev).cuda_stream),
                  "gsr_model_activations_forward")
        ctx.save_for_backward(raw[3], raw[4], opacity, albedo, scaling)
        # (gradlink) this backward runs after the attribute kernel's: it can take that kernel's gradient of the RAW quaternion along
        ctx.link = gradlink.current() if (ctx.needs_input_grad[3] and rotation_raw.is_contiguous()
                                          and rotation_raw.dtype == f32) else None
        if ctx.link is not None:
            ctx.link.act_rot_in_ptr, ctx.link.act_rot_out_ptr = rotation_raw.data_ptr(), rotation.data_ptr()
        return opacity, albedo, scaling, rotation, normal, occlusion

    @staticmethod
    def backward(ctx, g_opacity, g_albedo, g_scaling, g_rotation, g_normal, g_occlusion):
        rotation_raw, normal_raw, opacity, albedo, scaling = ctx.saved_tensors
        dev, f32 = opacity.device, torch.float32
        P = opacity.shape[0]
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        gs = [c(g_opacity), c(g_albedo), c(g_scaling), c(g_rotation), c(g_normal), c(g_occlusion)]
        new = lambda *s: torch.empty(s, dtype=f32, device=dev)  # noqa: E731
        d_op, d_al, d_sc, d_ro, d_no = new(P, 1), new(P, 3), new(P, 3), new(P, 4), new(P, 3)
        acc = None
        if ctx.link is not None and ctx.link.rot_grad is not None:
            acc, ctx.link.rot_grad = ctx.link.rot_grad, None
        with torch.cuda.device(dev):
            check(lib.gsr_model_activations_backward_acc(P, ptr(rotation_raw), ptr(normal_raw), ptr(opacity), ptr(albedo),
                                                         ptr(scaling), *[ptr(g) for g in gs], ptr(d_op), ptr(d_al), ptr(d_sc),
                                                         ptr(d_ro), ptr(d_no), ptr(acc), torch.cuda.current_stream(dev).cuda_stream),
                  "gsr_model_activations_backward")
        return d_op, d_al, d_sc, d_ro, d_no


def frame_activations(opacity_raw, albedo_raw, scaling_raw, rotation_raw, normal_raw):
    return _FrameActivations.apply(opacity_raw, albedo_raw, scaling_raw, rotation_raw, normal_raw)
