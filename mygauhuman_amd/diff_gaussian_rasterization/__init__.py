"""Operator API of the rasterizer -- same names, argument order, return values and error behaviour as the
reference's `diff_gaussian_rasterization` package (DGR/diff_gaussian_rasterization/__init__.py:21-223), backed by
the MI355X HIP library instead of the CUDA extension.

    GaussianRasterizationSettings   (:160-172)  NamedTuple, 12 fields, same order
    GaussianRasterizer              (:174-223)  nn.Module: forward(...) -> (color, radii, depth, alpha); markVisible
    rasterize_gaussians             (:21-42)    functional form
    _C                              (DGR/ext.cpp:15-19) raw bindings used by baking.py:259-282
"""
import os
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C
from .. import gradlink
from .._lib import N_EXTRA

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_gaussians_multi"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """CPU copy of an argument tuple, taken before the call so a crash cannot corrupt it (:17-19)."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_guarded(fn, args, debug, dump_name, phase):
    """Run a raw binding; in debug mode dump the inputs to `dump_name` if it raises (:83-90, :135-142)."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print(f"\nAn error occured in {phase}. Please forward {dump_name} for debugging.")
        raise


# GaussianRasterizer(...)(...) -- the call train.py / render.py make through gaussian_renderer -- never hands num_rendered to its caller
# (:215-223 returns colour, radii, depth, alpha), so it does not have to WAIT for it either: with SYNC_FREE (default on; environment
# GSR_SYNC_FREE_RASTER=0 or this attribute = False restore the reference's blocking read, CR/rasterizer_impl.cu:283) the module call
# sizes the binning buffer generously (_C.AsyncCapacity), keeps R on the device and examines the overflow flag at the end of the
# backward / at a later call / for a frame without backward at the end of the call itself: never silently.  The raw binding
# _C.rasterize_gaussians keeps the reference's return tuple and therefore its blocking read.
SYNC_FREE = os.environ.get("GSR_SYNC_FREE_RASTER", "1") != "0"


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                sync_free=False, will_backward=True):
        rs = raster_settings
        # the binding's positional order (DGR/rasterize_points.h:19-39)
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        # (GSR_FWD_ZERO_ROWS -- the forward preprocess zeroing the gradient rows so that the backward needs no fill kernel -- is
        # built and measured SLOWER in the render() frame: preprocess forward 9.7 -> 19.2 us and backward 18.2 -> 28.7 us (it then
        # clears the rows it consumed) against one 5 us fill kernel; the wrappers do not use it)
        ctx.rows_zeroed = False
        ctx.watch = None
        if sync_free:
            out = _call_guarded(lambda *x: _C.rasterize_gaussians_async(*x, rows_zeroed=ctx.rows_zeroed), args, rs.debug,
                                "snapshot_fw.dump", "forward")
            num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer = out[:8]
            if will_backward:
                ctx.watch = out[9]   # examined at the end of backward(), before an optimizer can consume the gradients
            elif out[9] is not None:
                _C.AsyncCapacity.check(out[9])   # a frame nobody will backpropagate through: raise on THIS call
        else:
            num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer = _call_guarded(
                lambda *x: _C.rasterize_gaussians(*x, rows_zeroed=ctx.rows_zeroed), args, rs.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, alpha)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer,
         alpha) = ctx.saved_tensors
        if ctx.watch is not None:
            _C.AsyncCapacity.poll(means3D.device)
        # DGR/rasterize_points.h:41-66
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, grad_alpha, sh,
                rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, alpha, rs.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _call_guarded(lambda *x: _C.rasterize_gaussians_backward(*x, rows_zeroed=ctx.rows_zeroed), args, rs.debug,
                                         "snapshot_bw.dump", "backward")
        if ctx.watch is not None:   # the whole backward is queued: wait for the FORWARD's overflow flag only
            _C.AsyncCapacity.check(ctx.watch)
        # gradients in the order of forward()'s inputs (:146-156); raster_settings gets None
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, sync_free=None):
    """sync_free: None = the module default SYNC_FREE."""
    sync_free = SYNC_FREE if sync_free is None else bool(sync_free)
    # decided HERE, in the caller's grad mode (inside Function.forward grad mode is always off)
    will_backward = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                                                  cov3Ds_precomp))
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, sync_free, will_backward)


_ZERO_IMAGES = {}


def _zero_image(planes, H, W, device):
    """A zero gradient image nobody writes to (the backward kernels only read their image gradients)."""
    key = (planes, H, W, device)
    z = _ZERO_IMAGES.get(key)
    if z is None:
        z = torch.zeros((planes, H, W), dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():  # memory of a graph's private pool must not outlive the graph
            _ZERO_IMAGES[key] = z   # (kept for good: a captured graph may hold its address)
    return z


class _RasterizeGaussiansMulti(torch.autograd.Function):
    """Extension: ONE preprocess + binning + blend for the main colour and 18 extra feature
    channels instead of seven rasterizer calls with identical geometry (gaussian_renderer/__init__.py:203-272)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, extra, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                sync_free=False, will_backward=True, loss_spec=None):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered,
                rs.debug)
        ctx.watch = None
        if sync_free:  # no host read of num_rendered: generous capacity + deferred overflow check (_C.AsyncCapacity)
            out = _C.rasterize_gaussians_async(*args, extra=extra)
            watch, out = out[9], out[:9]
            # a frame that will have a backward is examined there, before the optimizer can consume its gradients.  A frame
            # that will NOT (evaluation under no_grad, render.py-style loops) has nobody to examine a deferred flag: it waits for
            # its own flag words here, with the whole frame already queued -- an overflow raises on this very call, like the
            # reference's blocking path, which sizes the buffer and always renders
            if will_backward:
                ctx.watch = watch
            elif watch is not None:
                _C.AsyncCapacity.check(watch)
        else:
            out = _C.rasterize_gaussians(*args, extra=extra)
        ctx.rows_zeroed = False
        num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer, out_extra = out
        ctx.set_materialize_grads(False)  # untouched images arrive as None in backward, not as zero tensors
        # (gradlink) render(): the attribute kernel of this frame consumes the same positions and its backward runs after this one
        # (it waits for dL_dextra): the position gradient is parked for it to add in-kernel instead of autograd adding the two
        link = gradlink.current()
        ctx.park_means = link if (link is not None and link.attr_means_ptr is not None and link.attr_means_ptr == means3D.data_ptr()
                                  and ctx.needs_input_grad[0] and ctx.needs_input_grad[4]) else None
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # fused phase-1 training loss (_C.Phase1Loss): its value is one more output; the backward forms its image gradients in-kernel
        ctx.loss_spec, ctx.loss_stats, loss_out = loss_spec, None, ()
        if loss_spec is not None:
            loss, ctx.loss_stats = _C.phase1_loss_forward(loss_spec, color, alpha, out_extra)
            loss_out = (loss,)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
                              imgBuffer, alpha, extra, *((color, out_extra) if loss_spec is not None else ()))
        # six separate outputs (views of one [18,H,W] buffer): autograd then hands back one gradient per image -- None for
        # images the loss never touched -- instead of materialising a full 18-plane gradient per slice
        return (color, radii, depth, alpha) + tuple(out_extra[3 * i:3 * i + 3] for i in range(N_EXTRA // 3)) + loss_out

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_alpha, *grad_feats):
        rs = ctx.raster_settings
        grad_loss = None
        if ctx.loss_spec is not None:
            grad_feats, grad_loss = grad_feats[:-1], grad_feats[-1]
        if ctx.watch is not None:
            _C.AsyncCapacity.poll(ctx.saved_tensors[1].device)  # non-blocking here: waiting now would keep the backward kernels off the queue
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer, alpha,
         extra) = ctx.saved_tensors[:12]
        H, W = alpha.shape[-2], alpha.shape[-1]
        phase1 = None
        if grad_loss is not None:   # the fused loss took part in what is being differentiated: images without a further gradient stay None
            phase1 = (ctx.loss_spec, ctx.loss_stats, grad_loss, ctx.saved_tensors[12], ctx.saved_tensors[13])
        else:
            # images the loss never touched: one shared read-only zero image per (size, device) instead of a fill per backward
            grad_out_color = _zero_image(3, H, W, alpha.device) if grad_out_color is None else grad_out_color
            grad_depth = _zero_image(1, H, W, alpha.device) if grad_depth is None else grad_depth
            grad_alpha = _zero_image(1, H, W, alpha.device) if grad_alpha is None else grad_alpha
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_extra_in) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, grad_alpha, sh, rs.sh_degree, rs.campos,
            geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, alpha, rs.debug, out={"lean": True}, extra=extra,
            dL_dout_extra=list(grad_feats), rows_zeroed=ctx.rows_zeroed, phase1=phase1)
        if ctx.watch is not None:
            # the whole backward is queued; wait for the FORWARD's overflow flag only (the GPU stays busy with the backward)
            # so that an overflow raises before the optimizer consumes these gradients
            _C.AsyncCapacity.check(ctx.watch)
        if ctx.park_means is not None:
            ctx.park_means.means_grad, grad_means3D = grad_means3D, None
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_extra_in, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None, None, None, None)


def rasterize_gaussians_multi(means3D, means2D, sh, colors_precomp, extra_colors, opacities, scales, rotations, cov3Ds_precomp,
                              raster_settings, sync_free=False, loss_spec=None):
    """Blend the main colour (SHs or colors_precomp) and up to six extra [P,3] colour sets (a list, or one packed [P,18]
    tensor) in one pass.
    Returns (color, radii, depth, alpha, [image_i [3,H,W] for each extra colour set])."""
    P = means3D.shape[0]
    if isinstance(extra_colors, torch.Tensor):  # already packed: [P, 18] = six RGB triples side by side
        if extra_colors.dim() != 2 or extra_colors.shape[1] != N_EXTRA:
            raise Exception("rasterize_gaussians_multi: a packed extra-colour tensor must be [P, 18]")
        n, extra = 6, extra_colors
    else:
        n = len(extra_colors)
        if not 1 <= n <= 6:
            raise Exception("rasterize_gaussians_multi takes 1 to 6 extra colour sets")
        cols = list(extra_colors) + [torch.zeros((P, 3), dtype=means3D.dtype, device=means3D.device)] * (6 - n)
        extra = torch.cat(cols, dim=1)
    # decided HERE, in the caller's grad mode (inside Function.forward grad mode is always off and ctx.needs_input_grad ignores it)
    will_backward = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (means3D, means2D, sh, colors_precomp, extra, opacities, scales,
                                                                  rotations, cov3Ds_precomp))
    if loss_spec is not None and n != 6:
        raise Exception("rasterize_gaussians_multi: the fused phase-1 loss needs all six extra colour sets (normal and axis among them)")
    out = _RasterizeGaussiansMulti.apply(means3D, means2D, sh, colors_precomp, extra, opacities, scales, rotations, cov3Ds_precomp,
                                         raster_settings, sync_free, will_backward, loss_spec)
    if loss_spec is not None:
        return out[0], out[1], out[2], out[3], list(out[4:4 + n]), out[4 + N_EXTRA // 3]
    return out[0], out[1], out[2], out[3], list(out[4:4 + n])


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of the points in front of the near plane of this camera (:179-188)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])  # CPU, zero elements -> null pointer -> the other input mode (:200-210)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)

    def forward_multi(self, means3D, means2D, opacities, extra_colors, shs=None, colors_precomp=None, scales=None, rotations=None,
                      cov3D_precomp=None, sync_free=False, loss_spec=None):
        """Extension: like forward(), plus `extra_colors` (list of 1..6 [P,3] tensors) blended in the same pass.
        Returns (color, radii, depth, alpha, [extra images]).  sync_free=True skips the host read of num_rendered
        (_C.AsyncCapacity: generous binning capacity, overflow reported at backward / next call).
        loss_spec (a _C.Phase1Loss): the phase-1 training loss evaluated fused with the pass; its value is appended to the result."""
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        return rasterize_gaussians_multi(means3D, means2D, empty if shs is None else shs,
                                         empty if colors_precomp is None else colors_precomp, extra_colors, opacities,
                                         empty if scales is None else scales, empty if rotations is None else rotations,
                                         empty if cov3D_precomp is None else cov3D_precomp, self.raster_settings, sync_free, loss_spec)
