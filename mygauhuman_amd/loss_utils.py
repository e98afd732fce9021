"""Image losses of the reference training loop (utils/loss_utils.py:20-66, used at train.py:261-287).

    l1_loss, l2_loss       plain tensor expressions (identical to the reference)
    ssim(img1, img2)       fused HIP forward + backward (csrc/ssim.hip) behind the reference's signature; gradient flows to
                           img1 (the rendering), img2 is treated as ground truth
    ssim_torch(img1, img2) the reference's grouped-conv2d formulation, kept as the fp32 reference of the kernel (tests) and for
                           CPU tensors / window sizes other than 11
"""
from math import exp

import torch
import torch.nn.functional as F

from ._lib import check, lib, ptr


def l1_loss(network_output, gt):
    return torch.abs((network_output - gt)).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def _window_2d(window_size, sigma=1.5):
    """Normalised 1-D Gaussian (python-float exp, float32 tensor, float32 normalisation -- the arithmetic of
    utils/loss_utils.py:25-27) and its outer product, the [ws, ws] window of create_window (:29-33)."""
    centre = window_size // 2
    taps = torch.tensor([exp(-((i - centre) ** 2) / (2.0 * sigma * sigma)) for i in range(window_size)], dtype=torch.float32)
    taps = taps / taps.sum()
    return torch.outer(taps, taps)


def ssim_torch(img1, img2, window_size=11, size_average=True):
    """SSIM as the reference formulates it (utils/loss_utils.py:36-66): depth-wise conv2d with the Gaussian window, zero padding."""
    C = img1.size(-3)
    win = _window_2d(window_size).to(device=img1.device, dtype=img1.dtype).expand(C, 1, window_size, window_size).contiguous()
    blur = lambda t: F.conv2d(t, win, padding=window_size // 2, groups=C)  # noqa: E731
    m1, m2 = blur(img1), blur(img2)
    m11, m22, m12 = m1 * m1, m2 * m2, m1 * m2
    v1, v2, v12 = blur(img1 * img1) - m11, blur(img2 * img2) - m22, blur(img1 * img2) - m12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * m12 + c1) * (2 * v12 + c2)) / ((m11 + m22 + c1) * (v1 + v2 + c2))
    return smap.mean() if size_average else smap.mean(1).mean(1).mean(1)


class _SsimMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        dev = img1.device
        a, b = img1.detach().contiguous().float(), img2.detach().contiguous().float()
        H, W = a.shape[-2], a.shape[-1]
        planes = a.numel() // (H * W) if H * W else 0
        out = torch.empty_like(a)
        need = ctx.needs_input_grad[0]
        dA, dB, dC = (torch.empty_like(a) for _ in range(3)) if need else (None, None, None)
        with torch.cuda.device(dev):
            check(lib.gsr_ssim_forward(planes, H, W, ptr(a), ptr(b), ptr(out), ptr(dA), ptr(dB), ptr(dC),
                                       torch.cuda.current_stream(dev).cuda_stream), "gsr_ssim_forward")
        if need:
            ctx.save_for_backward(a, b, dA, dB, dC)
        ctx.dims = (planes, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, dA, dB, dC = ctx.saved_tensors
        planes, H, W = ctx.dims
        g = g.contiguous().float()
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            check(lib.gsr_ssim_backward(planes, H, W, ptr(a), ptr(b), ptr(g), 0.0, ptr(dA), ptr(dB), ptr(dC), ptr(out),
                                        torch.cuda.current_stream(a.device).cuda_stream), "gsr_ssim_backward")
        return out, None


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:36-66.  img1, img2: [..., C, H, W] on the GPU."""
    if not img1.is_cuda or window_size != 11:
        return ssim_torch(img1, img2, window_size, size_average)
    ssim_map = _SsimMap.apply(img1, img2)
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)
