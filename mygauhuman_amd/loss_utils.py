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


def gaussian(window_size, sigma):
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel):
    _1D_window = gaussian(window_size, 1.5).unsqueeze(1)
    _2D_window = _1D_window.mm(_1D_window.t()).float().unsqueeze(0).unsqueeze(0)
    return _2D_window.expand(channel, 1, window_size, window_size).contiguous()


def ssim_torch(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = create_window(window_size, channel).to(img1.device).type_as(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)


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
