"""Image losses of the reference training loop (utils/loss_utils.py:20-66, used at train.py:261-287).

    l1_loss, l2_loss       plain tensor expressions (identical to the reference)
    ssim(img1, img2)       fused HIP forward + backward (csrc/ssim.hip) behind the reference's signature; gradient flows to
                           img1 (the rendering), img2 is treated as ground truth
                           (the grouped-conv2d formulation it replaces is kept with the tests: tests/torch_reference.py)
"""
import torch

from ._lib import check, lib, ptr


def l1_loss(network_output, gt):
    return torch.abs((network_output - gt)).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


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
    """utils/loss_utils.py:36-66.  img1, img2: [..., C, H, W] on the GPU; the kernel is built for the reference's 11 x 11
    window (its only call sites, train.py:264,287).  No CPU / torch fallback."""
    if not img1.is_cuda:
        raise RuntimeError("ssim: tensors must live on a HIP device (no CPU path)")
    if window_size != 11:
        raise RuntimeError("ssim: only the reference's window_size = 11 is built")
    ssim_map = _SsimMap.apply(img1, img2)
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)
