"""The fused forward of the skinning-weight offset network (csrc/mlp.hip, mygauhuman_amd/nets.py) against the same network in
float64 torch ops (the arithmetic of nets/mlp_delta_weight_lbs.py:5-32,34-77 restated: the reference holds no fixture for it --
parity unpinned -- but the module's parameter names / shapes are the reference's, so its state_dict loads)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref64(dec, pts):
    x = pts[0].double()
    freqs = 2.0 ** torch.arange(10, dtype=torch.float64, device=x.device)
    parts = [x]
    for f in freqs:                      # (x, sin(f0 x), cos(f0 x), sin(f1 x), ...): get_embedder(10)
        parts += [torch.sin(x * f), torch.cos(x * f)]
    emb = torch.cat(parts, dim=1)
    net = emb
    for i, m in enumerate(dec.bw_linears):
        net = torch.relu(net @ m.weight[:, :, 0].double().t() + m.bias.double())
        if i == 2:
            net = torch.cat((emb, net), dim=1)
    return (net @ dec.bw_fc.weight[:, :, 0].double().t() + dec.bw_fc.bias.double()).t()[None]


@pytest.mark.parametrize("P", [1, 31, 128, 129, 4097, 200_000])
def test_fused_offset_decoder_matches_float64_restatement(P):
    from mygauhuman_amd.nets import FusedLBSOffsetDecoder
    torch.manual_seed(P)
    dec = FusedLBSOffsetDecoder().cuda()
    with torch.no_grad():
        for p in dec.parameters():      # asymmetric, non-tiny weights and biases in every layer (a swapped row / column must show)
            p.copy_(torch.randn_like(p) * (0.5 / np.sqrt(p.shape[1] if p.dim() > 1 else 4.0)))
    pts = (torch.rand(1, P, 3, device="cuda") * 2 - 1) * torch.tensor([0.45, 0.9, 0.15], device="cuda")
    with torch.no_grad():
        got = dec(pts)
        torch_path = dec.forward_torch(pts)
    want = _ref64(dec, pts)
    assert got.shape == (1, 24, P) and got.permute(0, 2, 1).is_contiguous()
    scale = float(want.abs().max())
    # f32 MFMA = a k-ordered fmaf chain: the error against float64 is that of any f32 evaluation (sin / cos of arguments up to 512 rad
    # included); the torch-op path of the module is held to the same bound
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale, float((got.double() - want).abs().max()) / scale
    assert float((torch_path.double() - want).abs().max()) <= 2e-5 * scale


def test_fused_offset_decoder_repacks_after_a_parameter_update_and_keeps_autograd():
    from mygauhuman_amd.nets import FusedLBSOffsetDecoder
    torch.manual_seed(0)
    dec = FusedLBSOffsetDecoder().cuda()
    pts = torch.rand(1, 1000, 3, device="cuda") - 0.5
    with torch.no_grad():
        a = dec(pts).clone()
        dec.bw_fc.bias.add_(1.0)          # an optimizer step changes the parameters in place
        b = dec(pts)
    assert torch.allclose(b, a + 1.0, atol=1e-5)
    out = dec(pts)                         # gradients recorded: the torch-op path, differentiable
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in dec.parameters())
    with torch.no_grad():
        assert torch.allclose(dec(pts), out.detach(), atol=2e-5 * float(out.abs().max()))


def test_fused_offset_decoder_loads_a_reference_shaped_state_dict():
    from mygauhuman_amd.nets import FusedLBSOffsetDecoder
    dec = FusedLBSOffsetDecoder()
    shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    assert shapes == {"bw_linears.0.weight": (128, 63, 1), "bw_linears.0.bias": (128,), "bw_linears.1.weight": (128, 128, 1),
                      "bw_linears.1.bias": (128,), "bw_linears.2.weight": (128, 128, 1), "bw_linears.2.bias": (128,),
                      "bw_linears.3.weight": (128, 191, 1), "bw_linears.3.bias": (128,), "bw_fc.weight": (24, 128, 1),
                      "bw_fc.bias": (24,)}
