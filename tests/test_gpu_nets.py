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


@pytest.fixture(params=["bf16x3", "f32"])
def precision(request):
    """both instruction choices of csrc/mlp.hip (nets.set_precision); the module default is restored afterwards"""
    from mygauhuman_amd import nets
    nets.set_precision(request.param)
    yield request.param
    nets.set_precision("bf16x3")


@pytest.mark.parametrize("P", [1, 31, 128, 129, 4097, 200_000])
def test_fused_offset_decoder_matches_float64_restatement(P, precision):
    from mygauhuman_amd.nets import FusedLBSOffsetDecoder
    torch.manual_seed(P)
    dec = FusedLBSOffsetDecoder().cuda()
    _randomise(dec, P)
    pts = (torch.rand(1, P, 3, device="cuda") * 2 - 1) * torch.tensor([0.45, 0.9, 0.15], device="cuda")
    with torch.no_grad():
        got = dec(pts)
        torch_path = dec.forward_torch(pts)
    want = _ref64(dec, pts)
    assert got.shape == (1, 24, P) and got.permute(0, 2, 1).is_contiguous()
    want = want.detach()
    scale = float(want.abs().max())
    # f32 MFMA = a k-ordered fmaf chain: the error against float64 is that of any f32 evaluation (sin / cos of arguments up to 512 rad
    # included: measured 6e-7 of the largest output); the bf16 instruction with both operands split in two terms drops 2^-16 of every
    # product: measured 7e-6; the torch-op path of the module is held to the same bound
    assert float((got.double() - want).abs().max()) <= (2e-6 if precision == "f32" else 2e-5) * scale, float((got.double() - want).abs().max()) / scale
    assert float((torch_path.double() - want).abs().max()) <= 2e-5 * scale


def _randomise(dec, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for p in dec.parameters():      # asymmetric, non-tiny weights and biases in every layer (a swapped row / column must show)
            p.copy_(torch.randn(p.shape, device="cuda", generator=g) * (0.5 / np.sqrt(p.shape[1] if p.dim() > 1 else 4.0)))


def _fragile_points(dec, pts, margin):
    """points where some pre-activation of some layer lies within `margin` x (that layer's largest) of zero, in float64: a forward
    with an error of that size may take the other side of the ReLU there, and ONE such flip moves a whole row of a weight gradient
    by O(1) -- the analogue of the rasterizer tests' fragile pixels.  Their dL/dout is set to zero, which silences them exactly."""
    x = pts[0].double()
    parts = [x]
    for f in 2.0 ** torch.arange(10, dtype=torch.float64, device=x.device):
        parts += [torch.sin(x * f), torch.cos(x * f)]
    emb = torch.cat(parts, dim=1)
    net, frag = emb, torch.zeros(x.shape[0], dtype=torch.bool, device=x.device)
    for i, m in enumerate(dec.bw_linears):
        z = net @ m.weight[:, :, 0].double().t() + m.bias.double()
        frag |= (z.abs() < margin * z.abs().max()).any(dim=1)
        net = torch.relu(z)
        if i == 2:
            net = torch.cat((emb, net), dim=1)
    return frag


@pytest.mark.parametrize("P", [1, 100, 128, 1025, 3000, 70_000])
def test_fused_offset_decoder_parameter_gradients_match_float64_autograd(P, precision):
    """backward = the forward again + dh = W^T dZ chained through the accumulator tiles + weight gradients as products over the points
    (csrc/mlp.hip), against torch autograd of the float64 restatement; P around the 256-point workgroups and the 1,024-point chunks."""
    from mygauhuman_amd.nets import FusedLBSOffsetDecoder
    dec = FusedLBSOffsetDecoder().cuda()
    _randomise(dec, P)
    g = torch.Generator(device="cuda").manual_seed(P + 1)
    pts = (torch.rand(1, P, 3, device="cuda", generator=g) * 2 - 1) * torch.tensor([0.45, 0.9, 0.15], device="cuda")
    w = torch.randn(1, 24, P, device="cuda", generator=g)          # dL/dout: asymmetric over outputs and points
    frag = _fragile_points(dec, pts, 1e-4)
    assert int(frag.sum()) <= max(1, P // 5)                       # (a few per cent of the points)
    w[:, :, frag] = 0.0
    out = dec(pts)
    assert out.grad_fn is not None and type(out.grad_fn).__name__ != "AddmmBackward0"
    (out * w).sum().backward()
    got = [p.grad.clone() for p in dec.parameters()]
    dec64 = FusedLBSOffsetDecoder().cuda().double()
    dec64.load_state_dict({k: v.double() for k, v in dec.state_dict().items()})
    (dec64.forward_torch(pts.double()) * w.double()).sum().backward()
    for (name, _), a, b in zip(dec.named_parameters(), got, [p.grad for p in dec64.parameters()]):
        scale = float(b.abs().max())
        if P == 1 and scale == 0.0:
            continue                                               # (the one point was fragile)
        err = float((a.double() - b).abs().max()) / scale
        # off the fragile points: f32 sums in arbitrary atomic order, measured 2e-7 .. 1e-6 (f32 instruction: bound 1e-5); the bf16
        # instruction with both operands split in two terms adds 2^-16 of every product in dh, measured <= 7e-6 (bound 5e-5)
        assert err <= (1e-5 if precision == "f32" else 5e-5), (name, err)


def test_fused_offset_decoder_repacks_after_a_parameter_update_and_trains():
    from mygauhuman_amd.nets import FusedLBSOffsetDecoder
    torch.manual_seed(0)
    dec = FusedLBSOffsetDecoder().cuda()
    pts = torch.rand(1, 1000, 3, device="cuda") - 0.5
    with torch.no_grad():
        a = dec(pts).clone()
        dec.bw_fc.bias.add_(1.0)          # an optimizer step changes the parameters in place
        b = dec(pts)
    assert torch.allclose(b, a + 1.0, atol=1e-5)
    opt = torch.optim.SGD(dec.parameters(), lr=1e-2)
    target = torch.randn(1, 24, 1000, device="cuda")
    losses = []
    for _ in range(20):                    # the fused forward + backward inside an ordinary training loop
        opt.zero_grad()
        loss = (dec(pts) - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] and all(np.isfinite(losses))
    with torch.no_grad():                  # and a pts that requires grad takes the torch ops (input gradient not built)
        ref = dec(pts)
    q = pts.clone().requires_grad_(True)
    out = dec(q)
    out.sum().backward()
    assert q.grad is not None and torch.allclose(out.detach(), ref, atol=2e-5 * float(ref.abs().max()))


def test_fused_offset_decoder_loads_a_reference_shaped_state_dict():
    from mygauhuman_amd.nets import FusedLBSOffsetDecoder
    dec = FusedLBSOffsetDecoder()
    shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    assert shapes == {"bw_linears.0.weight": (128, 63, 1), "bw_linears.0.bias": (128,), "bw_linears.1.weight": (128, 128, 1),
                      "bw_linears.1.bias": (128,), "bw_linears.2.weight": (128, 128, 1), "bw_linears.2.bias": (128,),
                      "bw_linears.3.weight": (128, 191, 1), "bw_linears.3.bias": (128,), "bw_fc.weight": (24, 128, 1),
                      "bw_fc.bias": (24,)}


def test_render_with_the_fused_offset_network_equals_render_with_its_torch_ops():
    """render() with motion_offset_flag on (gaussian_renderer/__init__.py:100-106: lbs_weights = pc.lweight_offset_decoder(means3D)):
    the network on the fused kernels against the same module in torch ops -- same images, same gradients of the network's and the
    model's parameters.  Gradients are compared on the f32 instruction (the two evaluations then agree to 5e-7 and take the same
    side of every ReLU; with the bf16 instruction's 7e-6 an occasional pre-activation near zero does not, which moves single rows of
    the weight gradients -- test_fused_offset_decoder_parameter_gradients... pins that path off such points); images on both."""
    import types
    from mygauhuman_amd import human_synth, nets
    from mygauhuman_amd.gaussian_renderer import render
    res = {}
    try:
        for mode in ("torch", "f32", "bf16x3"):
            nets.set_precision("bf16x3" if mode == "bf16x3" else "f32")
            model, body = human_synth.build(6000, 1500, "cuda", seed=3, motion=True, decoder="reference_size")
            model.lweight_offset_decoder.use_fused = mode != "torch"
            cam = human_synth.view_camera(body, 160, 128, 0, n_views=8, device="cuda")
            pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
            o = render(1, cam, model, pipe, torch.zeros(3, device="cuda"))
            (o["render"].mean() + 0.5 * o["render_alpha"].mean() + o["normal"].mean()).backward()
            net = list(model.lweight_offset_decoder.parameters())
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net)
            res[mode] = ([o[k].detach() for k in ("render", "render_alpha", "normal")],
                         [p.grad.clone() for p in net] + [p.grad.clone() for p in model.parameters() if p.grad is not None])
    finally:
        nets.set_precision("bf16x3")
    for mode in ("f32", "bf16x3"):
        for a, b in zip(res[mode][0], res["torch"][0]):
            d = (a - b).abs()
            if mode == "f32":
                assert float(d.max()) <= 2e-5
            else:
                # offsets 8e-6 apart move Gaussians by that much: a pixel whose cut-off test (alpha >= 1/255) sits within that of
                # the threshold gains or loses one contribution -- the rasterizer tests' fragile pixels; a handful of them, the rest 1e-4
                assert int((d > 1e-4).sum()) <= 8 and float(d.max()) <= 2e-2, (int((d > 1e-4).sum()), float(d.max()))
    for a, b in zip(res["f32"][1], res["torch"][1]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale + 1e-12, float((a - b).abs().max()) / max(scale, 1e-30)
