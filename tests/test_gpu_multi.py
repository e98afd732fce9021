"""Fused multi-feature rasterisation against the reference's structure: seven separate
rasterizer calls with identical geometry.  Images must agree to rounding, gradients to 1e-4."""
import types

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _settings(cam, bg, deg):
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam["H"], image_width=cam["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=util.to_dev(bg),
        scale_modifier=1.0, viewmatrix=util.to_dev(cam["viewmatrix"]), projmatrix=util.to_dev(cam["projmatrix"]), sh_degree=deg,
        campos=util.to_dev(cam["campos"]), prefiltered=False, debug=False)


@pytest.fixture(params=[0, 1], ids=["quadrant_waves", "block_waves_4x"])
def layout(request):
    """Options::blend_layout: 0 = a wave per 8x8 quadrant, 1 = a wave per 4x4 block blending four survivors per step."""
    from mygauhuman_amd import _lib
    util.skip_unless_experiments(request.param == 1)
    _lib.set_tuning("blend_layout", request.param)
    yield request.param
    _lib.set_tuning("blend_layout", _lib.DEFAULT_BLEND_LAYOUT)


@pytest.mark.parametrize("mode", ["sh", "precomp"])
@pytest.mark.parametrize("n_extra", [1, 6])
def test_forward_multi_equals_separate_passes(mode, n_extra, layout):
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizer
    P, W, H = 5000, 150, 100
    cam, g = util.make_scene(P, W, H, 21, 3, 0.03, 0.02)
    bg = np.array([0.3, 0.6, 0.1], np.float32)
    rng = np.random.default_rng(4)
    rast = GaussianRasterizer(_settings(cam, bg, 3))

    def leaves():
        t = {k: util.to_dev(v).requires_grad_(True) for k, v in g.items() if isinstance(v, np.ndarray)}
        t["extras"] = [util.to_dev(rng.uniform(0, 1, (P, 3)).astype(np.float32)).requires_grad_(True) for _ in range(n_extra)]
        t["means2D"] = torch.zeros((P, 3), device="cuda", requires_grad=True)
        return t

    rng = np.random.default_rng(4)
    a = leaves()
    rng = np.random.default_rng(4)
    b = leaves()
    if mode == "sh":
        kw = lambda t: dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])  # noqa: E731
    else:
        kw = lambda t: dict(colors_precomp=t["colors"], cov3D_precomp=t["cov3D"])  # noqa: E731
    # fused
    color, radii, depth, alpha, feats = rast.forward_multi(means3D=a["means3D"], means2D=a["means2D"], opacities=a["opacities"],
                                                           extra_colors=a["extras"], **kw(a))
    # separate (reference structure)
    c0, r0, d0, a0 = rast(means3D=b["means3D"], means2D=b["means2D"], opacities=b["opacities"], **kw(b))
    kwb = kw(b)
    geo = {k: v for k, v in kwb.items() if k in ("scales", "rotations", "cov3D_precomp")}
    sep = [rast(means3D=b["means3D"], means2D=b["means2D"], opacities=b["opacities"], colors_precomp=e, **geo)[0] for e in b["extras"]]
    assert torch.equal(radii, r0)
    np.testing.assert_allclose(color.detach().cpu().numpy(), c0.detach().cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(alpha.detach().cpu().numpy(), a0.detach().cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(depth.detach().cpu().numpy(), d0.detach().cpu().numpy(), atol=2e-4)
    for f, s in zip(feats, sep):
        np.testing.assert_allclose(f.detach().cpu().numpy(), s.detach().cpu().numpy(), atol=2e-5)
    wr = np.random.default_rng(9)
    ws = [util.to_dev(wr.normal(0, 1, (3, H, W)).astype(np.float32)) for _ in range(n_extra + 1)]
    wa = util.to_dev(wr.normal(0, 1, (1, H, W)).astype(np.float32))
    la = (color * ws[0]).sum() + (alpha * wa).sum() + sum((f * w).sum() for f, w in zip(feats, ws[1:]))
    lb = (c0 * ws[0]).sum() + (a0 * wa).sum() + sum((s * w).sum() for s, w in zip(sep, ws[1:]))
    la.backward()
    lb.backward()
    keys = ["means3D", "means2D", "opacities"] + (["shs", "scales", "rotations"] if mode == "sh" else ["colors", "cov3D"])
    for k in keys:
        util.assert_close(k, a[k].grad.cpu().numpy(), b[k].grad.cpu().numpy(), tol=1e-4, max_bad_frac=1e-4)
    for ea, eb in zip(a["extras"], b["extras"]):
        util.assert_close("extra colour grad", ea.grad.cpu().numpy(), eb.grad.cpu().numpy(), tol=1e-4, max_bad_frac=1e-4)


@pytest.mark.parametrize("used", [(0, 5), (2,), (3, 4), (), (1, 2, 3, 4, 5)])
@pytest.mark.parametrize("with_color", [True, False])
def test_forward_multi_partial_loss_masks_untouched_images(used, with_color):
    """Only some of the seven images enter the loss: autograd hands None for the others, the backward kernel gets a group
    mask and must produce the same gradients as separate passes (zeros for the untouched colour sets)."""
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizer
    P, W, H = 4000, 130, 90
    cam, g = util.make_scene(P, W, H, 33, 3, 0.03, 0.02)
    bg = np.array([0.2, 0.5, 0.7], np.float32)
    rast = GaussianRasterizer(_settings(cam, bg, 3))

    def leaves():
        rng = np.random.default_rng(8)
        t = {k: util.to_dev(v).requires_grad_(True) for k, v in g.items() if isinstance(v, np.ndarray)}
        t["extras"] = [util.to_dev(rng.uniform(0, 1, (P, 3)).astype(np.float32)).requires_grad_(True) for _ in range(6)]
        t["means2D"] = torch.zeros((P, 3), device="cuda", requires_grad=True)
        return t

    a, b = leaves(), leaves()
    kw = lambda t: dict(colors_precomp=t["colors"], cov3D_precomp=t["cov3D"])  # noqa: E731
    color, radii, depth, alpha, feats = rast.forward_multi(means3D=a["means3D"], means2D=a["means2D"], opacities=a["opacities"],
                                                           extra_colors=a["extras"], **kw(a))
    c0, r0, d0, a0 = rast(means3D=b["means3D"], means2D=b["means2D"], opacities=b["opacities"], **kw(b))
    sep = [rast(means3D=b["means3D"], means2D=b["means2D"], opacities=b["opacities"], colors_precomp=e, cov3D_precomp=b["cov3D"])[0]
           for e in b["extras"]]
    wr = np.random.default_rng(9)
    ws = [util.to_dev(wr.normal(0, 1, (3, H, W)).astype(np.float32)) for _ in range(7)]
    la = (depth * 0.0).sum() + sum((feats[i] * ws[1 + i]).sum() for i in used)
    lb = (d0 * 0.0).sum() + sum((sep[i] * ws[1 + i]).sum() for i in used)
    if with_color:
        la = la + (color * ws[0]).sum()
        lb = lb + (c0 * ws[0]).sum()
    la.backward()
    lb.backward()
    for k in ("means3D", "means2D", "opacities", "colors", "cov3D"):
        gb = b[k].grad if b[k].grad is not None else torch.zeros_like(b[k])
        util.assert_close(k, a[k].grad.cpu().numpy(), gb.cpu().numpy(), tol=1e-4, max_bad_frac=1e-4)
    for i, (ea, eb) in enumerate(zip(a["extras"], b["extras"])):
        if i in used:
            util.assert_close("extra colour grad", ea.grad.cpu().numpy(), eb.grad.cpu().numpy(), tol=1e-4, max_bad_frac=1e-4)
        else:
            assert float(ea.grad.abs().max()) == 0.0


def test_render_fused_equals_seven_passes(oracle):
    from mygauhuman_amd.gaussian_renderer import render
    from tests.test_gpu_render import _human_scene
    outs, grads = {}, {}
    for sep in (False, True):
        s = _human_scene(oracle, seed=2)
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, separate_feature_passes=sep)
        o = render(1, s.cam, s.model, pipe, util.to_dev(np.array([0.2, 0.3, 0.4], np.float32)))
        loss = sum(o[k].mean() * (i + 1) for i, k in enumerate(("render", "normal", "albedo", "occlusion", "roughness", "world_normal",
                                                                 "render_axis", "render_alpha")))
        loss.backward()
        outs[sep] = {k: v.detach().cpu().numpy() for k, v in o.items() if isinstance(v, torch.Tensor)}
        grads[sep] = [p.grad.cpu().numpy() for p in s.model.parameters()] + [o["viewspace_points"].grad.cpu().numpy()]
    for k in ("render", "render_depth", "render_alpha", "normal", "albedo", "occlusion", "roughness", "world_normal", "render_axis"):
        np.testing.assert_allclose(outs[False][k], outs[True][k], atol=3e-5, err_msg=k)
    np.testing.assert_array_equal(outs[False]["radii"], outs[True]["radii"])
    for ga, gb in zip(grads[False], grads[True]):
        util.assert_close("render grads", ga, gb, tol=1e-4, max_bad_frac=2e-4)


def test_sync_free_forward_multi_equals_blocking_and_reports_overflow(monkeypatch):
    """sync_free=True (no host read of num_rendered) must give the same images and gradients; an undersized binning capacity
    must surface as a RuntimeError at backward / at the explicit check, never silently."""
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizer, _C
    P, W, H = 4000, 130, 90
    cam, g = util.make_scene(P, W, H, 41, 3, 0.03, 0.02)
    bg = np.array([0.2, 0.5, 0.7], np.float32)
    rast = GaussianRasterizer(_settings(cam, bg, 3))

    def leaves():
        rng = np.random.default_rng(8)
        t = {k: util.to_dev(v).requires_grad_(True) for k, v in g.items() if isinstance(v, np.ndarray)}
        t["extras"] = [util.to_dev(rng.uniform(0, 1, (P, 3)).astype(np.float32)).requires_grad_(True) for _ in range(6)]
        t["means2D"] = torch.zeros((P, 3), device="cuda", requires_grad=True)
        return t

    def run(t, sync_free):
        out = rast.forward_multi(means3D=t["means3D"], means2D=t["means2D"], opacities=t["opacities"], extra_colors=t["extras"],
                                 shs=t["shs"], scales=t["scales"], rotations=t["rotations"], sync_free=sync_free)
        (out[0].sum() + out[3].sum() + out[4][0].sum() + out[4][5].mean()).backward()
        return out

    a, b = leaves(), leaves()
    oa, ob = run(a, False), run(b, True)
    assert torch.equal(oa[1], ob[1])
    for x, y in zip([oa[0], oa[2], oa[3]] + list(oa[4]), [ob[0], ob[2], ob[3]] + list(ob[4])):
        assert torch.equal(x, y)
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
        util.assert_close(k, b[k].grad.cpu().numpy(), a[k].grad.cpu().numpy(), tol=2e-5, max_bad_frac=1e-4)
    _C.AsyncCapacity.check_all()
    st = _C.AsyncCapacity._dev('cuda')
    assert st['largest_R'] > 0 and not st['pending']
    # undersized capacity: the frame renders only the background and the error is raised at backward
    monkeypatch.setattr(_C.AsyncCapacity, "capacity", classmethod(lambda cls, n, device=None: 64))
    c = leaves()
    out = rast.forward_multi(means3D=c["means3D"], means2D=c["means2D"], opacities=c["opacities"], extra_colors=c["extras"],
                             shs=c["shs"], scales=c["scales"], rotations=c["rotations"], sync_free=True)
    with pytest.raises(RuntimeError, match="binning capacity"):
        out[0].sum().backward()
    assert float((out[0].detach() - util.to_dev(bg)[:, None, None]).abs().max()) == 0.0
    assert not _C.AsyncCapacity._dev('cuda')['pending']
