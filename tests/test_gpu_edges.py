"""Empty / degenerate inputs through every operator of the package: nothing may crash, shapes must be right."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _settings(W, H):
    from mygauhuman_amd import cameras
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    cam = cameras.make_camera(W, H, 50.0)
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=util.to_dev(np.array([0.1, 0.2, 0.3], np.float32)),
        scale_modifier=1.0, viewmatrix=util.to_dev(cam["viewmatrix"]), projmatrix=util.to_dev(cam["projmatrix"]), sh_degree=3,
        campos=util.to_dev(cam["campos"]), prefiltered=False, debug=False)


def test_zero_gaussians_everywhere():
    from mygauhuman_amd import attributes, knn_cuda, lbs
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizer
    from mygauhuman_amd.simple_knn._C import distCUDA2
    W, H = 64, 48
    rs = _settings(W, H)
    rast = GaussianRasterizer(rs)
    z = lambda *s: torch.zeros(s, device="cuda", requires_grad=True)  # noqa: E731
    means, cols, op, cov = z(0, 3), z(0, 3), z(0, 1), z(0, 6)
    color, radii, depth, alpha = rast(means3D=means, means2D=z(0, 3), opacities=op, colors_precomp=cols, cov3D_precomp=cov)
    assert color.shape == (3, H, W) and radii.shape == (0,)
    # P == 0 short-circuits before any kernel, outputs stay zero-filled like the reference's (DGR/rasterize_points.cu:69-84)
    assert float(color.abs().max()) == 0 and float(alpha.abs().max()) == 0
    color.sum().backward()
    out = rast.forward_multi(means3D=means, means2D=z(0, 3), opacities=op, extra_colors=[z(0, 3)], colors_precomp=cols, cov3D_precomp=cov)
    assert out[4][0].shape == (3, H, W)
    (out[0].sum() + out[4][0].sum()).backward()
    assert distCUDA2(torch.zeros((0, 3), device="cuda")).shape == (0,)
    d, i = knn_cuda.knn_self(torch.zeros((0, 3), device="cuda"), 2)
    assert d.shape == (0, 2) and i.shape == (0, 2)
    d, i = knn_cuda.knn_nearest(torch.rand((10, 3), device="cuda"), torch.zeros((0, 3), device="cuda"))
    assert d.shape == (0,)
    cam = torch.zeros(3, device="cuda")
    view = torch.eye(4, device="cuda")
    c6, col, feat = attributes.frame_attributes(z(0, 3), z(0, 3, 3), z(0, 3), z(0, 3), 1.0, z(0, 4), z(0, 4), z(0, 3), z(0, 3), z(0, 3),
                                                z(0, 16, 3), 3, cam, view)
    assert c6.shape == (0, 6) and col.shape == (0, 3) and feat.shape == (0, 18)
    (c6.sum() + col.sum() + feat.sum()).backward()
    o = lbs.lbs_deform(torch.zeros((0, 3), device="cuda"), None, None, torch.eye(4, device="cuda").repeat(24, 1, 1),
                       torch.eye(4, device="cuda").repeat(24, 1, 1), torch.zeros((5, 3), device="cuda"), torch.zeros((5, 3), device="cuda"),
                       torch.zeros((5, 3), device="cuda"), torch.eye(3, device="cuda"), torch.zeros(3, device="cuda"),
                       torch.rand((5, 3), device="cuda"), torch.full((5, 24), 1 / 24, device="cuda"))
    assert o["world_pts"].shape == (0, 3)


def test_single_gaussian_and_offscreen_only():
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizer
    W, H = 40, 40
    rs = _settings(W, H)
    rast = GaussianRasterizer(rs)
    for pos in ([0.0, 0.0, 3.0], [0.0, 0.0, -3.0], [50.0, 0.0, 3.0]):   # visible, behind the camera, far off screen
        means = torch.tensor([pos], device="cuda", requires_grad=True)
        cols = torch.tensor([[0.9, 0.5, 0.1]], device="cuda", requires_grad=True)
        op = torch.tensor([[0.8]], device="cuda", requires_grad=True)
        cov = torch.tensor([[0.01, 0, 0, 0.01, 0, 0.01]], device="cuda", requires_grad=True)
        extra = [torch.rand((1, 3), device="cuda", requires_grad=True) for _ in range(6)]
        color, radii, depth, alpha, feats = rast.forward_multi(means3D=means, means2D=torch.zeros((1, 3), device="cuda", requires_grad=True),
                                                               opacities=op, extra_colors=extra, colors_precomp=cols, cov3D_precomp=cov)
        (color.sum() + feats[2].sum()).backward()
        visible = int(radii[0]) > 0
        assert visible == (pos == [0.0, 0.0, 3.0])
        assert torch.isfinite(means.grad).all() and torch.isfinite(cols.grad).all()
        if not visible:
            assert float(cols.grad.abs().max()) == 0 and float(alpha.max()) == 0
        else:
            assert float(alpha.max()) > 0.5 and float(extra[2].grad.abs().max()) > 0 and float(extra[0].grad.abs().max()) == 0


def test_prefiltered_violation_is_an_error_not_a_device_trap():
    """prefiltered=True promises that no point fails the frustum test (the reference __trap()s the device otherwise,
    CR/auxiliary.h:156-160).  Here the call fails with the reference's message and the process lives on."""
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
    P, W, H = 500, 64, 48
    cam, g = util.make_scene(P, W, H, 4, 1, 0.05, behind_frac=0.1)   # 10 % of the points are behind the camera
    d = util.to_dev
    kw = dict(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=d(np.zeros(3, np.float32)),
              scale_modifier=1.0, viewmatrix=d(cam["viewmatrix"]), projmatrix=d(cam["projmatrix"]), sh_degree=1, campos=d(cam["campos"]),
              debug=False)
    args = dict(means3D=d(g["means3D"]), means2D=torch.zeros(P, 3, device="cuda"), opacities=d(g["opacities"]), shs=d(g["shs"]),
                scales=d(g["scales"]), rotations=d(g["rotations"]))
    ok = GaussianRasterizer(GaussianRasterizationSettings(prefiltered=False, **kw))(**args)
    with pytest.raises(RuntimeError, match="filtered although prefiltered"):
        GaussianRasterizer(GaussianRasterizationSettings(prefiltered=True, **kw))(**args)
    # a truthful caller: only the visible points, prefiltered=True -> same image as the unfiltered call
    vis = GaussianRasterizer(GaussianRasterizationSettings(prefiltered=False, **kw)).markVisible(args["means3D"])
    sub = {k: v[vis].contiguous() for k, v in args.items()}
    img = GaussianRasterizer(GaussianRasterizationSettings(prefiltered=True, **kw))(**sub)[0]
    assert torch.allclose(img, ok[0], atol=1e-6)
    # sync-free entry: the flag word travels with the overflow flag and raises at the deferred check
    e = torch.empty(0)
    out = _C.rasterize_gaussians_async(kw["bg"], args["means3D"], e, args["opacities"], args["scales"], args["rotations"], 1.0, e,
                                       kw["viewmatrix"], kw["projmatrix"], cam["tanfovx"], cam["tanfovy"], H, W, args["shs"], 1,
                                       kw["campos"], True, False)
    with pytest.raises(RuntimeError, match="filtered although prefiltered"):
        _C.AsyncCapacity.check(out[9])


def test_async_capacity_context_manager_verifies_trailing_frames():
    """Forward-only loop over the sync-free entry: an overflowing LAST frame must not stay silent."""
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    P, W, H = 3000, 96, 64
    cam, g = util.make_scene(P, W, H, 6, 0, 0.05)
    d = util.to_dev
    e = torch.empty(0)
    a = (d(np.zeros(3, np.float32)), d(g["means3D"]), e, d(g["opacities"]), d(g["scales"]), d(g["rotations"]), 1.0, e,
         d(cam["viewmatrix"]), d(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, d(g["shs"]), 0, d(cam["campos"]), False, False)
    with _C.AsyncCapacity.frames():
        for _ in range(3):
            _C.rasterize_gaussians_async(*a)
    with pytest.raises(RuntimeError, match="exceeded the binning capacity"):
        with _C.AsyncCapacity.frames():
            _C.rasterize_gaussians_async(*a)
            _C.rasterize_gaussians_async(*a, capacity=16)  # the last frame of the loop overflows
    _C.AsyncCapacity.check_all()
