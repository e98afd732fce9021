"""BASELINE.json full sizes (C2: 50k/SH0/512^2 forward, C3: 200k/SH3/1024^2 fwd+bwd), checked through
size-independent properties: sortedness and range consistency of the binning, equality of the two binning
back-ends and of all blend configurations, determinism of the integer state, linearity of the backward."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _setup(P, W, H, deg):
    from mygauhuman_amd import synthetic
    cam, g = synthetic.uniform_scene(P, W, H, seed=0, sh_degree=deg)
    g["cov3D"] = np.zeros((P, 6), np.float32)
    return cam, g


def _check_binning(f, P, W, H):
    keys = util.hip_query(f, "KEYS_SORTED").view(np.uint64)
    pl = util.hip_query(f, "POINT_LIST").view(np.uint32)
    ranges = util.hip_query(f, "RANGES").view(np.uint32).astype(np.int64)
    tt = util.hip_query(f, "TILES_TOUCHED").view(np.uint32)
    off = util.hip_query(f, "POINT_OFFSETS").view(np.uint32)
    R = f["R"]
    assert R == int(tt.sum()) and int(off[-1]) == R
    np.testing.assert_array_equal(off, np.cumsum(tt.astype(np.uint64)).astype(np.uint32))
    assert np.all(np.diff(keys.astype(np.uint64)) >= 0) if R < 2 else bool(np.all(keys[1:] >= keys[:-1]))
    same = keys[1:] == keys[:-1]
    assert np.all(pl[1:][same] > pl[:-1][same])  # stability: ties keep Gaussian-index order
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    counts = np.bincount(tiles, minlength=ranges.shape[0])
    np.testing.assert_array_equal(ranges[:, 1] - ranges[:, 0], counts)
    nz = counts > 0
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    np.testing.assert_array_equal(ranges[nz, 0], starts[nz])
    radii = f["radii"].cpu().numpy()
    assert np.array_equal(np.bincount(pl, minlength=P) > 0, (radii > 0) & (tt > 0))
    return keys, pl, ranges


def test_c2_forward_properties():
    from mygauhuman_amd import _lib
    P, W, H = 50_000, 512, 512
    cam, g = _setup(P, W, H, 0)
    bg = np.zeros(3, np.float32)
    res = {}
    util.set_tile_cull(False)
    for mode in (_lib.BINNING_GLOBAL_RADIX, _lib.BINNING_TILE_BUCKET):
        _lib.check(_lib.lib.gsr_set_binning_mode(mode), "mode")
        f = util.hip_forward(cam, g, bg, "sh")
        res[mode] = (_check_binning(f, P, W, H), f)
    _lib.lib.gsr_set_binning_mode(_lib.DEFAULT_BINNING)
    util.set_tile_cull(_lib.DEFAULT_TILE_CULL)
    (ka, pa, ra), fa = res[0]
    (kb, pb, rb), fb = res[1]
    np.testing.assert_array_equal(ka, kb)
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ra, rb)
    assert torch.equal(fa["color"], fb["color"]) and torch.equal(fa["alpha"], fb["alpha"])
    # library default (tight tile culling): shorter lists, the SAME image bits -- culled instances never blended anything
    ft = util.hip_forward(cam, g, bg, "sh")
    rt = util.hip_query(ft, "RANGES").view(np.uint32).reshape(-1, 2).astype(np.int64)
    kept = int((rt[:, 1] - rt[:, 0]).sum())
    assert 0 < kept < fb["R"] and np.all(rt[:, 1] - rt[:, 0] <= rb[:, 1] - rb[:, 0])
    for k in ("color", "alpha", "depth"):
        assert torch.equal(ft[k], fb[k]), k
    assert torch.equal(ft["radii"], fb["radii"]) and ft["R"] == fb["R"]
    a = fa["alpha"]
    assert float(a.min()) >= 0 and float(a.max()) <= 1.0 + 1e-4
    assert float(util.to_dev(util.hip_query(fa, "FINAL_T")).min()) >= 0


def test_c3_forward_backward_properties():
    from mygauhuman_amd import _lib, synthetic
    P, W, H = 200_000, 1024, 1024
    cam, g = _setup(P, W, H, 3)
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    gt, mask = synthetic.loss_targets(W, H)
    outs = {}
    for waves in (1, 2, 4):
        _lib.set_tuning("blend_fwd_waves", waves)
        _lib.set_tuning("blend_bwd_waves", waves)
        if waves == 4:
            util.set_tile_cull(False)
            _check_binning(util.hip_forward(cam, g, bg, "sh"), P, W, H)
            util.set_tile_cull(_lib.DEFAULT_TILE_CULL)
        f = util.hip_forward(cam, g, bg, "sh")
        color, alpha = f["color"].cpu().numpy(), f["alpha"].cpu().numpy()
        if waves == 1:
            # the loss gradient of train.py:261-262 WITHOUT its 1 / (3 H W) normalisation: O(1) per pixel, so that the
            # per-Gaussian gradients are O(1..100) and the relative tolerances below mean something.  Fixed for all variants.
            dc = np.sign(color - gt).astype(np.float32)
            da = (0.2 * (alpha - mask)).astype(np.float32)
            dd = np.zeros_like(alpha)
        grads = util.hip_backward(f, dc, dd, da)
        outs[waves] = (f, color, alpha, grads, dc, da, dd)
    f4, c4, a4, g4, dc, da, dd = outs[4]
    assert f4["R"] > 1_000_000
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations"):
        assert float(np.abs(g4[k]).max()) > 1e-2, k  # the gradients are not vanishing: the comparisons below have teeth
    # tight tile culling (library default, the benched path) against the reference's full lists AT C3: the images are the same
    # bits (culled instances never blended anything) and the gradients agree to 1e-5 of the tensor scale (summation order)
    util.set_tile_cull(False)
    fn = util.hip_forward(cam, g, bg, "sh")
    gn = util.hip_backward(fn, dc, dd, da)
    util.set_tile_cull(_lib.DEFAULT_TILE_CULL)
    rt = util.hip_query(f4, "RANGES").view(np.uint32).reshape(-1, 2).astype(np.int64)
    rn = util.hip_query(fn, "RANGES").view(np.uint32).reshape(-1, 2).astype(np.int64)
    assert int((rt[:, 1] - rt[:, 0]).sum()) < int((rn[:, 1] - rn[:, 0]).sum()) == fn["R"]
    for k in ("color", "alpha", "depth"):
        assert torch.equal(f4[k], fn[k]), f"C3 {k}: tile culling changed the image"
    assert torch.equal(util.to_dev(util.hip_query(f4, "FINAL_T")), util.to_dev(util.hip_query(fn, "FINAL_T")))
    for k in g4:
        util.assert_close(f"{k} cull on/off", g4[k], gn[k], tol=1e-5, max_bad_frac=1e-5)
    for waves in (1, 2):
        f, c, a, gr = outs[waves][:4]
        # same per-pixel arithmetic up to FMA contraction choices of each template instantiation
        assert float(np.abs(c - c4).max()) < 5e-5 and float(np.abs(a - a4).max()) < 5e-5
        assert torch.equal(f["radii"], f4["radii"])
        for k in g4:  # the backward only differs in summation order
            util.assert_close(f"{k} waves{waves}", gr[k], g4[k], tol=1e-4, max_bad_frac=1e-5)
    # linearity of the backward in the incoming gradients: grads(2 dL) = 2 grads(dL); grads(dc,0,0)+grads(0,0,da) = grads(dc,0,da)
    g2 = util.hip_backward(f4, 2 * dc, dd, 2 * da)
    gc = util.hip_backward(f4, dc, dd, 0 * da)
    ga = util.hip_backward(f4, 0 * dc, dd, da)
    for k in g4:
        util.assert_close(f"{k} x2", g2[k], 2 * g4[k], tol=1e-4, max_bad_frac=1e-5)
        util.assert_close(f"{k} additivity", gc[k] + ga[k], g4[k], tol=1e-4, max_bad_frac=1e-5)
    # culled Gaussians get exactly zero gradient; visible ones finite
    invisible = f4["radii"].cpu().numpy() == 0
    for k, v in g4.items():
        assert np.isfinite(v).all(), k
        assert not np.any(v.reshape(P, -1)[invisible]), k
    _lib.set_tuning("blend_fwd_waves", 4)
    _lib.set_tuning("blend_bwd_waves", 4)


def test_c5_lbs_render_prune_at_500k(oracle):
    """BASELINE configs[4] at full size (fp32 SH; the fp16-SH leg is test_c5_fp16_sh_at_500k below): 500k articulated Gaussians -> per-frame LBS + fused render() forward / backward
    at 1024^2 -> k-NN based prune.  Size-independent properties: finite outputs, visible <=> radii > 0, gradients only on
    visible Gaussians, nearest-vertex distances equal to the brute-force oracle's, prune count equal to the mask count."""
    import types

    from mygauhuman_amd import densify, knn_cuda
    from mygauhuman_amd.gaussian_renderer import render
    from tools.train_demo import build
    P, V, W, H = 500_000, 6890, 1024, 1024
    model, cam, verts = build(P, V, W, H, seed=5)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    o = render(1, cam, model, pipe, torch.zeros(3, device="cuda"))
    assert o["render"].shape == (3, H, W) and o["normal"].shape == (3, H, W)
    for k in ("render", "render_alpha", "render_depth", "normal", "albedo", "render_axis"):
        assert torch.isfinite(o[k]).all(), k
    amax = float(o["render_alpha"].detach().max())
    assert 0.9 < amax <= 1.0 + 1e-4
    loss = o["render"].mean() + o["render_alpha"].mean() + o["normal"].mean() + o["render_axis"].mean()
    loss.backward()
    vis = o["visibility_filter"]
    assert int(vis.sum()) > 0.9 * P
    for p in (model._xyz, model._features_dc, model._scaling, model._rotation, model._opacity, model._normal):
        assert torch.isfinite(p.grad).all()
    assert float(model._features_dc.grad[~vis].abs().sum()) == 0.0 and float(model._opacity.grad[~vis].abs().sum()) == 0.0
    assert float(model._features_dc.grad[vis].abs().sum()) > 0.0
    # distance of every Gaussian to the SMPL surface (the prune prior, scene/gaussian_model.py:715-720) against brute force
    dist, idx = knn_cuda.knn_nearest(verts, model._xyz.detach())
    sub = np.random.default_rng(0).choice(P, 20000, replace=False)
    q = model._xyz.detach()[torch.from_numpy(sub).cuda()].cpu().numpy()
    ids = oracle.nearest_vertex(q, verts.cpu().numpy())
    np.testing.assert_array_equal(idx.cpu().numpy()[sub], ids)
    np.testing.assert_allclose(dist.cpu().numpy()[sub], oracle.nearest_dist(q, verts.cpu().numpy(), ids), rtol=2e-7)
    densify.training_setup(model, dict(xyz=1e-4))
    model._xyz.data[:1000] += 1.0                       # push 1000 Gaussians away from the body: they must be pruned
    mask = densify.densify_and_prune(model, 1e9, 0.0, 2.0, 0, t_vertices=verts)
    assert int(mask[:1000].sum()) == 1000
    assert model._xyz.shape[0] == P - int(mask.sum()) and model._features_rest.shape == (model._xyz.shape[0], 15, 3)


def test_c5_fp16_sh_at_500k():
    """BASELINE configs[4] storage mode at full size: 500k Gaussians, SH coefficients stored as fp16, 1024^2, forward + backward
    through the sync-free session (what bench.py --workload C5 runs).  Property: bit-identical outputs and gradients close to
    summation order against the fp32 path fed the SAME (fp16-rounded) coefficients."""
    import math

    from mygauhuman_amd import synthetic
    from mygauhuman_amd.fastpath import RasterSession
    P, W, H, deg = 500_000, 1024, 1024, 3
    cam, g = synthetic.uniform_scene(P, W, H, seed=0, sh_degree=deg, log_scale_mean=math.log(0.005))
    gt, mask = synthetic.loss_targets(W, H)
    to = util.to_dev
    half = to(g["shs"]).half()
    base = dict(means3D=to(g["means3D"]), opacities=to(g["opacities"]), scales=to(g["scales"]), rotations=to(g["rotations"]))
    camd = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
    bg = to(np.array([0.1, 0.2, 0.3], np.float32))
    res = {}
    for name, shs in (("f16", half), ("f32", half.float())):
        params = dict(base, shs=shs)
        s = RasterSession.calibrated(params, camd, bg, deg)
        color, depth, alpha, radii = s.forward(params, camd, bg, deg)
        dc = torch.sign(color - to(gt)).contiguous()
        da = (0.2 * (alpha - to(mask))).contiguous()
        out = dict(means3D=torch.empty(P, 3, device="cuda"), sh=torch.empty(P, 16, 3, device="cuda"), opacity=torch.empty(P, 1, device="cuda"),
                   scales=torch.empty(P, 3, device="cuda"), rotations=torch.empty(P, 4, device="cuda"))
        s.backward(params, camd, bg, deg, dc, s.dL_ddepth, da, out)
        assert not s.overflowed() and s.num_rendered() > 1_000_000
        res[name] = (color.clone(), alpha.clone(), depth.clone(), radii.clone(), {k: v.cpu().numpy() for k, v in out.items()})
    for i, k in enumerate(("color", "alpha", "depth", "radii")):
        assert torch.equal(res["f16"][i], res["f32"][i]), k
    for k, v in res["f16"][4].items():
        assert np.isfinite(v).all() and float(np.abs(v).max()) > 1e-2, k
        util.assert_close(f"C5 fp16-SH {k}", v, res["f32"][4][k], tol=1e-5, max_bad_frac=1e-5)


@pytest.mark.parametrize("P,W,H,deg", [(50_000, 512, 512, 0), (200_000, 1024, 1024, 3), (3000, 150, 70, 2)])
def test_fused_alpha_mask_loss_backward_equals_loss_kernel_plus_backward(P, W, H, deg):
    """session.backward_alpha_mask_loss forms the loss gradient inside the blend-backward kernel: the gradients must be those of
    alpha_mask_loss_backward() + backward() -- bit for bit with the deterministic reduction, to summation order without."""
    from mygauhuman_amd import _lib, synthetic
    from mygauhuman_amd.fastpath import RasterSession
    cam, g = synthetic.uniform_scene(P, W, H, seed=3, sh_degree=deg)
    gt, mask = synthetic.loss_targets(W, H)
    to = util.to_dev
    M = (deg + 1) ** 2
    params = dict(means3D=to(g["means3D"]), shs=to(g["shs"]), opacities=to(g["opacities"]), scales=to(g["scales"]), rotations=to(g["rotations"]))
    camd = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
    bg, gt_d, mask_d = to(np.array([0.1, 0.2, 0.3], np.float32)), to(gt), to(mask)
    new_out = lambda: dict(means3D=torch.empty(P, 3, device="cuda"), sh=torch.empty(P, M, 3, device="cuda"),  # noqa: E731
                           opacity=torch.empty(P, 1, device="cuda"), scales=torch.empty(P, 3, device="cuda"),
                           rotations=torch.empty(P, 4, device="cuda"))
    for det in (1, 0):
        _lib.set_tuning("deterministic", det)
        try:
            s = RasterSession.calibrated(params, camd, bg, deg)
            s.forward(params, camd, bg, deg)
            a, b = new_out(), new_out()
            dc, da = s.alpha_mask_loss_backward(gt_d, mask_d, 0.1)
            s.backward(params, camd, bg, deg, dc, s.dL_ddepth, da, a)
            s.backward_alpha_mask_loss(params, camd, bg, deg, gt_d, mask_d, 0.1, b)
            for k in a:
                assert float(a[k].abs().max()) > 0, k
                if det:
                    assert torch.equal(a[k], b[k]), k
                else:
                    util.assert_close(f"fused loss {k}", b[k].cpu().numpy(), a[k].cpu().numpy(), tol=2e-5, max_bad_frac=1e-5)
        finally:
            _lib.set_tuning("deterministic", 0)


@pytest.mark.parametrize("cfg", ["C2", "C3"])
def test_full_size_matches_oracle(oracle, cfg):
    """HIP with the LIBRARY DEFAULTS (tile-bucket binning, tight tile culling, LDS-fold backward: the benched path) against the
    CPU oracle at BASELINE.json's full sizes -- C2 = 50k / SH0 / 512^2 forward, C3 = 200k / SH3 / 1024^2 forward + backward
    (CR/forward.cu:261-383, CR/backward.cu:399-587 at the size the bench line is quoted on).  Integer state bit-exact, tile lists
    sublists of the reference lists in the reference order, images and all eight gradients at 1e-4 with O(1) upstream gradients
    and no element beyond 1e-3 of its tensor's scale."""
    import os
    from mygauhuman_amd import synthetic
    P, W, H, deg = (50_000, 512, 512, 0) if cfg == "C2" else (200_000, 1024, 1024, 3)
    cam, g = _setup(P, W, H, deg)
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    oracle.set_threads(min(16, os.cpu_count() or 1))
    try:
        ref = util.oracle_forward(oracle, cam, g, bg, "sh")
        f = util.hip_forward(cam, g, bg, "sh")
        pre, b, img = ref["pre"], ref["bin"], ref["img"]
        np.testing.assert_array_equal(f["radii"].cpu().numpy(), pre["radii"])
        assert f["R"] == b["R"]
        np.testing.assert_array_equal(util.hip_query(f, "TILES_TOUCHED").view(np.uint32), pre["tiles_touched"])
        np.testing.assert_array_equal(util.hip_query(f, "POINT_OFFSETS").view(np.uint32), b["offsets"])
        vis = pre["radii"] > 0
        for q, k in (("DEPTHS", "depths"), ("MEANS2D", "means2D"), ("CONIC_OPACITY", "conic_opacity"), ("RGB", "rgb"),
                     ("COV3D", "cov3D"), ("CLAMPED", "clamped")):
            np.testing.assert_array_equal(util.hip_query(f, q)[vis], pre[k][vis], err_msg=q)
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        kept = util.assert_lists_are_sublists(f, b, tiles)
        assert 0.5 * b["R"] < kept < b["R"]
        solid = img["fragile"] == 0
        assert solid.mean() > 0.995
        ncon = util.hip_query(f, "N_CONTRIB").view(np.uint32)
        assert np.all(ncon[solid] <= img["n_contrib"][solid])
        util.assert_close("final_T", util.hip_query(f, "FINAL_T"), img["final_T"], mask=solid)
        for k in ("color", "depth", "alpha"):
            util.assert_close(k, f[k].cpu().numpy(), img[k], mask=np.broadcast_to(solid, img[k].shape))
        if cfg == "C2":
            return
        gt, mask = synthetic.loss_targets(W, H)
        rng = np.random.default_rng(3)
        # the un-normalised loss gradient of train.py:261-262 (O(1) per pixel) plus a depth gradient so that all three image
        # gradients are live; zero on the pixels whose forward state may legitimately differ
        dc = (np.sign(img["color"] - gt) * solid).astype(np.float32)
        da = (0.2 * (img["alpha"] - mask) * solid).astype(np.float32)
        dd = (rng.normal(0, 0.3, (1, H, W)) * solid).astype(np.float32)
        want = oracle.rasterize_backward(ref, dc, dd, da)
        got = util.hip_backward(f, dc, dd, da)
        for n in ("dL_dmean2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
            assert float(np.abs(want[n]).max()) > 1e-2, n
            util.assert_close(n, got[n].reshape(want[n].shape), want[n], tol=1e-4, max_bad_frac=1e-4, outer_tol=1e-3)
    finally:
        oracle.set_threads(min(8, os.cpu_count() or 1))


@pytest.mark.parametrize("cfg", ["C3", "C5"])
def test_benched_session_call_matches_oracle(oracle, cfg):
    """The call bench.py TIMES -- RasterSession.forward + backward_alpha_mask_loss, library defaults -- against the oracle at the
    benched sizes (VERDICT r3 #7a, #7c): C3 = 200k / SH3 / 1024^2, C5 = 500k Gaussians with fp16-stored SH (the oracle gets the
    rounded coefficients in fp32).  The loss gradient the kernel forms in its prologue (train.py:261-262: L1(color, gt) +
    0.1 MSE(alpha, mask)) is restated in numpy from the session's own images and handed to oracle.rasterize_backward.  The
    targets are made EQUAL to the rendered values on the fragile pixels (cut-off decisions within 2e-5 of their threshold), so the
    fused kernel -- which cannot mask -- sees an exactly zero gradient there.  CR/backward.cu:399-587, CR/forward.cu:261-383."""
    import math
    import os

    from mygauhuman_amd import synthetic
    from mygauhuman_amd.fastpath import RasterSession
    P, W, H, deg = (200_000, 1024, 1024, 3) if cfg == "C3" else (500_000, 1024, 1024, 3)
    cam, g = synthetic.uniform_scene(P, W, H, seed=0, sh_degree=deg, log_scale_mean=math.log(0.01 if cfg == "C3" else 0.005))
    gt, mask = synthetic.loss_targets(W, H)
    to = util.to_dev
    shs_dev = to(g["shs"])
    if cfg == "C5":
        shs_dev = shs_dev.half()
        g = dict(g, shs=shs_dev.float().cpu().numpy())
    g["cov3D"] = np.zeros((P, 6), np.float32)
    params = dict(means3D=to(g["means3D"]), shs=shs_dev, opacities=to(g["opacities"]), scales=to(g["scales"]), rotations=to(g["rotations"]))
    camd = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
    bg = np.zeros(3, np.float32)
    oracle.set_threads(min(16, os.cpu_count() or 1))
    try:
        ref = util.oracle_forward(oracle, cam, g, bg, "sh")
        solid = ref["img"]["fragile"] == 0
        assert solid.mean() > 0.995
        s = RasterSession.calibrated(params, camd, to(bg), deg)
        color, depth, alpha, radii = s.forward(params, camd, to(bg), deg)
        np.testing.assert_array_equal(radii.cpu().numpy(), ref["pre"]["radii"])
        color_h, alpha_h = color.cpu().numpy(), alpha.cpu().numpy()
        util.assert_close("color", color_h, ref["img"]["color"], mask=np.broadcast_to(solid, color_h.shape))
        util.assert_close("alpha", alpha_h, ref["img"]["alpha"], mask=solid[None])
        gt2 = np.where(solid[None], gt, color_h).astype(np.float32)
        mask2 = np.where(solid[None], mask, alpha_h).astype(np.float32)
        out = dict(means3D=torch.empty(P, 3, device="cuda"), sh=torch.empty(P, 16, 3, device="cuda"), opacity=torch.empty(P, 1, device="cuda"),
                   scales=torch.empty(P, 3, device="cuda"), rotations=torch.empty(P, 4, device="cuda"))
        s.backward_alpha_mask_loss(params, camd, to(bg), deg, to(gt2), to(mask2), 0.1, out)
        assert not s.overflowed()
        npix = float(W * H)
        dc = (np.sign(color_h - gt2) / (3.0 * npix)).astype(np.float32)
        da = (2.0 * 0.1 / npix * (alpha_h - mask2)).astype(np.float32)
        assert float(np.abs(dc[:, ~solid]).max(initial=0.0)) == 0.0 and float(np.abs(da[:, ~solid]).max(initial=0.0)) == 0.0
        want = oracle.rasterize_backward(ref, dc, np.zeros((1, H, W), np.float32), da)
        for k, n in (("means3D", "dL_dmeans3D"), ("sh", "dL_dsh"), ("opacity", "dL_dopacity"), ("scales", "dL_dscales"),
                     ("rotations", "dL_drotations")):
            assert float(np.abs(want[n]).max()) > 0, n
            util.assert_close(f"{cfg} benched call {n}", out[k].cpu().numpy().reshape(want[n].shape), want[n], tol=1e-4, max_bad_frac=1e-4,
                              outer_tol=1e-3)
    finally:
        oracle.set_threads(min(8, os.cpu_count() or 1))


def test_render_bench_scene_matches_seven_oracle_passes(oracle):
    """render() on the scene bench.py's `extra.render_200k` times (200k articulated Gaussians, 1024^2, ring camera 0): all seven
    images + alpha + depth and the radii against SEVEN oracle rasterizer passes over the per-Gaussian inputs the HIP path itself
    produced (the pre-raster chain is pinned against float64 in test_gpu_render.py) -- list segments and the long-list sort at the
    benched size (VERDICT r3 #7b; gaussian_renderer/__init__.py:203-272)."""
    import os
    import types

    from mygauhuman_amd import human_synth, lbs
    from mygauhuman_amd.attributes import frame_attributes
    from mygauhuman_amd.gaussian_renderer import render
    P, V, W, H = 200_000, 6890, 1024, 1024
    model, body = human_synth.build(P, V, "cuda", seed=0)
    cam = human_synth.view_camera(body, W, H, 0, n_views=8, device="cuda")
    bg = np.zeros(3, np.float32)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    with torch.no_grad():
        out = render(1, cam, model, pipe, util.to_dev(bg))
        act = model.frame_activations()
        _, world, _, tf, _, wn = lbs.coarse_deform_c2source(model.SMPL_NEUTRAL, model.get_xyz[None], cam.smpl_param, cam.big_pose_smpl_param,
                                                            cam.big_pose_world_vertex[None], normals=act.normal[None], lean=True)
        cov_h, col_h, feat_h = frame_attributes(world.reshape(-1, 3), tf.reshape(-1, 3, 3), wn.reshape(-1, 3), act.scaling, 1.0,
                                                model._rotation, act.rotation, act.albedo, act.roughness, act.occlusion,
                                                (model._features_dc, model._features_rest), 3, cam.camera_center, cam.world_view_transform)
    c = cam.cam_np
    means, cov6, op = world.reshape(-1, 3).cpu().numpy(), cov_h.cpu().numpy(), act.opacity.cpu().numpy()
    feats = feat_h.cpu().numpy()
    sets = [col_h.cpu().numpy()] + [np.ascontiguousarray(feats[:, 3 * k:3 * k + 3]) for k in range(6)]
    keys = ("render", "normal", "world_normal", "albedo", "occlusion", "roughness", "render_axis")
    oracle.set_threads(min(16, os.cpu_count() or 1))
    try:
        solid = None
        for k, cs in zip(keys, sets):
            r = oracle.rasterize_forward(means, op, c["viewmatrix"], c["projmatrix"], c["campos"], W, H, c["tanfovx"], c["tanfovy"], bg,
                                         cov3D_precomp=cov6, colors_precomp=cs)
            if solid is None:
                solid = r["img"]["fragile"] == 0
                assert solid.mean() > 0.99
                np.testing.assert_array_equal(out["radii"].cpu().numpy(), r["pre"]["radii"])
                lens = r["bin"]["ranges"][:, 1].astype(np.int64) - r["bin"]["ranges"][:, 0].astype(np.int64)
                assert lens.max() > 1024, "the benched frame has lists that take the long-list sort and are cut into segments"
                util.assert_close("render_alpha", out["render_alpha"].cpu().numpy(), r["img"]["alpha"], mask=solid[None])
                util.assert_close("render_depth", out["render_depth"].cpu().numpy(), r["img"]["depth"], mask=solid[None])
            util.assert_close(k, out[k].cpu().numpy(), r["img"]["color"], mask=np.broadcast_to(solid, (3, H, W)))
    finally:
        oracle.set_threads(min(8, os.cpu_count() or 1))
