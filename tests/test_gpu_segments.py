"""List segments (csrc/gsr_common.h "list segments"): in a frame of unequal lists the backward walks the list of an outlier tile in
up to four segments at the same time, each started from a checkpoint the forward wrote at the segment boundary.  The results must be
those of the whole-list walk (and of the oracle); the order entries must show that the frame really was cut."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

TILE_MASK = (1 << 22) - 1


def _clustered_scene(P, W, H, seed, cluster_frac=0.6, deg=2):
    """A uniform cloud with most Gaussians pulled into a small region of the image: a few tiles get lists several times the mean.
    Low opacities in the cluster, so that pixels keep contributing deep into those lists."""
    cam, g = util.make_scene(P, W, H, seed, deg, 0.02, 0.0)
    k = int(P * cluster_frac)
    z = g["means3D"][:k, 2:3]
    t = np.float32(cam["tanfovx"])
    centre = np.array([[0.28, -0.22]], np.float32) * z * t
    g["means3D"][:k, :2] = centre + 0.16 * g["means3D"][:k, :2]
    g["opacities"][:k] *= 0.25
    return cam, g


def _order(f, tiles):
    o = util.hip_query(f, "ORDER").view(np.uint32)
    mode, slots = int(o[0] & 0xFF), int(o[1])
    entries = o[2:2 + slots]
    return mode, slots, entries


@pytest.fixture
def no_segments():
    from mygauhuman_amd import _lib

    def switch(on):
        _lib.set_tuning("blend_segments", _lib.DEFAULT_BLEND_SEGMENTS if on else 0)
    yield switch
    _lib.set_tuning("blend_segments", _lib.DEFAULT_BLEND_SEGMENTS)


@pytest.mark.parametrize("mode", ["sh", "precomp"])
@pytest.mark.parametrize("dims", [(7000, 160, 128), (5000, 150, 70)], ids=["160x128", "ragged_150x70"])
def test_segmented_backward_equals_whole_walk_and_oracle(oracle, mode, dims, no_segments):
    P, W, H = dims
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    cam, g = _clustered_scene(P, W, H, seed=31)
    bg = np.array([0.2, 0.5, 0.7], np.float32)
    rng = np.random.default_rng(3)
    ref = util.oracle_forward(oracle, cam, g, bg, mode)
    solid = ref["img"]["fragile"] == 0
    dc = (rng.normal(0, 1, (3, H, W)) * solid).astype(np.float32)
    dd = (rng.normal(0, 1, (1, H, W)) * solid).astype(np.float32)
    da = (rng.normal(0, 1, (1, H, W)) * solid).astype(np.float32)
    want = oracle.rasterize_backward(ref, dc, dd, da)

    res = {}
    for on in (True, False):
        no_segments(on)
        f = util.hip_forward(cam, g, bg, mode, debug=True)
        omode, slots, entries = _order(f, tiles)
        assert omode == 1
        nseg = ((entries >> 25) & 7) + 1
        seg = (entries >> 22) & 7
        tile = entries & TILE_MASK
        if on:
            assert slots > tiles and nseg.max() >= 3, (slots, tiles, int(nseg.max()))  # the frame really is cut
            # every tile appears once per segment, segments 0 .. nseg - 1, and the segment count is the same in all its entries
            for t in np.unique(tile[nseg > 1]):
                sel = tile == t
                assert sorted(seg[sel].tolist()) == list(range(int(nseg[sel][0]))) and np.all(nseg[sel] == nseg[sel][0])
            assert sorted(tile[seg == 0].tolist()) == list(range(tiles))
        else:
            assert slots == tiles and nseg.max() == 1 and sorted(tile.tolist()) == list(range(tiles))
        res[on] = (f, util.hip_backward(f, dc, dd, da, debug=True))
    # the forward is the same kernel either way (the checkpoints are only extra stores): images bit-identical
    for k in ("color", "depth", "alpha"):
        assert torch.equal(res[True][0][k], res[False][0][k])
    names = ["dL_dmean2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D"]
    names += ["dL_dsh", "dL_dscales", "dL_drotations"] if mode == "sh" else []
    for n in names:
        a, b = res[True][1][n], res[False][1][n]
        util.assert_close(n + " (segments vs whole walk)", a, b, tol=2e-5, max_bad_frac=1e-4, outer_tol=2e-4)
        util.assert_close(n + " (segments vs oracle)", a.reshape(want[n].shape), want[n], tol=1e-4, max_bad_frac=2e-4)


@pytest.mark.parametrize("used", [(0, 5), (1, 2, 3, 4, 5), ()])
def test_segmented_feature_backward_equals_whole_walk(used, no_segments):
    """The fused multi-feature pass (18 extra channels): checkpoints carry the extra accumulators too."""
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizer
    from tests.test_gpu_multi import _settings
    P, W, H = 7000, 160, 128
    cam, g = _clustered_scene(P, W, H, seed=32, deg=3)
    bg = np.array([0.3, 0.6, 0.1], np.float32)
    rast = GaussianRasterizer(_settings(cam, bg, 3))
    wr = np.random.default_rng(9)
    ws = [util.to_dev(wr.normal(0, 1, (3, H, W)).astype(np.float32)) for _ in range(7)]
    wa = util.to_dev(wr.normal(0, 1, (1, H, W)).astype(np.float32))
    grads, images = {}, {}
    for on in (True, False):
        no_segments(on)
        rng = np.random.default_rng(4)
        t = {k: util.to_dev(v).requires_grad_(True) for k, v in g.items() if isinstance(v, np.ndarray)}
        extras = [util.to_dev(rng.uniform(0, 1, (P, 3)).astype(np.float32)).requires_grad_(True) for _ in range(6)]
        m2 = torch.zeros((P, 3), device="cuda", requires_grad=True)
        color, radii, depth, alpha, feats = rast.forward_multi(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], extra_colors=extras,
                                                               shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        loss = (color * ws[0]).sum() + (alpha * wa).sum() + sum((feats[i] * ws[1 + i]).sum() for i in used)
        loss.backward()
        images[on] = [color.detach(), alpha.detach(), depth.detach()] + [f.detach() for f in feats]
        grads[on] = {k: t[k].grad.cpu().numpy() for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        grads[on]["means2D"] = m2.grad.cpu().numpy()
        for i, e in enumerate(extras):
            grads[on][f"extra{i}"] = (e.grad if e.grad is not None else torch.zeros_like(e)).cpu().numpy()
    for a, b in zip(images[True], images[False]):
        assert torch.equal(a, b)
    for k in grads[True]:
        util.assert_close(k, grads[True][k], grads[False][k], tol=2e-5, max_bad_frac=1e-4, outer_tol=2e-4)


def test_tail_cut_segments_equal_whole_walk(oracle):
    """Knob blend_tail_cut (default off): the lists visited LAST are cut in two as well, in a uniform scene that cuts nothing by
    itself.  Same checks as above: the order really has the pieces, the gradients are those of the whole-list walk and of the oracle."""
    from mygauhuman_amd import _lib
    P, W, H = 6000, 160, 128
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    cam, g = util.make_scene(P, W, H, 41, 2, 0.03, 0.0)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    rng = np.random.default_rng(5)
    ref = util.oracle_forward(oracle, cam, g, bg, "sh")
    solid = ref["img"]["fragile"] == 0
    dc = (rng.normal(0, 1, (3, H, W)) * solid).astype(np.float32)
    dd = (rng.normal(0, 1, (1, H, W)) * solid).astype(np.float32)
    da = (rng.normal(0, 1, (1, H, W)) * solid).astype(np.float32)
    want = oracle.rasterize_backward(ref, dc, dd, da)
    res = {}
    try:
        for cut in (8, 0):
            _lib.set_tuning("blend_tail_cut", cut)
            f = util.hip_forward(cam, g, bg, "sh", debug=True)
            omode, slots, entries = _order(f, tiles)
            nseg = ((entries >> 25) & 7) + 1
            if cut:
                assert omode == 1 and slots > tiles and nseg.max() == 2, (slots, tiles, int(nseg.max()))
            else:
                assert slots == tiles and nseg.max() == 1
            res[cut] = util.hip_backward(f, dc, dd, da, debug=True)
    finally:
        _lib.set_tuning("blend_tail_cut", 0)
    for n in ("dL_dmean2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        util.assert_close(n + " (tail cut vs whole walk)", res[8][n], res[0][n], tol=2e-5, max_bad_frac=1e-4, outer_tol=2e-4)
        util.assert_close(n + " (tail cut vs oracle)", res[8][n].reshape(want[n].shape), want[n], tol=1e-4, max_bad_frac=2e-4)
