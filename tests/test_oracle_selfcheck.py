"""Self-consistency of the CPU oracle (it is unpinned against reference *outputs* for the rasterizer, so it is
at least checked against itself): analytic backward vs central finite differences of its own forward,
binning invariants, and distCUDA2 Morton/box search == brute force."""
import numpy as np
import pytest

from mygauhuman_amd import synthetic


def _scene(P=300, W=80, H=56, seed=3, deg=3):
    cam, g = synthetic.uniform_scene(P, W, H, seed=seed, sh_degree=deg, log_scale_mean=np.log(0.05))
    return cam, g


def _forward(oracle, cam, g, bg, mode):
    kw = dict(scale_modifier=1.0)
    if mode == "sh":
        kw.update(scales=g["scales"], rotations=g["rotations"], shs=g["shs"], degree=g["sh_degree"])
    else:
        kw.update(cov3D_precomp=g["cov3D"], colors_precomp=g["colors"])
    return oracle.rasterize_forward(g["means3D"], g["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                                    cam["W"], cam["H"], cam["tanfovx"], cam["tanfovy"], bg, **kw)


def test_binning_invariants(oracle):
    cam, g = _scene(P=2000, W=130, H=70)
    fwd = _forward(oracle, cam, g, np.zeros(3, np.float32), "sh")
    b, pre = fwd["bin"], fwd["pre"]
    assert b["R"] == int(pre["tiles_touched"].sum()) and b["R"] > 0
    assert np.all(np.diff(b["keys_sorted"].astype(np.uint64)) >= 0)
    # stable: equal keys keep emission (= Gaussian index) order
    same = b["keys_sorted"][1:] == b["keys_sorted"][:-1]
    assert np.all(b["point_list"][1:][same] > b["point_list"][:-1][same])
    tiles = (b["keys_sorted"] >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles):
        lo, hi = b["ranges"][t]
        assert np.all(tiles[lo:hi] == t) and (hi - lo) == np.sum(tiles == t)
    assert sorted(b["keys_unsorted"].tolist()) == b["keys_sorted"].tolist()


def _random_cov(rng, P, s=0.05):
    A = rng.normal(0, s, (P, 3, 3)).astype(np.float32)
    return np.stack([(a @ a.T + 1e-4 * np.eye(3))[np.triu_indices(3)] for a in A]).astype(np.float32)


def _fd_compare(oracle, cam, g, bg, mode, checks, rng, rtol=0.02, atol=0.05):
    """checks: list of (input key, grad name, eps, direction mask or None)."""
    W, H = cam["W"], cam["H"]
    wc = rng.normal(0, 1, (3, H, W)).astype(np.float32)
    wa = rng.normal(0, 1, (1, H, W)).astype(np.float32)
    wd = np.zeros((1, H, W), np.float32)  # the reference drops d(depth image)/d(per-Gaussian depth), CR/backward.cu:541-549

    def loss(gg):
        f = _forward(oracle, cam, gg, bg, mode)
        return float((f["img"]["color"].astype(np.float64) * wc).sum() + (f["img"]["alpha"].astype(np.float64) * wa).sum())

    grads = oracle.rasterize_backward(_forward(oracle, cam, g, bg, mode), wc, wd, wa)
    for key, gname, eps, mask in checks:
        base = g[key]
        d = rng.normal(0, 1, base.shape).astype(np.float32)
        if mask is not None:
            d = d * mask
        gp, gm = dict(g), dict(g)
        gp[key] = (base + eps * d).astype(np.float32)
        gm[key] = (base - eps * d).astype(np.float32)
        fd = (loss(gp) - loss(gm)) / (2 * eps)
        an = float((grads[gname].reshape(base.shape).astype(np.float64) * d).sum())
        assert abs(fd - an) <= rtol * max(abs(an), abs(fd)) + atol, (mode, key, fd, an)


@pytest.mark.parametrize("mode", ["sh", "precomp"])
def test_backward_matches_finite_differences_overlapping(oracle, mode):
    """40 overlapping Gaussians, smooth renderer (cut-offs off): the blend recurrences (accum_rec, T replay,
    background term) are consistent with the forward.  Perturbations that move the integer radius / depth order
    (scales, z) are checked on isolated Gaussians below: they make the forward itself discontinuous."""
    cam, g = _scene(P=40, seed=3)
    g["scales"] = (g["scales"] * 1.6).astype(np.float32)
    rng = np.random.default_rng(7)
    g["cov3D"] = _random_cov(rng, 40, 0.08)
    bg = np.array([0.3, 0.1, 0.7], np.float32)
    xy = np.array([1, 1, 0], np.float32)
    checks = [("means3D", "dL_dmeans3D", 2e-4, xy), ("opacities", "dL_dopacity", 2e-3, None)]
    checks += [("shs", "dL_dsh", 1e-3, None)] if mode == "sh" else [("colors", "dL_dcolors", 1e-2, None)]
    oracle.set_thresholds(0.0, 0.0)
    try:
        for _ in range(3):
            _fd_compare(oracle, cam, g, bg, mode, checks, rng)
    finally:
        oracle.set_thresholds()


@pytest.mark.parametrize("mode", ["sh", "precomp"])
def test_backward_matches_finite_differences_single(oracle, mode):
    """One Gaussian at a time: every input (incl. z, scales, quaternion, cov3D) against finite differences."""
    rng = np.random.default_rng(9)
    bg = np.array([0.2, 0.5, 0.1], np.float32)
    oracle.set_thresholds(0.0, 0.0)
    try:
        for seed in range(4):
            cam, g = _scene(P=1, seed=seed)
            g["means3D"][:] = rng.uniform(-0.4, 0.4, (1, 3)) + np.array([0, 0, 3.0])
            g["scales"] = (g["scales"] * 1.2).astype(np.float32)
            g["cov3D"] = _random_cov(rng, 1, 0.06)
            checks = [("means3D", "dL_dmeans3D", 1e-3, None), ("opacities", "dL_dopacity", 1e-3, None)]
            if mode == "sh":
                checks += [("scales", "dL_dscales", 3e-4, None), ("rotations", "dL_drotations", 1e-3, None),
                           ("shs", "dL_dsh", 1e-3, None)]
            else:
                checks += [("cov3D", "dL_dcov3D", 3e-5, None), ("colors", "dL_dcolors", 1e-2, None)]
            _fd_compare(oracle, cam, g, bg, mode, checks, rng)
    finally:
        oracle.set_thresholds()


def test_depth_loss_reaches_opacity_only(oracle):
    """depth-image loss: gradient w.r.t. opacity matches finite differences (no path to geometry in the reference)."""
    cam, g = _scene(P=200)
    bg = np.zeros(3, np.float32)
    W, H = cam["W"], cam["H"]
    rng = np.random.default_rng(11)
    wd = rng.normal(0, 1, (1, H, W)).astype(np.float32)
    z3 = np.zeros((3, H, W), np.float32)
    z1 = np.zeros((1, H, W), np.float32)
    d = rng.normal(0, 1, g["opacities"].shape).astype(np.float32)
    eps = 1e-3

    def loss(op):
        gg = dict(g)
        gg["opacities"] = op.astype(np.float32)
        return float((_forward(oracle, cam, gg, bg, "sh")["img"]["depth"].astype(np.float64) * wd).sum())

    oracle.set_thresholds(0.0, 0.0)
    try:
        fwd = _forward(oracle, cam, g, bg, "sh")
        grads = oracle.rasterize_backward(fwd, z3, wd, z1)
        fd = (loss(g["opacities"] + eps * d) - loss(g["opacities"] - eps * d)) / (2 * eps)
    finally:
        oracle.set_thresholds()
    an = float((grads["dL_dopacity"].astype(np.float64) * d).sum())
    assert abs(fd - an) <= 0.02 * max(abs(an), abs(fd)) + 0.05


def test_dist2_morton_equals_brute(oracle):
    rng = np.random.default_rng(5)
    for P in (5, 700, 2600):
        pts = rng.normal(0, 1, (P, 3)).astype(np.float32)
        pts[: P // 10] = pts[P // 10: 2 * (P // 10)]  # coincident points -> zero distances
        a = oracle.dist2_brute(pts)
        b, codes, order = oracle.dist2_morton(pts)
        np.testing.assert_array_equal(a, b)
        assert np.all(np.diff(codes[order].astype(np.int64)) >= 0)


@pytest.mark.parametrize("kind", ["normal", "duplicates", "grid", "tiny"])
def test_knn_boxes_equals_brute(oracle, kind):
    """The box-pruned exact k-NN that checks the HIP k-NN at 200k / 500k points == the brute-force definition (indices incl.
    lowest-index ties, distance bits)."""
    rng = np.random.default_rng(11)
    P = 5000
    pts = rng.normal(0, 1, (P, 3)).astype(np.float32)
    if kind == "duplicates":
        pts = np.concatenate([pts[:P // 2], pts[:P // 2]]).astype(np.float32)
    elif kind == "grid":
        pts = np.stack(np.meshgrid(np.arange(15), np.arange(20), np.arange(12), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    elif kind == "tiny":
        pts = pts[:3]
    for k in (1, 2, 3):
        k = min(k, pts.shape[0])
        a, b = oracle.knn_self(pts, k), oracle.knn_self_boxes(pts, k)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
