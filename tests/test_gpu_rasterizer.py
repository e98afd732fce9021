"""GPU parity tests: the HIP rasterizer (through the raw `_C` bindings = the C ABI of libgsr.so) against the CPU
oracle on identical seeded inputs.  Integer state is compared bit-exactly, images / gradients within 1e-4."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

CASES = [
    # P, W, H, seed, deg, scale, behind_frac
    (3000, 130, 70, 0, 3, 0.02, 0.05),   # ragged image (not a multiple of 16), culled Gaussians
    (1500, 64, 64, 1, 0, 0.05, 0.0),     # SH degree 0, large splats -> long per-tile lists
    (1, 48, 32, 2, 3, 0.05, 0.0),        # single Gaussian
    (20000, 256, 192, 3, 2, 0.01, 0.02),
    (700, 16, 16, 4, 1, 0.2, 0.0),       # one tile, everything overlaps, list longer than a batch
    (9000, 24, 16, 5, 0, 0.3, 0.0),      # per-tile lists > 4096: the tile-bucket sort's chunk + global-merge path
    (1500, 16, 16, 6, 1, 0.2, 0.0),      # one list of 1025..2048 keys: four register runs of 512 + two LDS merge levels
    (3000, 16, 16, 7, 0, 0.2, 0.0),      # one list of 2049..4096 keys: four register runs of 1024
]


def _bg(seed):
    return np.random.default_rng(seed + 5).uniform(0, 1, 3).astype(np.float32)


@pytest.fixture(scope="module", params=[(1, 0, 0), (2, 0, 0), (4, 0, 0), (4, 1, 0), (2, 1, 0), (4, 2, 0), (4, 3, 0), (4, 3, 1)],
                ids=lambda p: f"waves{p[0]}red{p[1]}" + ("sp" if p[2] else ""))
def waves(request):
    """(waves per tile, backward reduction: 0 = DPP rows, 1 = MFMA on folded rows, 2 = transposed MFMA contraction, 3 = LDS folds,
    blend layout: 0 = a wave per 8x8 quadrant, 1 = a wave per 4x4 block with four survivors per step)"""
    from mygauhuman_amd import _lib
    w, red, layout = request.param
    util.skip_unless_experiments(red in (1, 2) or layout == 1)
    _lib.set_tuning("blend_fwd_waves", w)
    _lib.set_tuning("blend_bwd_waves", w)
    _lib.set_tuning("blend_bwd_reduce", red)
    _lib.set_tuning("blend_layout", layout)
    yield request.param
    _lib.set_tuning("blend_fwd_waves", 4)
    _lib.set_tuning("blend_bwd_waves", 4)
    _lib.set_tuning("blend_bwd_reduce", _lib.DEFAULT_BWD_REDUCE)
    _lib.set_tuning("blend_layout", _lib.DEFAULT_BLEND_LAYOUT)


@pytest.fixture(scope="module", params=["radix", "bucket", "bucket_tight"])
def binning(request):
    """radix / bucket: the reference's instance lists, bit for bit.  bucket_tight (the library default): instances whose
    tile the Gaussian cannot reach with alpha >= 1/255 are dropped -- images and gradients unchanged, lists are sublists."""
    from mygauhuman_amd import _lib
    _lib.check(_lib.lib.gsr_set_binning_mode(_lib.BINNING_GLOBAL_RADIX if request.param == "radix" else _lib.BINNING_TILE_BUCKET),
               "gsr_set_binning_mode")
    util.set_tile_cull(request.param == "bucket_tight")
    yield request.param
    _lib.lib.gsr_set_binning_mode(_lib.DEFAULT_BINNING)
    util.set_tile_cull(_lib.DEFAULT_TILE_CULL)


@pytest.mark.parametrize("case", CASES, ids=[f"P{c[0]}_{c[1]}x{c[2]}" for c in CASES])
@pytest.mark.parametrize("mode", ["sh", "precomp"])
def test_forward_matches_oracle(oracle, case, mode, waves, binning):
    P, W, H, seed, deg, scale, behind = case
    cam, g = util.make_scene(P, W, H, seed, deg, scale, behind)
    bg = _bg(seed)
    ref = util.oracle_forward(oracle, cam, g, bg, mode)
    f = util.hip_forward(cam, g, bg, mode, debug=True)
    pre, b, img = ref["pre"], ref["bin"], ref["img"]

    # ---- integer / index state: bit-exact
    np.testing.assert_array_equal(f["radii"].cpu().numpy(), pre["radii"])
    assert f["R"] == b["R"]
    np.testing.assert_array_equal(util.hip_query(f, "TILES_TOUCHED").view(np.uint32), pre["tiles_touched"])
    np.testing.assert_array_equal(util.hip_query(f, "POINT_OFFSETS").view(np.uint32), b["offsets"])
    tight = binning == "bucket_tight"
    if tight:
        kept = util.assert_lists_are_sublists(f, b, ((W + 15) // 16) * ((H + 15) // 16))
        assert kept <= b["R"]
    else:
        np.testing.assert_array_equal(util.hip_query(f, "KEYS_SORTED").view(np.uint64), b["keys_sorted"])
        np.testing.assert_array_equal(util.hip_query(f, "POINT_LIST").view(np.uint32), b["point_list"])
        np.testing.assert_array_equal(util.hip_query(f, "RANGES").view(np.uint32), b["ranges"])
    # ---- per-Gaussian float state: same operation order, no FMA contraction -> identical bits
    vis = pre["radii"] > 0
    np.testing.assert_array_equal(util.hip_query(f, "DEPTHS")[vis], pre["depths"][vis])
    np.testing.assert_array_equal(util.hip_query(f, "MEANS2D")[vis], pre["means2D"][vis])
    np.testing.assert_array_equal(util.hip_query(f, "CONIC_OPACITY")[vis], pre["conic_opacity"][vis])
    np.testing.assert_array_equal(util.hip_query(f, "RGB")[vis], pre["rgb"][vis])
    if mode == "sh":
        np.testing.assert_array_equal(util.hip_query(f, "COV3D")[vis], pre["cov3D"][vis])
        np.testing.assert_array_equal(util.hip_query(f, "CLAMPED")[vis], pre["clamped"][vis])
    # ---- images: 1e-4, except pixels where the oracle says a cut-off test was decided by < 2e-5 relative margin
    solid = img["fragile"] == 0
    assert solid.mean() > 0.995
    ncon = util.hip_query(f, "N_CONTRIB").view(np.uint32)
    if tight:  # positions count the shorter lists
        assert np.all(ncon[solid] <= img["n_contrib"][solid])
    else:
        np.testing.assert_array_equal(ncon[solid], img["n_contrib"][solid])
    util.assert_close("final_T", util.hip_query(f, "FINAL_T"), img["final_T"], mask=solid)
    util.assert_close("color", f["color"].cpu().numpy(), img["color"], mask=np.broadcast_to(solid, (3, H, W)))
    util.assert_close("depth", f["depth"].cpu().numpy(), img["depth"], mask=solid[None])
    util.assert_close("alpha", f["alpha"].cpu().numpy(), img["alpha"], mask=solid[None])


@pytest.mark.parametrize("case", CASES, ids=[f"P{c[0]}_{c[1]}x{c[2]}" for c in CASES])
@pytest.mark.parametrize("mode", ["sh", "precomp"])
def test_backward_matches_oracle(oracle, case, mode, waves):
    P, W, H, seed, deg, scale, behind = case
    cam, g = util.make_scene(P, W, H, seed, deg, scale, behind)
    bg = _bg(seed)
    rng = np.random.default_rng(seed + 9)
    ref = util.oracle_forward(oracle, cam, g, bg, mode)
    solid = ref["img"]["fragile"] == 0
    # zero the incoming gradient on fragile pixels: their forward state may legitimately differ
    dc = (rng.normal(0, 1, (3, H, W)) * solid).astype(np.float32)
    dd = (rng.normal(0, 1, (1, H, W)) * solid).astype(np.float32)
    da = (rng.normal(0, 1, (1, H, W)) * solid).astype(np.float32)
    want = oracle.rasterize_backward(ref, dc, dd, da)
    f = util.hip_forward(cam, g, bg, mode, debug=True)
    got = util.hip_backward(f, dc, dd, da, debug=True)
    names = ["dL_dmean2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D"]
    names += ["dL_dsh", "dL_dscales", "dL_drotations"] if mode == "sh" else []
    for n in names:
        util.assert_close(n, got[n].reshape(want[n].shape), want[n], tol=1e-4, max_bad_frac=2e-4)


def test_empty_and_invisible_inputs(oracle):
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    cam, g = util.make_scene(0, 32, 32, 0, 3)
    f = util.hip_forward(cam, g, np.zeros(3, np.float32), "sh")
    assert f["R"] == 0 and f["color"].abs().max().item() == 0 and f["geom"].numel() == 0
    # all Gaussians behind the camera: nothing rendered, background everywhere, backward gives zeros
    cam, g = util.make_scene(500, 40, 24, 1, 3, behind_frac=1.0)
    bg = np.array([0.25, 0.5, 0.75], np.float32)
    f = util.hip_forward(cam, g, bg, "sh", debug=True)
    assert f["R"] == 0 and int(f["radii"].max()) == 0
    np.testing.assert_array_equal(f["color"].cpu().numpy(), np.broadcast_to(bg[:, None, None], (3, 24, 40)))
    grads = util.hip_backward(f, np.ones((3, 24, 40), np.float32), np.ones((1, 24, 40), np.float32),
                              np.ones((1, 24, 40), np.float32), debug=True)
    assert all(np.all(v == 0) for v in grads.values())
    vis = _C.mark_visible(util.to_dev(g["means3D"]), util.to_dev(cam["viewmatrix"]), util.to_dev(cam["projmatrix"]))
    assert not vis.any()


def test_mark_visible_matches_oracle(oracle):
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    cam, g = util.make_scene(5000, 64, 64, 3, 0, behind_frac=0.3)
    g["means3D"][:10, 2] = 0.2  # exactly on the near plane -> culled (z <= 0.2)
    got = _C.mark_visible(util.to_dev(g["means3D"]), util.to_dev(cam["viewmatrix"]), util.to_dev(cam["projmatrix"]))
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.mark_visible(g["means3D"], cam["viewmatrix"], cam["projmatrix"]))


def test_forward_is_deterministic_and_stream_safe():
    cam, g = util.make_scene(20000, 200, 120, 5, 3)
    bg = np.zeros(3, np.float32)
    a = util.hip_forward(cam, g, bg, "sh")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b = util.hip_forward(cam, g, bg, "sh")
    s.synchronize()
    assert a["R"] == b["R"]
    for k in ("color", "depth", "alpha", "radii"):
        assert torch.equal(a[k], b[k])


def test_operator_api_autograd(oracle):
    """GaussianRasterizer (nn.Module) + autograd reproduce the raw-binding results; gradient slots line up."""
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    P, W, H = 2000, 96, 64
    cam, g = util.make_scene(P, W, H, 6, 3)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    dev = "cuda"
    t = {k: util.to_dev(v).requires_grad_(True) for k, v in g.items() if isinstance(v, np.ndarray)}
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=util.to_dev(bg),
        scale_modifier=1.0, viewmatrix=util.to_dev(cam["viewmatrix"]), projmatrix=util.to_dev(cam["projmatrix"]),
        sh_degree=3, campos=util.to_dev(cam["campos"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
    color, radii, depth, alpha = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"],
                                      scales=t["scales"], rotations=t["rotations"])
    rng = np.random.default_rng(1)
    ref = util.oracle_forward(oracle, cam, g, bg, "sh")
    solid = ref["img"]["fragile"] == 0
    wc = (rng.normal(0, 1, (3, H, W)) * solid).astype(np.float32)
    wa = (rng.normal(0, 1, (1, H, W)) * solid).astype(np.float32)
    loss = (color * util.to_dev(wc)).sum() + (alpha * util.to_dev(wa)).sum()
    loss.backward()
    want = oracle.rasterize_backward(ref, wc, np.zeros((1, H, W), np.float32), wa)
    util.assert_close("means3D.grad", t["means3D"].grad.cpu().numpy(), want["dL_dmeans3D"], max_bad_frac=2e-4)
    util.assert_close("means2D.grad", means2D.grad.cpu().numpy(), want["dL_dmean2D"], max_bad_frac=2e-4)
    util.assert_close("shs.grad", t["shs"].grad.cpu().numpy(), want["dL_dsh"], max_bad_frac=2e-4)
    util.assert_close("opacities.grad", t["opacities"].grad.cpu().numpy(), want["dL_dopacity"], max_bad_frac=2e-4)
    util.assert_close("scales.grad", t["scales"].grad.cpu().numpy(), want["dL_dscales"], max_bad_frac=2e-4)
    util.assert_close("rotations.grad", t["rotations"].grad.cpu().numpy(), want["dL_drotations"], max_bad_frac=2e-4)
    np.testing.assert_array_equal(radii.cpu().numpy(), ref["pre"]["radii"])
    assert rast.markVisible(t["means3D"].detach()).all()


def test_async_session_matches_sync_path_and_reports_overflow(oracle):
    """RasterSession (sync-free forward, fused loss gradient, in-place backward) == the reference-shaped bindings."""
    from mygauhuman_amd import parallel
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    from mygauhuman_amd.fastpath import RasterSession
    P, W, H = 6000, 144, 80
    cam, g = util.make_scene(P, W, H, 8, 3)
    bg = util.to_dev(np.array([0.2, 0.1, 0.4], np.float32))
    params = dict(means3D=util.to_dev(g["means3D"]), shs=util.to_dev(g["shs"]), opacities=util.to_dev(g["opacities"]),
                  scales=util.to_dev(g["scales"]), rotations=util.to_dev(g["rotations"]))
    camd = dict(cam, viewmatrix=util.to_dev(cam["viewmatrix"]), projmatrix=util.to_dev(cam["projmatrix"]),
                campos=util.to_dev(cam["campos"]))
    rng = np.random.default_rng(2)
    gt = util.to_dev(rng.uniform(0, 1, (3, H, W)).astype(np.float32))
    mask = util.to_dev((rng.uniform(0, 1, (1, H, W)) > 0.5).astype(np.float32))
    step = parallel.ViewParallelStep(params, 3, camd, bg)
    color, alpha, radii = step(camd, bg, gt, mask, reduce=False)
    assert not step.session.overflowed()
    e = torch.empty(0)
    R, c2, d2, a2, r2, gb, bb, ib = _C.rasterize_gaussians(bg, params["means3D"], e, params["opacities"], params["scales"],
                                                           params["rotations"], 1.0, e, camd["viewmatrix"], camd["projmatrix"],
                                                           cam["tanfovx"], cam["tanfovy"], H, W, params["shs"], 3,
                                                           camd["campos"], False, False)
    assert step.session.num_rendered() == R
    assert torch.equal(color, c2) and torch.equal(alpha, a2) and torch.equal(radii, r2)
    dc = torch.sign(c2 - gt) / c2.numel()
    da = 0.2 * (a2 - mask) / a2.numel()
    # (the step forms the loss gradient inside its backward kernel; the stand-alone loss kernel against the same expressions:)
    k_dc, k_da = step.session.alpha_mask_loss_backward(gt, mask, 0.1)
    np.testing.assert_allclose(k_dc.cpu().numpy(), dc.cpu().numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(k_da.cpu().numpy(), da.cpu().numpy(), rtol=1e-6, atol=1e-12)
    grads = _C.rasterize_gaussians_backward(bg, params["means3D"], r2, e, params["scales"], params["rotations"], 1.0, e,
                                            camd["viewmatrix"], camd["projmatrix"], cam["tanfovx"], cam["tanfovy"], dc,
                                            torch.zeros_like(a2), da, params["shs"], 3, camd["campos"], gb, R, bb, ib, a2, False)
    for name, ref in (("means3D", grads[3]), ("sh", grads[5]), ("opacity", grads[2]), ("scales", grads[6]), ("rotations", grads[7])):
        util.assert_close(name, step.grads[name].cpu().numpy(), ref.cpu().numpy(), tol=2e-5, max_bad_frac=1e-4)
    # a session that is too small reports overflow and renders only the background
    small = RasterSession(P, W, H, 16, "cuda", capacity=max(1, R // 3))
    col, _, _, _ = small.forward(params, camd, bg, 3)
    assert small.overflowed() and small.num_rendered() == R
    np.testing.assert_array_equal(col.cpu().numpy(), np.broadcast_to(bg.cpu().numpy()[:, None, None], (3, H, W)))


@pytest.mark.parametrize("P,W,H", [(5000, 150, 100), (300, 64, 48)])
def test_fp16_sh_storage_equals_fp32_on_rounded_coefficients(P, W, H):
    """fp16 SH storage (BASELINE configs[4]): halves are widened exactly on load, so every output must be BIT-identical to
    the fp32 path fed with the same (fp16-rounded) coefficients; dL_dsh comes back in fp32."""
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    from mygauhuman_amd.fastpath import RasterSession
    cam, g = util.make_scene(P, W, H, 12, 3)
    sh16 = util.to_dev(g["shs"]).half()
    sh32 = sh16.float()
    bg = util.to_dev(np.array([0.2, 0.3, 0.4], np.float32))
    d = util.to_dev
    e = torch.empty(0)
    args = lambda sh: (bg, d(g["means3D"]), e, d(g["opacities"]), d(g["scales"]), d(g["rotations"]), 1.0, e, d(cam["viewmatrix"]),  # noqa: E731
                       d(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, sh, 3, d(cam["campos"]), False, False)
    a, b = _C.rasterize_gaussians(*args(sh16)), _C.rasterize_gaussians(*args(sh32))
    assert a[0] == b[0]
    for x, y in zip(a[1:5], b[1:5]):
        assert torch.equal(x, y)
    rng = np.random.default_rng(0)
    dc, dd, da = (d(rng.normal(0, 1, s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (1, H, W)))
    bw = lambda out, sh: _C.rasterize_gaussians_backward(bg, d(g["means3D"]), out[4], e, d(g["scales"]), d(g["rotations"]), 1.0, e,  # noqa: E731
                                                         d(cam["viewmatrix"]), d(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], dc, dd, da,
                                                         sh, 3, d(cam["campos"]), out[5], out[0], out[6], out[7], out[3], False)
    ga, gb = bw(a, sh16), bw(b, sh32)
    assert ga[5].dtype == torch.float32 and ga[5].shape == (P, 16, 3)
    for x, y in zip(ga, gb):   # atomics sum in arbitrary order: equal to rounding
        util.assert_close("grad", x.cpu().numpy(), y.cpu().numpy(), tol=2e-5, max_bad_frac=1e-4)
    # the sync-free session takes the half tensor as well
    params = dict(means3D=d(g["means3D"]), shs=sh16, opacities=d(g["opacities"]), scales=d(g["scales"]), rotations=d(g["rotations"]))
    camd = dict(cam, viewmatrix=d(cam["viewmatrix"]), projmatrix=d(cam["projmatrix"]), campos=d(cam["campos"]))
    s = RasterSession(P, W, H, 16, "cuda", capacity=a[0] + 1000)
    col, dep, alp, rad = s.forward(params, camd, bg, 3)
    assert torch.equal(col, a[1]) and torch.equal(rad, a[4])
    with pytest.raises(RuntimeError):   # only the 16-coefficient layout has a half path
        _C.rasterize_gaussians(*args(sh16[:, :9].contiguous()))


def test_deterministic_backward_and_per_stream_knobs(oracle):
    """(1) "deterministic" = fixed-order reduction of the gradient rows: two backward runs give the same BITS, and the values agree
    with the atomic mode to summation order.  (2) Re-entrancy: two threads drive forward + backward on two streams with
    different knobs set per stream (binning back-end, culling, wave counts, reduction) at the same time; each must produce
    exactly what it produces alone (no knob leaks between streams: integer state bit-identical, images bit-identical)."""
    import threading

    from mygauhuman_amd import _lib
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    P, W, H = 9000, 208, 144
    cam, g = util.make_scene(P, W, H, 21, 3, 0.03, 0.02)
    d = util.to_dev
    bg = d(np.array([0.3, 0.2, 0.1], np.float32))
    e = torch.empty(0)
    T = {k: d(g[k]) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    cm = {k: d(cam[k]) for k in ("viewmatrix", "projmatrix", "campos")}
    rng = np.random.default_rng(5)
    dc, dd, da = (d(rng.normal(0, 1, s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (1, H, W)))

    def frame():
        o = _C.rasterize_gaussians(bg, T["means3D"], e, T["opacities"], T["scales"], T["rotations"], 1.0, e, cm["viewmatrix"],
                                   cm["projmatrix"], cam["tanfovx"], cam["tanfovy"], H, W, T["shs"], 3, cm["campos"], False, False)
        gr = _C.rasterize_gaussians_backward(bg, T["means3D"], o[4], e, T["scales"], T["rotations"], 1.0, e, cm["viewmatrix"],
                                             cm["projmatrix"], cam["tanfovx"], cam["tanfovy"], dc, dd, da, T["shs"], 3, cm["campos"],
                                             o[5], o[0], o[6], o[7], o[3], False)
        ranges = _C.query_state("RANGES", P, o[0], W, H, o[5], o[6], o[7])
        plist = _C.query_state("POINT_LIST", P, o[0], W, H, o[5], o[6], o[7])
        return o, gr, ranges, plist

    # ---- (1) determinism
    ref = frame()
    _lib.set_tuning("deterministic", 1)
    try:
        a, b = frame(), frame()
    finally:
        _lib.set_tuning("deterministic", 0)
    for x, y, z in zip(a[1], b[1], ref[1]):
        assert torch.equal(x, y)                                         # run-to-run bit-identical
        util.assert_close("det vs atomics", x.cpu().numpy(), z.cpu().numpy(), tol=2e-5, max_bad_frac=1e-4)
    want = oracle.rasterize_backward(util.oracle_forward(oracle, cam, g, bg.cpu().numpy(), "sh"), dc.cpu().numpy(), dd.cpu().numpy(),
                                     da.cpu().numpy())
    util.assert_close("det vs oracle", a[1][3].cpu().numpy(), want["dL_dmeans3D"], max_bad_frac=2e-4)

    # ---- (2) two streams, two threads, different knobs
    knobs = [dict(binning_mode=0, blend_fwd_waves=1, blend_bwd_waves=2, blend_bwd_reduce=1 if util.has_experiments() else 0),
             dict(binning_mode=1, tile_cull=1, blend_fwd_waves=4, blend_bwd_waves=4, blend_bwd_reduce=0, deterministic=1)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st, kn in zip(streams, knobs):
        for k, v in kn.items():
            _lib.set_tuning(k, v, stream=st)
    torch.cuda.synchronize()

    def run(i, out, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                out[i] = frame()
        streams[i].synchronize()

    alone = [None, None]
    for i in range(2):
        run(i, alone, 1)
    both = [None, None]
    th = [threading.Thread(target=run, args=(i, both, 4)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    try:
        for i in range(2):
            (o1, g1, r1, p1), (o2, g2, r2, p2) = alone[i], both[i]
            assert o1[0] == o2[0] and torch.equal(r1, r2) and torch.equal(p1, p2), i   # the stream's own binning back-end ran
            for x, y in zip(o1[1:5], o2[1:5]):
                assert torch.equal(x, y), i
        assert int((alone[0][2][:, 1] - alone[0][2][:, 0]).sum()) > int((alone[1][2][:, 1] - alone[1][2][:, 0]).sum())  # radix: no culling
        for x, y in zip(alone[1][1], both[1][1]):
            assert torch.equal(x, y)   # stream 1 is deterministic: concurrent work on the other stream changes nothing
    finally:
        for st in streams:
            _lib.clear_stream_tuning(st)


@pytest.mark.parametrize("P,W,H,scale", [(9000, 208, 144, 0.03), (20000, 400, 304, 0.01), (6000, 200, 56, 0.05), (3000, 17, 200, 0.1),
                                         (500, 16, 16, 0.2), (3000, 16, 400, 0.1), (3000, 400, 16, 0.1)])
def test_tile_visiting_order_does_not_change_results(P, W, H, scale):
    """Options::tile_order (0 natural, 1 longest lists first, 2 / 3 blocks of 2 x 2 / 4 x 2 tiles by summed length, a block per
    XCD): the blend kernels only VISIT the tiles in another order (and on other XCDs); images, per-pixel state and -- with the
    fixed-order reduction -- gradients keep their bits.  Ragged tile grids (odd numbers of tile rows / columns, one tile, a
    one-tile-wide strip) exercise the padding slots of the block modes."""
    from mygauhuman_amd import _lib
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    cam, g = util.make_scene(P, W, H, 13, 3, scale, 0.02)
    d = util.to_dev
    bg = d(np.array([0.3, 0.2, 0.1], np.float32))
    e = torch.empty(0)
    T = {k: d(g[k]) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
    cm = {k: d(cam[k]) for k in ("viewmatrix", "projmatrix", "campos")}
    rng = np.random.default_rng(5)
    dc, dd, da = (d(rng.normal(0, 1, s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (1, H, W)))
    res = {}
    _lib.set_tuning("deterministic", 1)
    try:
        for mode in (0, 1, 2, 3):
            _lib.set_tuning("tile_order", mode)
            o = _C.rasterize_gaussians(bg, T["means3D"], e, T["opacities"], T["scales"], T["rotations"], 1.0, e, cm["viewmatrix"],
                                       cm["projmatrix"], cam["tanfovx"], cam["tanfovy"], H, W, T["shs"], 3, cm["campos"], False, False)
            gr = _C.rasterize_gaussians_backward(bg, T["means3D"], o[4], e, T["scales"], T["rotations"], 1.0, e, cm["viewmatrix"],
                                                 cm["projmatrix"], cam["tanfovx"], cam["tanfovy"], dc, dd, da, T["shs"], 3,
                                                 cm["campos"], o[5], o[0], o[6], o[7], o[3], False)
            res[mode] = ([o[1], o[2], o[3], _C.query_state("N_CONTRIB", P, o[0], W, H, o[5], o[6], o[7]),
                          _C.query_state("FINAL_T", P, o[0], W, H, o[5], o[6], o[7])], list(gr))
    finally:
        _lib.set_tuning("deterministic", 0)
        _lib.set_tuning("tile_order", _lib.DEFAULT_TILE_ORDER)
    for mode in (1, 2, 3):
        for a, b in zip(res[0][0], res[mode][0]):
            assert torch.equal(a, b), mode
        for a, b in zip(res[0][1], res[mode][1]):
            assert torch.equal(a, b), mode
    with pytest.raises(_lib.GsrError):
        _lib.set_tuning("tile_order", 4)
