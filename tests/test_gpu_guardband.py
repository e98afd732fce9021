"""Out-of-bounds WRITES of the HIP kernels, looked for with guard bands (the GPU pool offers no device
sanitizer).  Every tensor the bindings allocate -- outputs, gradient outputs and the three opaque scratch buffers the library
carves its own arrays from -- is placed inside a larger byte buffer whose margins (4 KB in front, >= 4 KB behind, filled with
0xA5) are checked after the call.  A kernel that stores one element before or after an array it was given, for any of the
ragged / one-tile / long-list scenes and every wave / reduction / binning configuration, fails here; a write further away than
the margin, or a stray read, is not seen.  (Writes INSIDE a scratch buffer to the wrong private array are what the bit-exact
state comparisons of test_gpu_rasterizer.py catch.)"""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_rasterizer import CASES, _bg

pytestmark = pytest.mark.gpu

GUARD = 4096
PATTERN = 0xA5


class GuardedTorch:
    """Stands in for the `torch` module inside a binding module: empty / zeros / empty_like on a HIP device come out of a
    guarded byte buffer; everything else is torch."""

    def __init__(self):
        self.bufs = []  # (whole uint8 buffer, first payload byte, payload bytes, tag)

    def __getattr__(self, name):
        return getattr(torch, name)

    def _alloc(self, shape, dtype, device, fill, tag):
        dtype = dtype or torch.float32
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        shape = tuple(int(s) for s in shape)
        if device is None or torch.device(device).type != "cuda":
            return (torch.zeros if fill else torch.empty)(shape, dtype=dtype, device=device)
        item = torch.empty((), dtype=dtype).element_size()
        n = item
        for s in shape:
            n *= s
        n = int(n)
        total = GUARD + (n + 255) // 256 * 256 + GUARD
        whole = torch.full((total,), PATTERN, dtype=torch.uint8, device=device)
        payload = whole[GUARD:GUARD + n]
        if fill:
            payload.zero_()
        self.bufs.append((whole, GUARD, n, tag))
        return payload.view(dtype).view(shape)

    def empty(self, *shape, dtype=None, device=None, **kw):
        return self._alloc(shape, dtype, device, False, "empty")

    def zeros(self, *shape, dtype=None, device=None, **kw):
        return self._alloc(shape, dtype, device, True, "zeros")

    def empty_like(self, t, **kw):
        return self._alloc(tuple(t.shape), t.dtype, t.device, False, "empty_like")

    def check(self, what):
        torch.cuda.synchronize()
        n_checked = 0
        for whole, first, n, tag in self.bufs:
            front, back = whole[:first], whole[first + n:]
            bad_f, bad_b = (front != PATTERN).nonzero(), (back != PATTERN).nonzero()
            assert bad_f.numel() == 0, f"{what}: write {first - int(bad_f.max())} bytes IN FRONT of a {n}-byte {tag} tensor"
            assert bad_b.numel() == 0, f"{what}: write {int(bad_b.min())} bytes BEHIND a {n}-byte {tag} tensor"
            n_checked += 1
        self.bufs = []
        return n_checked


@pytest.fixture()
def guarded(monkeypatch):
    from mygauhuman_amd import loss_utils
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    g = GuardedTorch()
    monkeypatch.setattr(_C, "torch", g)
    monkeypatch.setattr(loss_utils, "torch", g)
    return g


CONFIGS = [(4, 3, "bucket_tight"), (4, 0, "bucket_tight"), (1, 0, "bucket"), (2, 1, "radix"), (4, 2, "bucket_tight"), (2, 0, "bucket_tight")]


@pytest.fixture(params=CONFIGS, ids=lambda c: f"waves{c[0]}red{c[1]}{c[2]}")
def config(request):
    from mygauhuman_amd import _lib
    w, red, binning = request.param
    util.skip_unless_experiments(red in (1, 2))
    _lib.set_tuning("blend_fwd_waves", w)
    _lib.set_tuning("blend_bwd_waves", w)
    _lib.set_tuning("blend_bwd_reduce", red)
    _lib.check(_lib.lib.gsr_set_binning_mode(_lib.BINNING_GLOBAL_RADIX if binning == "radix" else _lib.BINNING_TILE_BUCKET), "mode")
    util.set_tile_cull(binning == "bucket_tight")
    yield request.param
    _lib.set_tuning("blend_fwd_waves", 4)
    _lib.set_tuning("blend_bwd_waves", 4)
    _lib.set_tuning("blend_bwd_reduce", _lib.DEFAULT_BWD_REDUCE)
    _lib.lib.gsr_set_binning_mode(_lib.DEFAULT_BINNING)
    util.set_tile_cull(_lib.DEFAULT_TILE_CULL)


RAGGED = CASES + [
    (400, 1, 1, 11, 0, 0.3, 0.0),        # one pixel
    (2500, 517, 300, 12, 3, 0.02, 0.1),  # the image size of the SSIM case that preceded round 2's abort
    (5000, 33, 17, 13, 2, 0.1, 0.0),     # two ragged tile rows / columns, long lists
]


@pytest.mark.parametrize("case", RAGGED, ids=[f"P{c[0]}_{c[1]}x{c[2]}" for c in RAGGED])
@pytest.mark.parametrize("mode", ["sh", "precomp"])
def test_rasterizer_writes_stay_inside_their_arrays(guarded, config, case, mode):
    P, W, H, seed, deg, scale, behind = case
    cam, g = util.make_scene(P, W, H, seed, deg, scale, behind)
    rng = np.random.default_rng(seed)
    f = util.hip_forward(cam, g, _bg(seed), mode)
    assert guarded.check("forward") >= 7  # colour, depth, alpha, radii + three scratch buffers
    dc, dd, da = (rng.normal(0, 1, (c, H, W)).astype(np.float32) for c in (3, 1, 1))
    util.hip_backward(f, dc, dd, da)
    assert guarded.check("backward") >= 7


@pytest.mark.parametrize("case", [RAGGED[0], RAGGED[4], RAGGED[6], RAGGED[8], RAGGED[9]], ids=lambda c: f"P{c[0]}_{c[1]}x{c[2]}")
@pytest.mark.parametrize("det", [0, 1])
def test_fused_feature_pass_and_async_entry_writes_stay_inside(guarded, case, det):
    from mygauhuman_amd import _lib
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    P, W, H, seed, deg, scale, behind = case
    cam, g = util.make_scene(P, W, H, seed, deg, scale, behind)
    dev = "cuda"
    t = {k: util.to_dev(v, dev) for k, v in g.items() if isinstance(v, np.ndarray)}
    e = torch.empty(0)
    extra = torch.rand((P, 18), device=dev) if det == 0 else None   # the deterministic (test) mode is built for the plain pass only
    args = (util.to_dev(_bg(seed)), t["means3D"], t["colors"], t["opacities"], e, e, 1.0, t["cov3D"], util.to_dev(cam["viewmatrix"]),
            util.to_dev(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, e, 0, util.to_dev(cam["campos"]), False, False)
    _lib.set_tuning("deterministic", det)
    try:
        for sync_free in (False, True):
            if sync_free:
                out = _C.rasterize_gaussians_async(*args, extra=extra, capacity=max(4096, 40 * P))
                R, color, depth, alpha, radii, geomB, binB, imgB, out_extra, watch = out
                _C.AsyncCapacity.check(watch)
            else:
                R, color, depth, alpha, radii, geomB, binB, imgB = _C.rasterize_gaussians(*args, extra=extra)[:8]
            assert guarded.check("fused forward") >= 7
            grads = [torch.rand((3, H, W), device=dev) if k in (0, 2, 5) else None for k in range(6)] if extra is not None else None
            _C.rasterize_gaussians_backward(args[0], t["means3D"], radii, t["colors"], e, e, 1.0, t["cov3D"], args[8], args[9],
                                            cam["tanfovx"], cam["tanfovy"], torch.rand((3, H, W), device=dev),
                                            torch.rand((1, H, W), device=dev), torch.rand((1, H, W), device=dev), e, 0, args[16],
                                            geomB, R, binB, imgB, alpha, False, extra=extra, dL_dout_extra=grads)
            assert guarded.check("fused backward") >= 7
    finally:
        _lib.set_tuning("deterministic", 0)


@pytest.mark.parametrize("shape", [(1, 3, 64, 64), (1, 3, 97, 131), (3, 40, 23), (2, 3, 11, 5), (1, 1, 300, 517), (1, 1, 1)])
def test_ssim_writes_stay_inside(guarded, shape):
    from mygauhuman_amd import loss_utils
    img2 = torch.rand(shape, device="cuda")
    img1 = (img2 + 0.1).clamp(0, 1).requires_grad_(True)
    v = loss_utils.ssim(img1, img2)
    assert guarded.check("ssim forward") >= 4
    (1.0 - v).backward()
    assert guarded.check("ssim backward") >= 1
