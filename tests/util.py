"""Shared helpers of the parity tests: seeded scenes, the HIP path through the raw `_C` bindings, the oracle path."""
import numpy as np
import torch

from mygauhuman_amd import synthetic


def make_scene(P, W, H, seed=0, deg=3, scale=0.02, behind_frac=0.0):
    cam, g = synthetic.uniform_scene(P, W, H, seed=seed, sh_degree=deg, log_scale_mean=float(np.log(scale)))
    rng = np.random.default_rng(seed + 77)
    if behind_frac > 0 and P > 0:  # some Gaussians behind / too close to the camera -> culled
        k = max(1, int(P * behind_frac))
        g["means3D"][:k, 2] = rng.uniform(-1.0, 0.2, k).astype(np.float32)
    A = rng.normal(0, scale, (P, 3, 3)).astype(np.float32)
    g["cov3D"] = (np.stack([(a @ a.T + 1e-6 * np.eye(3, dtype=np.float32))[np.triu_indices(3)] for a in A]).astype(np.float32)
                  if P else np.zeros((0, 6), np.float32))
    return cam, g


def oracle_forward(oracle, cam, g, bg, mode):
    kw = dict(scale_modifier=1.0)
    if mode == "sh":
        kw.update(scales=g["scales"], rotations=g["rotations"], shs=g["shs"], degree=g["sh_degree"])
    else:
        kw.update(cov3D_precomp=g["cov3D"], colors_precomp=g["colors"])
    return oracle.rasterize_forward(g["means3D"], g["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                                    cam["W"], cam["H"], cam["tanfovx"], cam["tanfovy"], bg, **kw)


def to_dev(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def hip_forward(cam, g, bg, mode, debug=False, dev="cuda"):
    """Forward through the raw binding (same 19 positional args as the reference's _C.rasterize_gaussians)."""
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    e = torch.empty(0)
    t = {k: to_dev(v, dev) for k, v in g.items() if isinstance(v, np.ndarray)}
    args = dict(bg=to_dev(bg, dev), means3D=t["means3D"], opac=t["opacities"], view=to_dev(cam["viewmatrix"], dev),
                proj=to_dev(cam["projmatrix"], dev), campos=to_dev(cam["campos"], dev))
    if mode == "sh":
        colors, scales, rots, cov, sh, deg = e, t["scales"], t["rotations"], e, t["shs"], g["sh_degree"]
    else:
        colors, scales, rots, cov, sh, deg = t["colors"], e, e, t["cov3D"], e, 0
    out = _C.rasterize_gaussians(args["bg"], args["means3D"], colors, args["opac"], scales, rots, 1.0, cov, args["view"],
                                 args["proj"], cam["tanfovx"], cam["tanfovy"], cam["H"], cam["W"], sh, deg,
                                 args["campos"], False, debug)
    R, color, depth, alpha, radii, geomB, binB, imgB = out
    return dict(R=R, color=color, depth=depth, alpha=alpha, radii=radii, geom=geomB, bin=binB, img=imgB, args=args,
                colors=colors, scales=scales, rots=rots, cov=cov, sh=sh, deg=deg, P=t["means3D"].shape[0], W=cam["W"],
                H=cam["H"], cam=cam)


def hip_query(f, what):
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    return _C.query_state(what, f["P"], f["R"], f["W"], f["H"], f["geom"], f["bin"], f["img"]).cpu().numpy()


def hip_backward(f, dL_dcolor, dL_ddepth, dL_dalpha, debug=False):
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    a, cam = f["args"], f["cam"]
    dev = a["means3D"].device
    out = _C.rasterize_gaussians_backward(
        a["bg"], a["means3D"], f["radii"], f["colors"], f["scales"], f["rots"], 1.0, f["cov"], a["view"], a["proj"],
        cam["tanfovx"], cam["tanfovy"], to_dev(dL_dcolor, dev), to_dev(dL_ddepth, dev), to_dev(dL_dalpha, dev), f["sh"],
        f["deg"], a["campos"], f["geom"], f["R"], f["bin"], f["img"], f["alpha"], debug)
    names = ["dL_dmean2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
    return {n: o.cpu().numpy() for n, o in zip(names, out)}


def assert_close(name, got, want, tol=1e-4, mask=None, max_bad_frac=0.0, atol=0.0, outer_tol=None):
    """|got - want| <= tol * max(|want|, scale) + atol elementwise, where scale = the tensor's OWN 99.9th percentile
    magnitude (sums of many +/- terms are accurate relative to the terms, not to the possibly cancelled result).  There is no
    built-in absolute floor: a tensor of tiny gradients is held to tol relative to its own size (pass atol where an
    absolute floor is meant).
    max_bad_frac > 0 lets that fraction of the elements sit outside `tol` (float atomics in arbitrary order put the odd element
    on the tolerance) -- but NO element, without exception, may be further off than outer_tol (default 10 x tol) by the same
    measure: a defect that hits one Gaussian in thousands with an arbitrarily wrong value fails whatever the fraction."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if got.size == 0:
        return
    scale = float(np.percentile(np.abs(want), 99.9))
    if scale == 0.0:
        scale = float(np.abs(want).max())
    err = np.abs(got - want)
    bound = tol * np.maximum(np.abs(want), scale) + atol
    bad = err > bound
    if mask is not None:
        bad &= mask
    far = err > (10.0 * tol if outer_tol is None else outer_tol) * np.maximum(np.abs(want), scale) + atol
    if mask is not None:
        far &= mask
    assert not far.any(), (f"{name}: {far.sum()} / {far.size} elements beyond the outer bound; worst index "
                           f"{np.unravel_index(np.argmax(np.where(far, err, 0)), err.shape)} err {err[far].max():.3e} "
                           f"want {want.flat[np.argmax(np.where(far, err, 0))]:.3e} (scale {scale:.3e})")
    frac = bad.mean()
    assert frac <= max_bad_frac, (f"{name}: {bad.sum()} / {bad.size} elements off; max err {err[bad].max():.3e} "
                                  f"(scale {scale:.3e}, tol {tol:g}, atol {atol:g})")


def set_tile_cull(on):
    """Tuning knob "tile_cull" of the tile-bucket binning back-end (library default: on)."""
    from mygauhuman_amd import _lib
    _lib.set_tuning("tile_cull", int(bool(on)))


def assert_lists_are_sublists(f, ref_bin, tiles):
    """Tight tile culling: every tile's list must be a subsequence (same order) of the reference list of that tile."""
    ranges = hip_query(f, "RANGES").view(np.uint32).reshape(-1, 2).astype(np.int64)
    pl = hip_query(f, "POINT_LIST").view(np.uint32)
    rr, rp = ref_bin["ranges"].astype(np.int64), ref_bin["point_list"]
    kept = 0
    for t in range(tiles):
        mine = pl[ranges[t, 0]:ranges[t, 1]]
        ref = rp[rr[t, 0]:rr[t, 1]]
        assert len(mine) <= len(ref)
        pos = {int(g): i for i, g in enumerate(ref)}   # a Gaussian appears at most once per tile
        idx = [pos[int(g)] for g in mine]              # KeyError = an instance the reference does not have
        assert all(a < b for a, b in zip(idx, idx[1:])), f"tile {t}: order differs from the reference list"
        kept += len(mine)
    return kept


class GetterOnlyModel:
    """A view of a HumanGaussianModel that offers render() only what the reference's GaussianModel offers: the property getters
    (no frame_activations(), no direct access to the two SH parameter tensors) -- render() then takes its torch-op path."""
    _HIDDEN = ("frame_activations", "_features_dc", "_features_rest")

    def __init__(self, model):
        object.__setattr__(self, "_m", model)

    def __getattr__(self, name):
        if name in GetterOnlyModel._HIDDEN:
            raise AttributeError(name)
        return getattr(object.__getattribute__(self, "_m"), name)


def has_experiments():
    """Was libgsr.so built with GSR_BUILD_EXPERIMENTS (the not-adopted kernels)?  The default build is not; parametrised tests skip
    the knob values that select them (python -m mygauhuman_amd.build --experiments && pytest -m gpu runs them all)."""
    from mygauhuman_amd import _lib
    return bool(_lib.lib.gsr_has_experiments())


def skip_unless_experiments(is_experimental):
    import pytest
    if is_experimental and not has_experiments():
        pytest.skip("experiment kernel: not in the default build (python -m mygauhuman_amd.build --experiments)")
