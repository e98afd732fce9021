"""Densify / prune / optimizer-state surgery on the GPU (mygauhuman_amd.densify: row plans applied by gsr_gather_rows,
SMPL-distance prune through gsr_knn_nearest) against the numpy restatement of the reference (oracle/densify_oracle.py).
Copied rows must be bit-identical; recomputed rows (split children) agree to fp32 rounding."""
import copy

import numpy as np
import pytest
import torch

from tests.test_densify_cpu import SHAPES, assert_state_equal, make_state

pytestmark = pytest.mark.gpu


def to_model(st, with_adam=True):
    from mygauhuman_amd import densify
    from mygauhuman_amd.scene_model import HumanGaussianModel
    m = HumanGaussianModel(3, device="cuda")
    for g in densify.GROUPS:
        setattr(m, densify.ATTR[g], torch.nn.Parameter(torch.from_numpy(st["params"][g]).cuda()))
    densify.training_setup(m, {g: 1e-3 for g in densify.GROUPS})
    if with_adam:
        for g in densify.GROUPS:
            p = getattr(m, densify.ATTR[g])
            m.optimizer.state[p] = dict(step=torch.tensor(3.0, device="cuda"), exp_avg=torch.from_numpy(st["exp_avg"][g]).cuda(),
                                        exp_avg_sq=torch.from_numpy(st["exp_avg_sq"][g]).cuda())
    m.xyz_gradient_accum = torch.from_numpy(st["xyz_gradient_accum"]).cuda()
    m.denom = torch.from_numpy(st["denom"]).cuda()
    m.max_radii2D = torch.from_numpy(st["max_radii2D"]).cuda()
    return m


def from_model(m):
    from mygauhuman_amd import densify
    out = dict(params={}, exp_avg={}, exp_avg_sq={})
    for g in densify.GROUPS:
        p = getattr(m, densify.ATTR[g])
        out["params"][g] = p.detach().cpu().numpy()
        st = m.optimizer.state.get(p)
        assert st is not None and m.optimizer.param_groups[densify.GROUPS.index(g)]["params"][0] is p
        out["exp_avg"][g] = st["exp_avg"].cpu().numpy()
        out["exp_avg_sq"][g] = st["exp_avg_sq"].cpu().numpy()
    for s in ("xyz_gradient_accum", "denom", "max_radii2D"):
        out[s] = getattr(m, s).cpu().numpy()
    return out


def test_prune_points_moves_everything_bit_exactly():
    from mygauhuman_amd import densify
    from oracle import densify_oracle as do
    st = make_state(5000, 11)
    mask = np.random.default_rng(1).uniform(0, 1, 5000) < 0.4
    want = copy.deepcopy(st)
    do.prune_points(want, mask)
    m = to_model(st)
    densify.prune_points(m, torch.from_numpy(mask).cuda())
    assert_state_equal(from_model(m), want)
    # the optimizer still steps on the new parameters
    for g in densify.GROUPS:
        p = getattr(m, densify.ATTR[g])
        p.grad = torch.ones_like(p)
    m.optimizer.step()
    empty = torch.zeros(m._xyz.shape[0], dtype=torch.bool, device="cuda")
    empty[:] = True
    densify.prune_points(m, empty)      # prune everything: zero rows everywhere, no crash
    assert m._xyz.shape == (0, 3) and m.optimizer.state[m._xyz]["exp_avg"].shape == (0, 3)


@pytest.mark.parametrize("seed,max_screen", [(21, 20), (22, 0)])
def test_densify_and_prune_matches_reference_sequence(oracle, seed, max_screen):
    from mygauhuman_amd import densify
    from oracle import densify_oracle as do
    P, extent, thr, min_op = 6000, 2.0, 4e-4, 0.25
    st = make_state(P, seed)
    rng = np.random.default_rng(seed + 1)
    verts = rng.uniform(-1, 1, (700, 3)).astype(np.float32)
    st["params"]["xyz"] = (verts[rng.integers(0, 700, P)] + rng.normal(0, 0.03, (P, 3))).astype(np.float32)
    st["denom"] = np.maximum(st["denom"], rng.integers(0, 2, (P, 1))).astype(np.float32)   # a few zero denominators -> NaN path
    unit = rng.normal(0, 1, (4 * P, 3)).astype(np.float32)

    def dist_fn(xyz):
        ids = oracle.nearest_vertex(xyz, verts)
        return oracle.nearest_dist(xyz, verts, ids)

    want = do.densify_and_prune(copy.deepcopy(st), thr, min_op, extent, max_screen, 0.01, unit, dist_fn)
    m = to_model(st)
    densify.densify_and_prune(m, thr, min_op, extent, max_screen, t_vertices=torch.from_numpy(verts).cuda(),
                              unit_samples=torch.from_numpy(unit).cuda())
    got = from_model(m)
    n = want["params"]["xyz"].shape[0]
    assert got["params"]["xyz"].shape[0] == n and n != P
    for g in do.GROUPS:
        np.testing.assert_allclose(got["params"][g], want["params"][g], rtol=2e-6, atol=2e-7, err_msg=g)
        np.testing.assert_array_equal(got["exp_avg"][g], want["exp_avg"][g], err_msg=g)
        np.testing.assert_array_equal(got["exp_avg_sq"][g], want["exp_avg_sq"][g], err_msg=g)
    for g in ("f_dc", "f_rest", "opacity", "rotation", "normal", "albedo", "roughness"):   # pure copies: bit-exact
        np.testing.assert_array_equal(got["params"][g], want["params"][g], err_msg=g)
    for s in ("xyz_gradient_accum", "denom", "max_radii2D"):
        np.testing.assert_array_equal(got[s], want[s], err_msg=s)
        assert not got[s].any()


def test_gather_rows_rejects_bad_arguments_and_cpu_tensors():
    from mygauhuman_amd import _lib, densify
    st = make_state(16, 0)
    m = to_model(st)
    with pytest.raises(_lib.GsrError):
        _lib.check(_lib.lib.gsr_gather_rows(0, None, None, None, None, 4, None, None), "gsr_gather_rows")
    for g in densify.GROUPS:
        setattr(m, densify.ATTR[g], torch.nn.Parameter(getattr(m, densify.ATTR[g]).detach().cpu()))
    with pytest.raises(RuntimeError):
        densify.apply_plan(m, torch.arange(4, dtype=torch.int32), reset_stats=False)
    assert set(SHAPES) == set(densify.GROUPS)


def _twin_state(P, seed):
    """A cloud in which half of the Gaussians have a near-identical twin (their k-NN partner, KL << 0.1) and the rest have
    far neighbours (KL >> 0.4), so that all three KL operations select a real subset."""
    st = make_state(P, seed)
    rng = np.random.default_rng(seed + 7)
    p = st["params"]
    h = P // 4
    p["xyz"] = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    tw = slice(h, 2 * h)                              # rows h..2h-1 are twins of rows 0..h-1
    s = np.exp(p["scaling"][:h])
    p["xyz"][tw] = p["xyz"][:h] + (0.05 * s * rng.normal(0, 1, (h, 3))).astype(np.float32)
    p["scaling"][tw] = p["scaling"][:h] + rng.normal(0, 0.01, (h, 3)).astype(np.float32)
    p["rotation"][tw] = p["rotation"][:h] + rng.normal(0, 0.005, (h, 4)).astype(np.float32)
    st["denom"] = np.maximum(st["denom"], 1).astype(np.float32)
    return st


def _assert_kl_state(got, want, do):
    assert got["params"]["xyz"].shape[0] == want["params"]["xyz"].shape[0]
    for g in do.GROUPS:
        np.testing.assert_allclose(got["params"][g], want["params"][g], rtol=3e-6, atol=3e-7, err_msg=g)
        np.testing.assert_array_equal(got["exp_avg"][g], want["exp_avg"][g], err_msg=g)
        np.testing.assert_array_equal(got["exp_avg_sq"][g], want["exp_avg_sq"][g], err_msg=g)
    for s in ("xyz_gradient_accum", "denom", "max_radii2D"):
        assert not got[s].any() and got[s].shape == want[s].shape


@pytest.mark.parametrize("op", ["clone", "split", "merge"])
def test_kl_variants_match_restatement(op):
    """kl_densify_and_clone / kl_densify_and_split / kl_merge (pairs from gsr_knn_self, rows moved by gsr_gather_rows) against
    the numpy restatement with a brute-force k-NN."""
    from mygauhuman_amd import densify
    from oracle import densify_oracle as do
    P, extent, thr = 2400, 2.0, 8e-4
    st = _twin_state(P, 31)
    rng = np.random.default_rng(3)
    unit = rng.normal(0, 1, (2 * P, 3)).astype(np.float32)
    grads = (st["xyz_gradient_accum"] / st["denom"]).astype(np.float32)
    want = copy.deepcopy(st)
    # the KL threshold goes into the middle of the gap of the pair divergences around the reference's default (0.4 / 0.1), so
    # that fp32 (product) vs fp64 (restatement) rounding cannot flip a decision
    kl0 = np.sort(do.kl_pairs(st)[0])
    nominal = 0.1 if op == "merge" else 0.4
    j = int(np.searchsorted(kl0, nominal))
    edge = float(0.5 * (kl0[j - 1] + kl0[j]))
    assert kl0[j] - kl0[j - 1] > 1e-4 * edge and 0.5 * nominal < edge < 2 * nominal
    if op == "clone":
        sel, kl = do.kl_densify_and_clone(want, grads, thr, extent, 0.01, unit, kl_threshold=edge)
    elif op == "split":
        sel, kl = do.kl_densify_and_split(want, grads, thr, extent, 0.01, unit, kl_threshold=edge)
    else:
        sel, kl = do.kl_merge(want, grads, thr, extent, 0.01, kl_threshold=edge)
    assert 20 < sel.sum() < P // 2, sel.sum()
    m = to_model(st)
    g = torch.from_numpy(grads).cuda()
    if op == "clone":
        got_sel = densify.kl_densify_and_clone(m, g, thr, extent, kl_threshold=edge, unit_samples=torch.from_numpy(unit).cuda())
    elif op == "split":
        got_sel = densify.kl_densify_and_split(m, g, thr, extent, kl_threshold=edge, unit_samples=torch.from_numpy(unit).cuda())
    else:
        got_sel = densify.kl_merge(m, g, thr, extent, kl_threshold=edge)
    np.testing.assert_array_equal(got_sel.cpu().numpy(), sel)
    _assert_kl_state(from_model(m), want, do)
    # the optimizer steps on the re-registered parameters
    for grp in densify.GROUPS:
        p = getattr(m, densify.ATTR[grp])
        p.grad = torch.ones_like(p)
    m.optimizer.step()


def test_kl_merge_without_candidates_is_a_no_op():
    from mygauhuman_amd import densify
    st = _twin_state(800, 5)
    m = to_model(st)
    before = from_model(m)
    sel = densify.kl_merge(m, torch.zeros((800, 1), device="cuda"), 1.0, 2.0)   # no gradient reaches the threshold
    assert int(sel.sum()) == 0
    assert_state_equal(from_model(m), before)
