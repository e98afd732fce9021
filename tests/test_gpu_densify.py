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
