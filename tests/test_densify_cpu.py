"""Row plans of mygauhuman_amd.densify (device-agnostic tensor code) against the numpy restatement of the reference's
prune / clone / split (oracle/densify_oracle.py).  The plans are applied here with plain torch indexing; the HIP gather that
applies them in the product is covered by tests/test_gpu_densify.py."""
import copy

import numpy as np
import pytest
import torch

from mygauhuman_amd import densify
from oracle import densify_oracle as do

SHAPES = dict(xyz=(3,), f_dc=(1, 3), f_rest=(15, 3), opacity=(1,), scaling=(3,), rotation=(4,), normal=(3,), albedo=(3,), roughness=(1,))


def make_state(P, seed):
    rng = np.random.default_rng(seed)
    params = {g: rng.normal(0, 1, (P,) + s).astype(np.float32) for g, s in SHAPES.items()}
    params["scaling"] = rng.normal(-4.0, 0.8, (P, 3)).astype(np.float32)
    st = dict(params=params, exp_avg={g: rng.normal(0, 1, v.shape).astype(np.float32) for g, v in params.items()},
              exp_avg_sq={g: rng.uniform(0, 1, v.shape).astype(np.float32) for g, v in params.items()},
              xyz_gradient_accum=rng.uniform(0, 3e-3, (P, 1)).astype(np.float32),
              denom=rng.integers(0, 4, (P, 1)).astype(np.float32), max_radii2D=rng.uniform(0, 30, P).astype(np.float32))
    return st


def apply_plan_torch(state, plan, reset_stats):
    """What gsr_gather_rows does, in numpy."""
    plan = plan.numpy().astype(np.int64)
    new = (plan & densify.NEW_ROW) != 0
    rows = plan & (densify.NEW_ROW - 1)
    out = copy.deepcopy(state)
    for g in do.GROUPS:
        out["params"][g] = state["params"][g][rows]
        for m in ("exp_avg", "exp_avg_sq"):
            v = state[m][g][rows].copy()
            v[new] = 0
            out[m][g] = v
    for s in ("xyz_gradient_accum", "denom", "max_radii2D"):
        out[s] = np.zeros((len(rows),) + state[s].shape[1:], np.float32) if reset_stats else state[s][rows]
    return out


def assert_state_equal(a, b):
    for g in do.GROUPS:
        np.testing.assert_array_equal(a["params"][g], b["params"][g], err_msg=g)
        np.testing.assert_array_equal(a["exp_avg"][g], b["exp_avg"][g], err_msg=g)
        np.testing.assert_array_equal(a["exp_avg_sq"][g], b["exp_avg_sq"][g], err_msg=g)
    for s in ("xyz_gradient_accum", "denom", "max_radii2D"):
        np.testing.assert_array_equal(a[s], b[s], err_msg=s)


def test_prune_plan_matches_reference_semantics():
    st = make_state(500, 0)
    mask = np.random.default_rng(1).uniform(0, 1, 500) < 0.3
    want = copy.deepcopy(st)
    do.prune_points(want, mask)
    got = apply_plan_torch(st, densify.plan_prune(torch.from_numpy(mask)), reset_stats=False)
    assert_state_equal(got, want)
    assert got["params"]["xyz"].shape[0] == int((~mask).sum())


def test_clone_and_split_plans_match_reference_semantics():
    P, extent, pd, thr = 800, 2.0, 0.01, 4e-4
    st = make_state(P, 2)
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = st["xyz_gradient_accum"] / st["denom"]
    grads[np.isnan(grads)] = 0
    grads[np.isinf(grads)] = 1.0
    tg = torch.from_numpy(grads)
    # clone
    want = copy.deepcopy(st)
    sel_ref = do.densify_and_clone(want, grads, thr, extent, pd)
    sel = densify.clone_mask(tg, torch.exp(torch.from_numpy(st["params"]["scaling"])), thr, extent, pd)
    assert np.array_equal(sel.numpy(), sel_ref) and 0 < sel_ref.sum() < P
    got = apply_plan_torch(st, densify.plan_clone(sel), reset_stats=True)
    assert_state_equal(got, want)
    # split on the grown set (padded gradients)
    unit = np.random.default_rng(3).normal(0, 1, (4 * P, 3)).astype(np.float32)
    want2 = copy.deepcopy(want)
    sel2_ref = do.densify_and_split(want2, grads, thr, extent, pd, unit)
    P1 = got["params"]["xyz"].shape[0]
    sel2 = densify.split_mask(tg, P1, torch.exp(torch.from_numpy(got["params"]["scaling"])), thr, extent, pd)
    assert np.array_equal(sel2.numpy(), sel2_ref) and sel2_ref.sum() > 0
    plan, children, first = densify.plan_split(sel2, 2)
    got2 = apply_plan_torch(got, plan, reset_stats=True)
    # the children's xyz / scaling are rewritten after the gather (same formula as densify.densify_and_split)
    ch = children.numpy().astype(np.int64)
    stds = np.exp(got["params"]["scaling"][ch])
    R = do.build_rotation(got["params"]["rotation"][ch])
    got2["params"]["xyz"][first:] = np.einsum("nij,nj->ni", R, stds * unit[:len(ch)]) + got["params"]["xyz"][ch]
    got2["params"]["scaling"][first:] = np.log(stds / np.float32(1.6))
    for g in do.GROUPS:
        np.testing.assert_allclose(got2["params"][g], want2["params"][g], rtol=1e-6, atol=1e-7, err_msg=g)
        np.testing.assert_array_equal(got2["exp_avg"][g], want2["exp_avg"][g])
    assert got2["params"]["xyz"].shape[0] == P1 + int(sel2_ref.sum())


def test_reset_opacity_and_stats_on_cpu_tensors():
    class M:
        pass
    m = M()
    P = 64
    st = make_state(P, 5)
    for g in do.GROUPS:
        setattr(m, densify.ATTR[g], torch.nn.Parameter(torch.from_numpy(st["params"][g])))
    type(m).get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    densify.training_setup(m, dict(xyz=1e-3, opacity=0.05))
    assert [g["name"] for g in m.optimizer.param_groups] == list(do.GROUPS)
    m._opacity.grad = torch.ones_like(m._opacity)
    m.optimizer.step()
    densify.reset_opacity(m)
    assert float(torch.sigmoid(m._opacity.detach()).max()) <= 0.01 + 1e-6
    assert float(m.optimizer.state[m._opacity]["exp_avg"].abs().max()) == 0.0
    vs = torch.zeros(P, 3, requires_grad=True)
    filt = torch.rand(P) > 0.5
    with pytest.raises(RuntimeError):  # like the reference, a missing screen-space gradient is an error, not a silent + 0
        densify.add_densification_stats(m, vs, filt)
    vs.grad = torch.randn(P, 3)
    densify.add_densification_stats(m, vs, filt)
    ref = dict(xyz_gradient_accum=np.zeros((P, 1), np.float32), denom=np.zeros((P, 1), np.float32))
    do.add_densification_stats(ref, vs.grad.numpy(), filt.numpy())
    np.testing.assert_allclose(m.xyz_gradient_accum.numpy(), ref["xyz_gradient_accum"], rtol=1e-6)
    np.testing.assert_array_equal(m.denom.numpy(), ref["denom"])


def test_kl_div_known_answers_and_restatement():
    """kl_div is plain tensor math (runs on CPU tensors): closed-form cases + the float64 numpy restatement of :740-762."""
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))  # noqa: E731
    q_id = np.array([[1.0, 0, 0, 0]], np.float32)
    one = np.ones((1, 3), np.float32)
    z = np.zeros((1, 3), np.float32)
    same = densify.kl_div(t(z), t(q_id), t(one), t(z), t(q_id), t(one))
    assert abs(float(same)) < 1e-6
    shifted = densify.kl_div(t(z), t(q_id), t(one), t([[0.3, -0.4, 1.2]]), t(q_id), t(one))     # 1/2 |d|^2
    assert abs(float(shifted) - 0.5 * (0.09 + 0.16 + 1.44)) < 1e-6
    wider = densify.kl_div(t(z), t(q_id), t(one), t(z), t(q_id), t(2 * one))                    # S1 = 4 I
    assert abs(float(wider) - 0.5 * (0.75 + np.log(64.0) - 3.0)) < 1e-6
    rng = np.random.default_rng(5)
    n = 500
    mu0, mu1 = rng.normal(0, 1, (n, 3)).astype(np.float32), rng.normal(0, 1, (n, 3)).astype(np.float32)
    q0, q1 = rng.normal(0, 1, (n, 4)).astype(np.float32), rng.normal(0, 1, (n, 4)).astype(np.float32)
    s0, s1 = np.exp(rng.normal(-1, 0.5, (n, 3))).astype(np.float32), np.exp(rng.normal(-1, 0.5, (n, 3))).astype(np.float32)
    got = densify.kl_div(t(mu0), t(q0), t(s0), t(mu1), t(q1), t(s1)).numpy()
    want = do.kl_div(mu0, q0, s0, mu1, q1, s1)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-4)
    assert (want > -1e-9).all()  # a divergence
    # rotating both Gaussians and the offset by the same rotation changes nothing

    def qmul(a, b):
        w1, x1, y1, z1 = a
        w2, x2, y2, z2 = b.T
        return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                         w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1)
    qa = rng.normal(0, 1, 4)
    qa /= np.linalg.norm(qa)
    Ra = do.build_rotation(qa[None].astype(np.float32))[0].astype(np.float64)
    again = do.kl_div((mu0 @ Ra.T).astype(np.float32), qmul(qa, q0.astype(np.float64)).astype(np.float32), s0,
                      (mu1 @ Ra.T).astype(np.float32), qmul(qa, q1.astype(np.float64)).astype(np.float32), s1)
    np.testing.assert_allclose(again, want, rtol=1e-4, atol=1e-4)
