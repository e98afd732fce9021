"""Per-frame attribute kernel (csrc/attributes.hip) against the reference's chain of torch ops (fp32, run on the same
device): forward values and every input gradient.  Tolerances: 2e-5 relative-to-scale on values, 1e-4 on gradients."""
import types

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _inputs(P, M, seed, dev, equal_scales=False):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev).requires_grad_(True)  # noqa: E731
    A = rng.normal(0, 1, (P, 3, 3)) * 0.3 + np.eye(3)
    scales = np.exp(rng.normal(-4, 0.5, (P, 3)))
    if equal_scales:
        scales[:, 1] = scales[:, 0]
        scales[::2, 2] = scales[::2, 0]
    d = dict(means3D=t(rng.uniform(-1, 1, (P, 3))), transforms=t(A), world_normals=t(rng.normal(0, 1, (P, 3))),
             scales=t(scales), rot_cov=t(rng.normal(0, 1, (P, 4))), rot_axis=t(rng.normal(0, 1, (P, 4))),
             albedo=t(rng.uniform(0, 1, (P, 3))), roughness=t(rng.uniform(0, 1, (P, 3))), occlusion=t(rng.uniform(0, 1, (P, 3))),
             shs=t(np.concatenate([rng.normal(0, 1, (P, 1, 3)), rng.normal(0, 0.3, (P, M - 1, 3))], 1)) if M else None)
    cam = torch.tensor([0.3, -0.2, -3.0], device=dev)
    view = torch.from_numpy(np.linalg.qr(rng.normal(0, 1, (4, 4)))[0].astype(np.float32)).to(dev)
    return d, cam, view


def _run(fn, d, cam, view, mod, deg, w):
    for v in d.values():
        if v is not None:
            v.grad = None
    cov, col, feat = fn(d["means3D"], d["transforms"], d["world_normals"], d["scales"], mod, d["rot_cov"], d["rot_axis"], d["albedo"],
                        d["roughness"], d["occlusion"], d["shs"], deg, cam, view)
    loss = (cov * w[0]).sum() + (feat * w[2]).sum()
    if col is not None:
        loss = loss + (col * w[1]).sum()
    loss.backward()
    return (cov.detach(), None if col is None else col.detach(), feat.detach()), {k: (torch.zeros_like(v) if v.grad is None else v.grad.detach().clone()) for k, v in d.items() if v is not None}


@pytest.mark.parametrize("P,M,deg,equal", [(1000, 16, 3, False), (777, 16, 2, False), (513, 16, 0, False), (300, 4, 1, False),
                                           (256, 0, 0, False), (400, 16, 3, True), (1, 16, 3, False)])
def test_frame_attributes_match_torch_chain(P, M, deg, equal):
    from mygauhuman_amd.attributes import frame_attributes
    from tests.torch_reference import frame_attributes_torch
    dev = torch.device("cuda:0")
    d, cam, view = _inputs(P, M, 5 + P, dev, equal)
    g = torch.Generator(device="cpu").manual_seed(P)
    w = [torch.randn((P, 6), generator=g).to(dev) * 1e4, torch.randn((P, 3), generator=g).to(dev), torch.randn((P, 18), generator=g).to(dev)]
    mod = 1.3
    (cov_r, col_r, feat_r), gr = _run(frame_attributes_torch, d, cam, view, mod, deg, w)
    (cov_k, col_k, feat_k), gk = _run(frame_attributes, d, cam, view, mod, deg, w)
    assert torch.allclose(cov_k, cov_r, rtol=2e-5, atol=1e-9 + 2e-5 * float(cov_r.abs().max()))
    assert torch.allclose(feat_k, feat_r, rtol=0, atol=2e-5)
    if M:
        assert torch.allclose(col_k, col_r, rtol=0, atol=2e-5)
    for k in gr:
        scale = float(gr[k].abs().max()) + 1e-12
        err = float((gk[k] - gr[k]).abs().max()) / scale
        assert err < 1e-4, (k, err, scale)


def test_frame_attributes_null_gradients_and_errors():
    from mygauhuman_amd._lib import GsrError
    from mygauhuman_amd.attributes import frame_attributes
    dev = torch.device("cuda:0")
    d, cam, view = _inputs(64, 16, 1, dev)
    cov, col, feat = frame_attributes(d["means3D"], d["transforms"], d["world_normals"], d["scales"], 1.0, d["rot_cov"], d["rot_axis"],
                                      d["albedo"], d["roughness"], d["occlusion"], d["shs"], 3, cam, view)
    feat[:, 6:9].sum().backward()  # only the albedo columns: everything else must come back exactly zero
    assert torch.equal(d["albedo"].grad, torch.ones_like(d["albedo"]))
    for k in ("means3D", "transforms", "scales", "rot_cov", "rot_axis", "shs", "roughness", "occlusion", "world_normals"):
        assert float(d[k].grad.abs().max()) == 0.0, k
    with pytest.raises(GsrError):
        frame_attributes(d["means3D"], d["transforms"], d["world_normals"], d["scales"], 1.0, d["rot_cov"], d["rot_axis"], d["albedo"],
                         d["roughness"], d["occlusion"], d["shs"][:, :4], 3, cam, view)
    with pytest.raises(RuntimeError):
        cpu = {k: (v.detach().cpu() if v is not None else None) for k, v in d.items()}
        frame_attributes(cpu["means3D"], cpu["transforms"], cpu["world_normals"], cpu["scales"], 1.0, cpu["rot_cov"], cpu["rot_axis"],
                         cpu["albedo"], cpu["roughness"], cpu["occlusion"], cpu["shs"], 3, cam.cpu(), view.cpu())


@pytest.mark.parametrize("P", [1, 777, 20000])
def test_frame_activations_match_the_property_getters(P):
    """csrc/activations.hip against the torch ops of the reference's getters (scene/gaussian_model.py:157-199) and
    render()'s opacity.repeat(1, 3): values and the gradients autograd derives, N(0,1) upstream gradients."""
    import torch.nn.functional as F
    from mygauhuman_amd.activations import frame_activations
    rng = np.random.default_rng(P)
    mk = lambda *s: torch.from_numpy(rng.normal(0, 1.5, s).astype(np.float32)).cuda()  # noqa: E731
    raw = [mk(P, 1), mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 3)]
    ups = [mk(P, 1), mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 3), mk(P, 3)]

    def getters(o, a, s, r, n):
        op = torch.sigmoid(o)
        return op, torch.sigmoid(a), torch.exp(s), F.normalize(r), n / n.norm(dim=1, keepdim=True), op.repeat(1, 3)

    res = {}
    for name, fn in (("hip", frame_activations), ("torch", getters)):
        leaves = [t.clone().requires_grad_(True) for t in raw]
        outs = fn(*leaves)
        sum((o * u).sum() for o, u in zip(outs, ups)).backward()
        res[name] = ([o.detach().cpu().numpy() for o in outs], [t.grad.cpu().numpy() for t in leaves])
    for a, b in zip(res["hip"][0], res["torch"][0]):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7)
    for k, (a, b) in enumerate(zip(res["hip"][1], res["torch"][1])):
        util.assert_close(f"raw gradient {k}", a, b, tol=2e-5, max_bad_frac=0)
    # outputs that do not reach the loss hand None to the backward
    leaves = [t.clone().requires_grad_(True) for t in raw]
    outs = frame_activations(*leaves)
    (outs[0] * ups[0]).sum().backward()
    want = [t.clone().requires_grad_(True) for t in raw]
    (torch.sigmoid(want[0]) * ups[0]).sum().backward()
    np.testing.assert_allclose(leaves[0].grad.cpu().numpy(), want[0].grad.cpu().numpy(), rtol=2e-5, atol=1e-7)
    for t in leaves[1:]:
        assert float(t.grad.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        frame_activations(*[t.cpu() for t in raw])


@pytest.mark.parametrize("P,deg", [(1000, 3), (777, 2), (1, 3), (300, 0)])
def test_frame_attributes_read_the_two_sh_tensors_in_place(P, deg):
    """shs = (features_dc [P,1,3], features_rest [P,15,3]): same bits as the concatenated [P,16,3] array, gradients land in
    the two tensors directly."""
    from mygauhuman_amd.attributes import frame_attributes
    dev = torch.device("cuda:0")
    d, cam, view = _inputs(P, 16, 11 + P, dev)
    g = torch.Generator(device="cpu").manual_seed(P)
    w = [torch.randn((P, 6), generator=g).to(dev), torch.randn((P, 3), generator=g).to(dev), torch.randn((P, 18), generator=g).to(dev)]
    (cov_a, col_a, feat_a), ga = _run(frame_attributes, d, cam, view, 1.0, deg, w)
    dc = d["shs"].detach()[:, :1].clone().requires_grad_(True)
    rest = d["shs"].detach()[:, 1:].clone().requires_grad_(True)
    d2 = dict(d, shs=(dc, rest))
    for v in d.values():
        v.grad = None
    cov, col, feat = frame_attributes(d2["means3D"], d2["transforms"], d2["world_normals"], d2["scales"], 1.0, d2["rot_cov"],
                                      d2["rot_axis"], d2["albedo"], d2["roughness"], d2["occlusion"], d2["shs"], deg, cam, view)
    ((cov * w[0]).sum() + (col * w[1]).sum() + (feat * w[2]).sum()).backward()
    assert torch.equal(cov, cov_a) and torch.equal(col, col_a) and torch.equal(feat, feat_a)
    assert torch.equal(dc.grad, ga["shs"][:, :1]) and torch.equal(rest.grad, ga["shs"][:, 1:])
    for k in ("means3D", "transforms", "scales", "rot_cov", "albedo"):
        assert torch.equal(d[k].grad, ga[k]), k
