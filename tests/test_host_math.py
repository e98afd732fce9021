"""CPU tests of the host-side torch helpers against golden vectors / numpy restatements."""
import os

import numpy as np
import torch

from mygauhuman_amd import covariance, sh_utils


def test_eval_sh_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "sh_eval.npz"))
    sh, dirs = torch.from_numpy(g["sh"]), torch.from_numpy(g["dirs"])
    for deg in range(4):
        np.testing.assert_allclose(sh_utils.eval_sh(deg, sh, dirs).numpy(), g[f"rgb_deg{deg}"], rtol=1e-6, atol=1e-7)


def test_covariance_matches_numpy():
    rng = np.random.default_rng(0)
    P = 50
    s = np.exp(rng.normal(-3, 0.5, (P, 3))).astype(np.float32)
    q = rng.normal(0, 1, (P, 4)).astype(np.float32)
    T = rng.normal(0, 1, (P, 3, 3)).astype(np.float32)
    got = covariance.build_covariance_from_scaling_rotation(torch.from_numpy(s), 1.5, torch.from_numpy(q), torch.from_numpy(T)).numpy()
    for i in range(P):
        w, x, y, z = q[i] / np.linalg.norm(q[i])
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], np.float64)
        L = R @ np.diag(1.5 * s[i].astype(np.float64))
        C = T[i].astype(np.float64) @ (L @ L.T) @ T[i].astype(np.float64).T
        np.testing.assert_allclose(got[i], C[np.triu_indices(3)], rtol=2e-4, atol=1e-7)


def test_minimum_axis_reproduces_reference_indexing():
    """utils/general_utils.py:144-149 takes row 0 of the column-sorted rotation matrix."""
    s = torch.tensor([[0.3, 0.1, 0.2]])
    q = torch.tensor([[0.9, 0.1, 0.3, -0.2]])
    R = covariance.build_rotation(q)[0]
    idx = torch.argsort(s[0])
    want = R[0, idx]
    np.testing.assert_allclose(covariance.get_minimum_axis(s, q)[0].numpy(), want.numpy(), rtol=1e-6)
    n, nf = covariance.flip_align_view(torch.tensor([[0.0, 0.0, 1.0]]), torch.tensor([[0.0, 0.0, 1.0]]))
    assert n[0, 2] == -1.0 and not bool(nf[0, 0])
    v = torch.tensor([[1.0, 2.0, 3.0]])
    m = torch.arange(16, dtype=torch.float32).view(4, 4)
    np.testing.assert_allclose(covariance.transformVector3x3(v, m).numpy(), (v @ m[:3, :3]).numpy())


def test_smplx_lbs_and_rodrigues_match_reference_golden(golden_dir):
    """Config 1 plumbing: the torch mirror of smplx lbs() against vectors from the imported reference."""
    from mygauhuman_amd import lbs
    g = np.load(os.path.join(golden_dir, "lbs_smpl.npz"))
    T = torch.from_numpy
    verts, J, A, Tm = lbs.smplx_lbs(T(g["betas"]), T(g["pose"]), T(g["smpl_v_template"]), T(g["smpl_shapedirs"]),
                                    T(g["smpl_posedirs"]), T(g["smpl_J_regressor"]), T(g["smpl_parents"]), T(g["smpl_weights"]))
    np.testing.assert_allclose(verts[0].numpy(), g["verts"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(J[0].numpy(), g["J_transformed"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(A[0].numpy(), g["A"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(Tm[0].numpy(), g["T"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(lbs.batch_rodrigues(T(g["rodrigues_in"])).numpy(), g["rodrigues_out"], rtol=1e-5, atol=1e-6)
    # the torch checker of the pose kernel builds the same A (same chain, gaussian_model.py:914-980)
    from tests.torch_reference import pose_transforms_torch
    smpl = dict(v_template=T(g["smpl_v_template"]), shapedirs=T(g["smpl_shapedirs"]), J_regressor=T(g["smpl_J_regressor"]),
                kintree_table=torch.stack([T(g["smpl_parents"]), torch.arange(24)]))
    A2, _, _, _ = pose_transforms_torch(smpl, dict(shapes=T(g["betas"]), poses=T(g["pose"]), R=None, Th=None))
    np.testing.assert_allclose(A2[0].numpy(), g["A"], rtol=1e-4, atol=2e-6)
