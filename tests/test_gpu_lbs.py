"""GPU parity tests of the SMPL LBS kernels: forward against the CPU oracle (oracle/lbs_oracle.c, which is pinned to
the imported reference smplx.lbs through tests/golden), backward against float64 torch autograd of a restatement of
the per-point math (the reference relies on autograd for this gradient too).  Tolerance 1e-4 (fp32)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int32)
BIG_POSE = np.zeros(72, np.float32)
BIG_POSE[5], BIG_POSE[8], BIG_POSE[23], BIG_POSE[26] = np.deg2rad(45), -np.deg2rad(45), -np.deg2rad(30), np.deg2rad(30)


def make_smpl(V, seed):
    rng = np.random.default_rng(seed)
    vt = rng.uniform(-1, 1, (V, 3)).astype(np.float32) * np.array([0.9, 0.9, 0.15], np.float32)
    sd = rng.normal(0, 0.01, (V, 3, 10)).astype(np.float32)
    pd = rng.normal(0, 0.001, (V, 3, 207)).astype(np.float32)  # gaussian_model.py layout [V,3,207]
    J = rng.uniform(0, 1, (24, V)).astype(np.float32)
    J /= J.sum(1, keepdims=True)
    w = rng.uniform(0, 1, (V, 24)).astype(np.float32) ** 4
    w /= w.sum(1, keepdims=True)
    return dict(v_template=vt, shapedirs=sd, posedirs=pd, J_regressor=J, weights=w.astype(np.float32), parents=PARENTS)


def make_case(oracle, V, P, seed, with_offsets):
    rng = np.random.default_rng(seed)
    m = make_smpl(V, seed)
    betas = rng.normal(0, 1, 10).astype(np.float32)
    pose = rng.normal(0, 0.2, 72).astype(np.float32)
    zeros = np.zeros(10, np.float32)
    rot_big, rot_pose = oracle.rodrigues(BIG_POSE), oracle.rodrigues(pose)
    A_big, _ = oracle.joint_transforms(m, zeros, rot_big)
    A_pose, _ = oracle.joint_transforms(m, betas, rot_pose)
    off_big, off_pose = oracle.pose_offsets(m["posedirs"], rot_big), oracle.pose_offsets(m["posedirs"], rot_pose)
    off_shape = oracle.shape_offsets(m["shapedirs"], betas)
    # big-pose vertices = smplx lbs of the template (any vertex cloud works for the nearest-vertex search)
    big_verts = (m["v_template"] + 0.01 * rng.normal(0, 1, (V, 3))).astype(np.float32)
    query = (big_verts[rng.integers(0, V, P)] + rng.normal(0, 0.02, (P, 3))).astype(np.float32)
    normals = rng.normal(0, 1, (P, 3)).astype(np.float32)
    ang = 0.4
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    Th = np.array([0.1, -0.3, 2.5], np.float32)
    loff = rng.normal(0, 0.5, (P, 24)).astype(np.float32) if with_offsets else None
    return dict(m=m, betas=betas, pose=pose, A_big=A_big, A_pose=A_pose, off_big=off_big, off_pose=off_pose,
                off_shape=off_shape, big_verts=big_verts, query=query, normals=normals, R=R, Th=Th, loff=loff)


@pytest.mark.parametrize("V,P", [(700, 1), (700, 1000), (6890, 20000)])
@pytest.mark.parametrize("with_offsets", [False, True])
def test_lbs_forward_matches_oracle(oracle, V, P, with_offsets):
    from mygauhuman_amd import lbs
    c = make_case(oracle, V, P, V + P, with_offsets)
    ids = oracle.nearest_vertex(c["query"], c["big_verts"])
    want = oracle.lbs_deform(c["query"], c["normals"], ids, c["m"]["weights"], c["A_big"], c["A_pose"], c["off_big"],
                             c["off_shape"], c["off_pose"], c["R"], c["Th"], lbs_off=c["loff"])
    d = util.to_dev
    got = lbs.lbs_deform(d(c["query"]), d(c["normals"]), None if c["loff"] is None else d(c["loff"]), d(c["A_big"]),
                         d(c["A_pose"]), d(c["off_big"]), d(c["off_shape"]), d(c["off_pose"]), d(c["R"]), d(c["Th"]),
                         d(c["big_verts"]), d(c["m"]["weights"]))
    np.testing.assert_array_equal(got["vert_ids"].cpu().numpy(), ids)  # index work: bit-exact
    for k_got, k_want in (("world_pts", "world_src"), ("smpl_pts", "smpl_src"), ("bweights", "bweights"),
                          ("transforms", "transforms"), ("translation", "translation"), ("world_normals", "world_normals")):
        np.testing.assert_allclose(got[k_got].cpu().numpy(), want[k_want], rtol=1e-4, atol=1e-4, err_msg=k_got)


def test_coarse_deform_c2source_matches_oracle_pipeline(oracle):
    """The drop-in for GaussianModel.coarse_deform_c2source (torch glue + HIP kernel) against the oracle pieces."""
    from mygauhuman_amd import lbs
    V, P = 900, 3000
    c = make_case(oracle, V, P, 5, True)
    d = util.to_dev
    m = c["m"]
    smpl = dict(v_template=d(m["v_template"]), shapedirs=d(m["shapedirs"]), posedirs=d(m["posedirs"]),
                J_regressor=d(m["J_regressor"]), weights=d(m["weights"]),
                kintree_table=torch.from_numpy(np.stack([PARENTS, np.arange(24)]).astype(np.int64)).cuda())
    params = dict(poses=d(c["pose"][None]), shapes=d(c["betas"][None]), R=d(c["R"]), Th=d(c["Th"][None]))
    t_params = dict(poses=d(BIG_POSE[None]), shapes=d(np.zeros((1, 10), np.float32)), R=d(np.eye(3, dtype=np.float32)),
                    Th=d(np.zeros((1, 3), np.float32)))
    smpl_src, world, bw, tf, transl, wn = lbs.coarse_deform_c2source(
        smpl, d(c["query"][None]), params, t_params, d(c["big_verts"][None]), lbs_weights=d(c["loff"][None]),
        return_transl=True, normals=d(c["normals"][None]))
    ids = oracle.nearest_vertex(c["query"], c["big_verts"])
    want = oracle.lbs_deform(c["query"], c["normals"], ids, m["weights"], c["A_big"], c["A_pose"], c["off_big"],
                             c["off_shape"], c["off_pose"], c["R"], c["Th"], lbs_off=c["loff"])
    assert world.shape == (1, P, 3) and tf.shape == (1, P, 3, 3) and bw.shape == (1, P, 24)
    np.testing.assert_allclose(world[0].cpu().numpy(), want["world_src"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tf[0].cpu().numpy(), want["transforms"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(transl[0].cpu().numpy(), want["translation"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(wn[0].cpu().numpy(), want["world_normals"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(smpl_src[0].cpu().numpy(), want["smpl_src"], rtol=1e-4, atol=1e-4)
    # the big-pose side and the shape offsets are cached on their (frame-constant) input tensors: a second call reuses them and
    # gives the same bits, an in-place change of an input (version bump) recomputes
    n0 = len(lbs._CONSTANTS.entries)
    again = lbs.coarse_deform_c2source(smpl, d(c["query"][None]), params, t_params, d(c["big_verts"][None]),
                                       lbs_weights=d(c["loff"][None]), return_transl=True, normals=d(c["normals"][None]))
    assert len(lbs._CONSTANTS.entries) == n0 and torch.equal(again[1], world) and torch.equal(again[3], tf)
    t_params["poses"].mul_(0.5)
    params["shapes"].add_(0.25)
    moved = lbs.coarse_deform_c2source(smpl, d(c["query"][None]), params, t_params, d(c["big_verts"][None]),
                                       lbs_weights=d(c["loff"][None]), return_transl=True, normals=d(c["normals"][None]))
    fresh_t = {k: v.clone() for k, v in t_params.items()}
    fresh_p = {k: v.clone() for k, v in params.items()}
    fresh = lbs.coarse_deform_c2source(smpl, d(c["query"][None]), fresh_p, fresh_t, d(c["big_verts"][None]),
                                       lbs_weights=d(c["loff"][None]), return_transl=True, normals=d(c["normals"][None]))
    assert not torch.equal(moved[1], world) and torch.equal(moved[1], fresh[1]) and torch.equal(moved[3], fresh[3])
    # inputs inside the autograd graph are never cached
    t_grad = dict(t_params, poses=t_params["poses"].clone().requires_grad_(True))
    n1 = len(lbs._CONSTANTS.entries)
    lbs.coarse_deform_c2source(smpl, d(c["query"][None]), params, t_grad, d(c["big_verts"][None]))
    assert len(lbs._CONSTANTS.entries) == n1


def _torch_deform(query, normals, loff, A_big, A_pose, off_big, off_shape, off_pose, R, Th, ids, weights):
    """float64 restatement of the per-point math (gaussian_model.py:776-872) for autograd."""
    bw = weights[ids]
    if loff is not None:
        bw = torch.softmax(torch.log(bw + 1e-9) + loff, dim=-1)
    Ab = (bw @ A_big.reshape(24, 16)).reshape(-1, 4, 4)
    Ap = (bw @ A_pose.reshape(24, 16)).reshape(-1, 4, 4)
    Ri = torch.inverse(Ab[:, :3, :3])
    q = (Ri @ (query - Ab[:, :3, 3])[..., None])[..., 0]
    n = (Ri @ normals[..., None])[..., 0]
    q = q - off_big[ids] + off_shape[ids] + off_pose[ids]
    src = (Ap[:, :3, :3] @ q[..., None])[..., 0] + Ap[:, :3, 3]
    sn = (Ap[:, :3, :3] @ n[..., None])[..., 0]
    Rinv = torch.inverse(R)
    return src @ Rinv + Th, R @ (Ap[:, :3, :3] @ Ri), sn @ Rinv


@pytest.mark.parametrize("with_offsets", [False, True])
def test_lbs_backward_matches_autograd(oracle, with_offsets):
    from mygauhuman_amd import lbs
    V, P = 800, 2500
    c = make_case(oracle, V, P, 11, with_offsets)
    rng = np.random.default_rng(3)
    gw, gt, gn = (rng.normal(0, 1, (P, 3)), rng.normal(0, 1, (P, 3, 3)), rng.normal(0, 1, (P, 3)))
    ids = oracle.nearest_vertex(c["query"], c["big_verts"])
    # ---- float64 autograd reference on the CPU
    t64 = lambda a, g=False: torch.tensor(np.asarray(a, np.float64), requires_grad=g)  # noqa: E731
    rq, rn, rA, ro = t64(c["query"], True), t64(c["normals"], True), t64(c["A_pose"], True), t64(c["off_pose"], True)
    rl = t64(c["loff"], True) if with_offsets else None
    w, tf, wn = _torch_deform(rq, rn, rl, t64(c["A_big"]), rA, t64(c["off_big"]), t64(c["off_shape"]), ro, t64(c["R"]),
                              t64(c["Th"]), torch.from_numpy(ids.astype(np.int64)), t64(c["m"]["weights"]))
    ((w * t64(gw)).sum() + (tf * t64(gt)).sum() + (wn * t64(gn)).sum()).backward()
    # ---- HIP path
    d = util.to_dev
    hq, hn = d(c["query"]).requires_grad_(True), d(c["normals"]).requires_grad_(True)
    hA, ho = d(c["A_pose"]).requires_grad_(True), d(c["off_pose"]).requires_grad_(True)
    hl = d(c["loff"]).requires_grad_(True) if with_offsets else None
    o = lbs.lbs_deform(hq, hn, hl, d(c["A_big"]), hA, d(c["off_big"]), d(c["off_shape"]), ho, d(c["R"]), d(c["Th"]),
                       d(c["big_verts"]), d(c["m"]["weights"]))
    loss = (o["world_pts"] * d(gw.astype(np.float32))).sum() + (o["transforms"] * d(gt.astype(np.float32))).sum() + \
           (o["world_normals"] * d(gn.astype(np.float32))).sum()
    loss.backward()
    util.assert_close("d_query", hq.grad.cpu().numpy(), rq.grad.numpy(), tol=1e-4)
    util.assert_close("d_normals", hn.grad.cpu().numpy(), rn.grad.numpy(), tol=1e-4)
    util.assert_close("d_A_pose", hA.grad.cpu().numpy()[:, :3, :], rA.grad.numpy()[:, :3, :], tol=1e-4)
    util.assert_close("d_off_pose", ho.grad.cpu().numpy(), ro.grad.numpy(), tol=1e-4)
    if with_offsets:
        util.assert_close("d_lbs_offsets", hl.grad.cpu().numpy(), rl.grad.numpy(), tol=1e-4)


def _vertex_ids(lbs, search, query, verts, c):
    d = util.to_dev
    old = lbs.NEAREST_VERTEX_SEARCH
    lbs.NEAREST_VERTEX_SEARCH = search
    try:
        V = verts.shape[0]
        w = np.full((V, 24), 1.0 / 24, np.float32)
        z = np.zeros((V, 3), np.float32)
        out = lbs.lbs_deform(d(query), None, None, d(c["A_big"]), d(c["A_pose"]), d(z), d(z), d(z), d(c["R"]), d(c["Th"]), d(verts), d(w))
        return out["vert_ids"].cpu().numpy()
    finally:
        lbs.NEAREST_VERTEX_SEARCH = old


@pytest.mark.parametrize("case", ["near", "far_outside", "duplicates", "coincident", "planar", "single", "clusters"])
def test_nearest_vertex_grid_equals_brute_and_oracle(oracle, case):
    """The grid search must return the brute-force answer bit for bit (lowest index on ties), also for queries far outside
    the vertex bounding box and for degenerate vertex clouds."""
    from mygauhuman_amd import lbs
    c = make_case(oracle, 700, 10, 3, False)
    rng = np.random.default_rng(11)
    V, P = 3000, 5000
    verts = (rng.uniform(-1, 1, (V, 3)) * np.array([0.9, 0.9, 0.15])).astype(np.float32)
    query = (verts[rng.integers(0, V, P)] + rng.normal(0, 0.02, (P, 3))).astype(np.float32)
    if case == "far_outside":
        query = (rng.normal(0, 1, (P, 3)) * 5).astype(np.float32)
    elif case == "duplicates":   # every vertex appears four times; queries sit exactly on vertices
        verts = np.concatenate([verts[:750]] * 4).astype(np.float32)
        query = verts[rng.integers(0, V, P)].copy()
    elif case == "coincident":   # zero-extent cloud
        verts = np.tile(np.array([[0.3, -0.2, 0.1]], np.float32), (64, 1))
    elif case == "planar":
        verts[:, 2] = 0.25
    elif case == "single":
        verts = verts[:1]
    elif case == "clusters":     # two far-apart blobs: most cells empty, long ring walks
        verts = np.concatenate([verts[:1500] * 0.01, verts[1500:] * 0.01 + 4.0]).astype(np.float32)
        query = (rng.uniform(-1, 5, (P, 3))).astype(np.float32)
    want = oracle.nearest_vertex(query, verts)
    got_grid = _vertex_ids(lbs, "grid", query, verts, c)
    got_brute = _vertex_ids(lbs, "brute", query, verts, c)
    np.testing.assert_array_equal(got_brute, want)
    np.testing.assert_array_equal(got_grid, want)


@pytest.mark.parametrize("with_correct", [False, True])
def test_smpl_pose_kernel_matches_torch_chain(oracle, with_correct):
    """csrc/pose.hip against the torch formulation of the chain (tests/torch_reference.py: rodrigues + rigid_chain) evaluated in
    float64 with autograd: A, rot_mats and the gradients w.r.t. poses, correct_Rs and joints.  Tolerance 1e-4 (fp32)."""
    from mygauhuman_amd import lbs
    rng = np.random.default_rng(21)
    dev = torch.device("cuda:0")
    poses = torch.tensor(rng.normal(0, 0.4, (1, 72)), dtype=torch.float32, device=dev, requires_grad=True)
    joints = torch.tensor(rng.normal(0, 0.3, (24, 3)), dtype=torch.float32, device=dev, requires_grad=True)
    cr = None
    if with_correct:
        cr_np = np.stack([np.eye(3) + rng.normal(0, 0.05, (3, 3)) for _ in range(23)])
        cr = torch.tensor(cr_np, dtype=torch.float32, device=dev, requires_grad=True)
    wA = torch.tensor(rng.normal(0, 1, (24, 4, 4)), dtype=torch.float32, device=dev)
    wR = torch.tensor(rng.normal(0, 1, (24, 3, 3)), dtype=torch.float32, device=dev)
    parents = tuple(int(v) for v in PARENTS)

    rot, A = lbs._SmplPose.apply(poses, cr, joints, parents)
    ((A * wA).sum() + (rot * wR).sum()).backward()
    got = dict(A=A.detach(), rot=rot.detach(), d_poses=poses.grad.clone(), d_joints=joints.grad.clone(),
               d_cr=None if cr is None else cr.grad.clone())

    p64 = poses.detach().double().requires_grad_(True)
    j64 = joints.detach().double().requires_grad_(True)
    c64 = None if cr is None else cr.detach().double().requires_grad_(True)
    rot64 = lbs.batch_rodrigues(p64.view(-1, 3)).view(1, 24, 3, 3)
    if c64 is not None:
        rot64 = torch.cat([rot64[:, 0:1], torch.matmul(rot64[0, 1:], c64)[None]], dim=1)
    from tests.torch_reference import rigid_chain
    A64 = rigid_chain(rot64, j64[None], list(parents))
    ((A64[0] * wA.double()).sum() + (rot64[0] * wR.double()).sum()).backward()

    def close(a, b, name):
        scale = float(b.abs().max()) + 1e-12
        err = float((a.double() - b).abs().max()) / scale
        assert err < 1e-4, (name, err)
    close(got["A"], A64[0].detach(), "A")
    close(got["rot"], rot64[0].detach(), "rot_mats")
    close(got["d_poses"], p64.grad, "d_poses")
    close(got["d_joints"], j64.grad, "d_joints")
    if cr is not None:
        close(got["d_cr"], c64.grad, "d_correct_Rs")
    assert torch.equal(A[:, 3], torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev).expand(24, 4))
    with pytest.raises(RuntimeError):
        lbs._SmplPose.apply(poses, cr, joints, (0,) + tuple(range(1, 24)))  # parents[1] = 1 does not precede joint 1


@pytest.mark.parametrize("R,K", [(20670, 207), (1, 1), (1000, 256), (37, 64), (16, 65)])
def test_row_gemv_forward_backward(R, K):
    """csrc/gemv.hip (pose blend-shape product) against torch.matmul in float64."""
    from mygauhuman_amd import lbs
    g = torch.Generator().manual_seed(R + K)
    mat = torch.randn((R, K), generator=g).cuda()
    vec = torch.randn((K,), generator=g).cuda().requires_grad_(True)
    w = torch.randn((R,), generator=g).cuda()
    out = lbs._RowGemv.apply(mat, vec)
    (out * w).sum().backward()
    ref = mat.double() @ vec.detach().double()
    dref = mat.double().t() @ w.double()
    assert float((out.detach().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max() + 1e-30)
    assert float((vec.grad.double() - dref).abs().max()) <= 2e-5 * float(dref.abs().max() + 1e-30)


def test_vertex_grid_cache_reuse_and_invalidation(oracle):
    """The cached grid must be reused for the same vertex tensor and rebuilt after an in-place change of the vertices."""
    from mygauhuman_amd import lbs
    c = make_case(oracle, 700, 10, 5, False)
    rng = np.random.default_rng(0)
    verts = torch.from_numpy((rng.uniform(-1, 1, (2000, 3))).astype(np.float32)).cuda()
    query = (verts[rng.integers(0, 2000, 3000)] + 0.01 * torch.randn(3000, 3, device="cuda")).contiguous()
    d = util.to_dev
    w = d(np.full((2000, 24), 1 / 24, np.float32))
    z = d(np.zeros((2000, 3), np.float32))

    def ids():
        return lbs.lbs_deform(query, None, None, d(c["A_big"]), d(c["A_pose"]), z, z, z, d(c["R"]), d(c["Th"]), verts, w, lean=True)["vert_ids"].cpu().numpy()
    n0 = len(lbs._GRIDS.entries)
    a = ids()
    n1 = len(lbs._GRIDS.entries)
    b = ids()
    assert n1 == n0 + 1 and len(lbs._GRIDS.entries) == n1          # second call reused the entry
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, oracle.nearest_vertex(query.cpu().numpy(), verts.cpu().numpy()))
    verts.mul_(-1.0)                                               # in-place change: version bump -> new grid
    e = ids()
    assert len(lbs._GRIDS.entries) == n1 + 1
    np.testing.assert_array_equal(e, oracle.nearest_vertex(query.cpu().numpy(), verts.cpu().numpy()))
    out = lbs.lbs_deform(query, None, None, d(c["A_big"]), d(c["A_pose"]), z, z, z, d(c["R"]), d(c["Th"]), verts, w, lean=True)
    assert out["bweights"].numel() == 0 and out["smpl_pts"].numel() == 0 and out["world_pts"].shape == (3000, 3)


@pytest.mark.parametrize("case", ["walk", "duplicates", "coincident", "single", "slot_reuse"])
def test_temporal_nearest_vertex_cache_is_exact(oracle, case):
    """The temporal cache (csrc/lbs.hip "exact temporal cache"; the reference searches every frame, scene/gaussian_model.py:775):
    ids bit-identical to the full search -- and to the oracle's brute force -- while points WALK ACROSS Voronoi boundaries in
    steps from far below to far above the typical gap between the two nearest vertices; for vertex clouds with exact ties
    (duplicates, a zero-extent cloud: every entry has rho = 0 and is searched every frame), for a single vertex (rho unbounded),
    and when the point slots are refilled with different points (what densify / prune does to a slot)."""
    from mygauhuman_amd import lbs
    c = make_case(oracle, 700, 10, 3, False)
    rng = np.random.default_rng(5)
    V, P = 3000, 20000
    verts_np = (rng.uniform(-1, 1, (V, 3)) * np.array([0.9, 0.9, 0.15])).astype(np.float32)
    if case == "duplicates":
        verts_np = np.concatenate([verts_np[:750]] * 4).astype(np.float32)
    elif case == "coincident":
        verts_np = np.tile(np.array([[0.3, -0.2, 0.1]], np.float32), (64, 1))
    elif case == "single":
        verts_np = verts_np[:1]
    Vn = verts_np.shape[0]
    verts = torch.from_numpy(verts_np).cuda()
    d = util.to_dev
    w, z = d(np.full((Vn, 24), 1 / 24, np.float32)), d(np.zeros((Vn, 3), np.float32))
    A_big, A_pose, R, Th = d(c["A_big"]), d(c["A_pose"]), d(c["R"]), d(c["Th"])
    query = torch.from_numpy((verts_np[rng.integers(0, Vn, P)] + rng.normal(0, 0.02, (P, 3))).astype(np.float32)).cuda()
    # a fixed direction per point: the walk crosses cell after cell of the Voronoi diagram
    direction = torch.from_numpy(rng.normal(0, 1, (P, 3)).astype(np.float32)).cuda()
    direction /= direction.norm(dim=1, keepdim=True)

    def ids(cached):
        old = lbs.NN_TEMPORAL_CACHE
        lbs.NN_TEMPORAL_CACHE = cached
        try:
            return lbs.lbs_deform(query, None, None, A_big, A_pose, z, z, z, R, Th, verts, w, lean=True)["vert_ids"].cpu().numpy()
        finally:
            lbs.NN_TEMPORAL_CACHE = old
    assert lbs.NN_TEMPORAL_CACHE is True   # the default
    first = ids(True)                      # makes the entries (full search)
    np.testing.assert_array_equal(first, oracle.nearest_vertex(query.cpu().numpy(), verts_np))
    changed_total, steps = 0, [0.0, 1e-6, 1e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 1e-4, 1e-4, 0.0]
    prev = first
    for k, step in enumerate(steps):
        query.add_(direction * step)
        if case == "slot_reuse" and k == 5:   # the slots now hold other points altogether
            query.copy_(query[torch.randperm(P, device="cuda")])
        got = ids(True)
        want = ids(False)
        np.testing.assert_array_equal(got, want, err_msg=f"step {k} ({step})")
        if k in (0, 6, len(steps) - 1):
            np.testing.assert_array_equal(got, oracle.nearest_vertex(query.cpu().numpy(), verts_np))
        changed_total += int((got != prev).sum())
        prev = got
        misses, _ = lbs._GRIDS.nn_cache_stats(verts, P)
        if case == "walk":
            if step == 0.0:
                assert misses < 0.12 * P, (k, misses)      # only the entries whose two nearest vertices tie within the guard
            if step >= 3e-2:
                assert misses > 0.5 * P, (k, misses)       # a step of the order of the vertex spacing: most points re-search
        if case in ("duplicates", "coincident"):
            # exact ties everywhere: rho = 0, no neighbourhood is ever trusted -- only a point that has not moved at all keeps its id
            assert misses == (0 if step == 0.0 else P), (k, step, misses)
        if case == "single":
            assert misses == 0
    if case in ("walk", "slot_reuse"):
        assert changed_total > P // 2, "the walk must have crossed Voronoi boundaries for the test to mean anything"
