"""SMPL linear-blend skinning of the canonical Gaussians -- host side of the LBS row of the hot path.

Mirrors GaussianModel.coarse_deform_c2source (scene/gaussian_model.py:768-872: same arguments / return tuple; `smpl` replaces
self.SMPL_NEUTRAL).  Its helpers (:894-1013: rodrigues, rigid transformation chain, transform parameters) run as ONE
single-wave HIP kernel each way (csrc/pose.hip, smpl_pose_transforms below); the plain-torch formulation of that chain lives
with the tests (tests/torch_reference.py) as its checker.  The per-vertex blend-shape offsets are HIP GEMVs.  Everything per POINT
(nearest SMPL vertex, weight softmax, 24-way blends, 3x3 inverse, offsets, posing, world transform) is ONE HIP
kernel forward (gsr_lbs_forward) and ONE backward (gsr_lbs_backward) behind a torch.autograd.Function, instead of
~40 torch kernels + KNN_CUDA + an autograd graph over [P, 24, 16] intermediates.
"""
import ctypes as C

import torch

from ._lib import check, lib, ptr


def batch_rodrigues(rot_vecs):
    """Axis-angle vectors [N,3] -> rotation matrices [N,3,3] by Rodrigues' formula in its outer-product form
    R = cos(t) I + sin(t) [n]x + (1 - cos(t)) n n^T, with t = |v + 1e-8| and n = v / t (the epsilon convention of
    smplx/lbs.py:batch_rodrigues, which the golden vectors pin).  Host-side helper of smplx_lbs (BASELINE config 1); the
    per-frame pose path of render() runs csrc/pose.hip instead."""
    theta = (rot_vecs + 1e-8).norm(dim=1)
    n = rot_vecs / theta[:, None]
    c, s_ = torch.cos(theta)[:, None, None], torch.sin(theta)[:, None, None]
    cross = torch.zeros((rot_vecs.shape[0], 3, 3), dtype=rot_vecs.dtype, device=rot_vecs.device)
    cross[:, 0, 1], cross[:, 0, 2], cross[:, 1, 2] = -n[:, 2], n[:, 1], -n[:, 0]
    cross = cross - cross.transpose(1, 2)
    eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)
    return c * eye + s_ * cross + (1 - c) * (n[:, :, None] * n[:, None, :])


class _RowGemv(torch.autograd.Function):
    """out[R] = mat[R,K] @ vec[K] with the HIP GEMV (csrc/gemv.hip); gradient w.r.t. vec only (mat = constant blend shapes)."""

    @staticmethod
    def forward(ctx, mat, vec):
        dev = mat.device
        m, v = mat.detach().contiguous().float(), vec.detach().contiguous().float()
        out = torch.empty((m.shape[0],), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib.gsr_gemv_rows(m.shape[0], m.shape[1], ptr(m), ptr(v), ptr(out), torch.cuda.current_stream(dev).cuda_stream),
                  "gsr_gemv_rows")
        ctx.save_for_backward(m)
        ctx.vshape = vec.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        dv = torch.empty((m.shape[1],), dtype=torch.float32, device=m.device)
        g = g.contiguous().float()
        with torch.cuda.device(m.device):
            check(lib.gsr_gemv_rows_t(m.shape[0], m.shape[1], ptr(m), ptr(g), ptr(dv), torch.cuda.current_stream(m.device).cuda_stream),
                  "gsr_gemv_rows_t")
        return None, dv.view(ctx.vshape)


_IDENT3 = {}


def _ident3(dtype, device):
    """A cached 3 x 3 identity per (dtype, device): torch.eye is a fill + a strided write every frame."""
    key = (dtype, str(device))
    t = _IDENT3.get(key)
    if t is None or (t.is_cuda and torch.cuda.is_current_stream_capturing()):
        t = torch.eye(3, dtype=dtype, device=device)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            _IDENT3[key] = t
    return t


def pose_offsets(smpl, rot_mats):
    """(R[1:] - I).flatten() [1,207] @ posedirs^T -> per-vertex offsets [V,3] (gaussian_model.py:805-811,827-839): one
    HBM-streaming HIP GEMV (rocBLAS needs 60 us for this 17 MB product).  No CPU / torch fallback."""
    posedirs = smpl["posedirs"]
    V = smpl["v_template"].shape[0]
    ident = _ident3(rot_mats.dtype, rot_mats.device)
    feat = (rot_mats[:, 1:] - ident).reshape(rot_mats.shape[0], -1)
    pd = posedirs.reshape(V * 3, -1)
    if not (pd.is_cuda and feat.shape[0] == 1 and pd.shape[1] <= 256 and pd.dtype == torch.float32):
        raise RuntimeError("pose_offsets: float32 posedirs [V*3, K <= 256] on a HIP device and batch size 1 are required")
    return _RowGemv.apply(pd, feat[0]).view(V, 3)


def shape_offsets(smpl, shapes):
    sd = smpl["shapedirs"][..., :shapes.shape[-1]]
    return torch.matmul(sd, shapes.reshape(-1, 1)).squeeze(-1)


class _FrameConstants:
    """Results that depend only on tensors which stay the same frame after frame (a camera's big-pose parameters, a subject's
    betas): recomputed only when one of the input tensors is another tensor or was written to.  Key = (address, version,
    shape, device, dtype) of every input AND of the SMPL tables the values are computed from (`smpl_keys`: an in-place edit of
    posedirs / shapedirs / J_regressor invalidates the entry); the entry keeps the inputs alive, so an address cannot be handed
    to different data while the entry lives.  Bypassed -- plain recomputation -- for inputs inside the autograd graph, for
    inference tensors (they have no version counter) and while a stream is being captured into a graph (memory of a graph's
    private pool must not end up in a process-wide cache).  LRU, a few hundred entries: two per (camera, subject) pair."""

    def __init__(self, capacity=256):
        from collections import OrderedDict
        self.capacity, self.entries = capacity, OrderedDict()

    @staticmethod
    def _sig(t):
        return (t.data_ptr(), t._version, tuple(t.shape), str(t.device), t.dtype)

    def get(self, tag, smpl, tensors, compute, smpl_keys=()):
        tables = tuple(smpl[k] for k in smpl_keys)
        everything = tuple(tensors) + tables
        if (any(t.requires_grad for t in everything) or torch.is_grad_enabled() and any(t.grad_fn is not None for t in everything)
                or any(t.is_inference() for t in everything) or (everything[0].is_cuda and torch.cuda.is_current_stream_capturing())):
            return compute()
        key = (tag, id(smpl)) + tuple(self._sig(t) for t in everything)
        hit = self.entries.get(key)
        if hit is not None:
            self.entries.move_to_end(key)
            return hit[1]
        with torch.no_grad():
            value = compute()
        self.entries[key] = ((smpl,) + everything, value)
        if len(self.entries) > self.capacity:
            self.entries.popitem(last=False)
        return value


_CONSTANTS = _FrameConstants()


def parents_host(smpl):
    """Kinematic-tree parents as a host tuple, read from the device once per SMPL dict (the reference indexes the device
    tensor joint by joint: 23 blocking reads per chain)."""
    cached = smpl.get("_parents_host")
    if cached is None:
        cached = tuple(int(v) for v in smpl["kintree_table"][0].tolist())
        smpl["_parents_host"] = cached
    return cached


class _SmplPose(torch.autograd.Function):
    """poses [72] (+ correct_Rs [23,3,3]) + joints [24,3] -> (rot_mats [24,3,3], A [24,4,4]); csrc/pose.hip."""

    @staticmethod
    def forward(ctx, poses, correct_Rs, joints, parents):
        if not poses.is_cuda:
            raise RuntimeError("SMPL pose kernel: tensors must live on a HIP device (no CPU path)")
        dev, f32 = poses.device, torch.float32
        c = lambda t: None if t is None else t.detach().contiguous().float()  # noqa: E731
        poses_c, cr, j = c(poses).reshape(72), c(correct_Rs), c(joints).reshape(24, 3)
        if cr is not None:
            cr = cr.reshape(23, 9)
        par = (C.c_int * 24)(*parents)
        rot = torch.empty((24, 3, 3), dtype=f32, device=dev)
        A = torch.empty((24, 4, 4), dtype=f32, device=dev)
        with torch.cuda.device(dev):
            check(lib.gsr_smpl_pose_forward(ptr(poses_c), ptr(cr), ptr(j), par, ptr(rot), ptr(A),
                                            torch.cuda.current_stream(dev).cuda_stream), "gsr_smpl_pose_forward")
        ctx.has_cr = cr is not None
        ctx.save_for_backward(*([poses_c, j] + ([cr] if cr is not None else [])))
        ctx.meta = (parents, poses.shape, None if correct_Rs is None else correct_Rs.shape, joints.shape)
        return rot, A

    @staticmethod
    def backward(ctx, g_rot, g_A):
        saved = ctx.saved_tensors
        poses, j = saved[0], saved[1]
        cr = saved[2] if ctx.has_cr else None
        parents, p_shape, c_shape, j_shape = ctx.meta
        dev, f32 = poses.device, torch.float32
        par = (C.c_int * 24)(*parents)
        need_p, need_c, need_j = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.has_cr, ctx.needs_input_grad[2]
        d_p = torch.empty((72,), dtype=f32, device=dev) if need_p else None
        d_c = torch.empty((23, 9), dtype=f32, device=dev) if need_c else None
        d_j = torch.empty((24, 3), dtype=f32, device=dev) if need_j else None
        g_A = torch.zeros((24, 4, 4), dtype=f32, device=dev) if g_A is None else g_A.contiguous().float()
        g_rot = None if g_rot is None else g_rot.contiguous().float()
        with torch.cuda.device(dev):
            check(lib.gsr_smpl_pose_backward(ptr(poses), ptr(cr), ptr(j), par, ptr(g_A), ptr(g_rot), ptr(d_p), ptr(d_c), ptr(d_j),
                                             torch.cuda.current_stream(dev).cuda_stream), "gsr_smpl_pose_backward")
        return (None if d_p is None else d_p.view(p_shape), None if d_c is None else d_c.view(c_shape),
                None if d_j is None else d_j.view(j_shape), None)


def smpl_pose_transforms(smpl, params, correct_Rs=None):
    """HIP counterpart of batch_rodrigues + get_transform_params_torch for batch size 1:
    returns (A [1,24,4,4], rot_mats [1,24,3,3], joints [1,24,3])."""
    betas = params["shapes"]
    nb = int(betas.shape[-1])
    # joints = J_regressor (v_template + shapedirs beta) = J_template + J_shapedirs beta: the two regressed tables are
    # constants of the SMPL model and are cached on the dict (the per-frame [24 x 6890] x [6890 x 3] product is a one-tile,
    # K = 6890 rocBLAS launch of 60 us)
    cache = smpl.get("_joint_tables")
    sig = (nb,) + tuple(_FrameConstants._sig(smpl[k]) for k in ("J_regressor", "v_template", "shapedirs"))
    if cache is None or cache[0] != sig:
        with torch.no_grad():
            Jt = torch.matmul(smpl["J_regressor"], smpl["v_template"])                                       # [24, 3]
            Js = torch.einsum("jv,vcl->jcl", smpl["J_regressor"], smpl["shapedirs"][..., :nb].float())       # [24, 3, nb]
        cache = (sig, Jt.contiguous(), Js.contiguous())
        smpl["_joint_tables"] = cache
    # (a subject's betas are the same tensor frame after frame: the small product is cached on it)
    joints = _CONSTANTS.get("joints", smpl, (betas,),
                            lambda: cache[1] + torch.matmul(cache[2], betas.reshape(-1, 1).float()).squeeze(-1),
                            smpl_keys=("J_regressor", "v_template", "shapedirs"))
    rot, A = _SmplPose.apply(params["poses"], correct_Rs, joints, parents_host(smpl))
    return A[None], rot[None], joints[None]


NEAREST_VERTEX_SEARCH = "grid"  # "grid" (uniform vertex grid, default) or "brute"; identical results


class _VertexGridCache:
    """Grid workspaces of reference-vertex tensors that were seen before.  The big-pose vertices of a subject are the same
    tensor frame after frame (one per camera, scene/cameras.py:72), so their 46 us single-workgroup grid build is done once.
    Key = (storage address, tensor version, shape): any in-place change bumps the version; the entry holds a reference to the
    tensor, so its address cannot be handed to another tensor while the entry lives.  Memory: ~0.7 MB per entry, LRU."""

    def __init__(self, capacity=1024):
        from collections import OrderedDict
        self.capacity, self.entries = capacity, OrderedDict()

    def get(self, verts):
        key = (verts.data_ptr(), verts._version, tuple(verts.shape), str(verts.device))
        hit = self.entries.get(key)
        if hit is not None:  # hit[0] keeps the storage alive, so the address cannot belong to different data at this version
            self.entries.move_to_end(key)
            return hit[1], True
        ws = torch.empty((lib.gsr_lbs_workspace_bytes(verts.shape[0]),), dtype=torch.uint8, device=verts.device)
        self.entries[key] = (verts, ws, {})
        if len(self.entries) > self.capacity:
            self.entries.popitem(last=False)
        return ws, False

    def nn_cache(self, verts, P):
        """The temporal nearest-vertex cache of P query points against this vertex tensor (csrc/lbs.hip "exact temporal cache"):
        (buffer, valid).  An entry of it is a statement about the vertex set alone, so it lives and dies with the vertex grid --
        a new vertex tensor (or an in-place change: _version) is a new key and starts empty; densify / prune change P and get a
        buffer of their own."""
        key = (verts.data_ptr(), verts._version, tuple(verts.shape), str(verts.device))
        caches = self.entries[key][2]
        hit = caches.get(P)
        if hit is not None:
            return hit, True
        if len(caches) >= 4:   # (a model that changes size every few hundred iterations: keep the newest)
            caches.pop(next(iter(caches)))
        buf = torch.empty((lib.gsr_lbs_nn_cache_bytes(P),), dtype=torch.uint8, device=verts.device)
        caches[P] = buf
        return buf, False

    def nn_cache_stats(self, verts, P):
        """(misses of the last cached frame, searches since the cache was made) or None -- for tests and tools (synchronises)."""
        key = (verts.data_ptr(), verts._version, tuple(verts.shape), str(verts.device))
        e = self.entries.get(key)
        if e is None or P not in e[2]:
            return None
        w = e[2][P][-64:].view(torch.int32)[:3].cpu()
        return int(w[2]), int(w[1])


_GRIDS = _VertexGridCache()
# exact temporal cache of the nearest vertex (results identical to the search, tests/test_gpu_lbs.py); False: search every frame
NN_TEMPORAL_CACHE = True


class _LBSDeform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, normals, lbs_offsets, A_big, A_pose, off_big, off_shape, off_pose, R, Th, smpl_verts, weights,
                lean=False):
        if not query.is_cuda:
            raise RuntimeError("LBS deform: tensors must live on a HIP device (no CPU path)")
        dev, f32 = query.device, torch.float32
        P, V = query.shape[0], smpl_verts.shape[0]
        c = lambda t: None if t is None else t.detach().contiguous().float()  # noqa: E731
        query_c, normals_c, loff = c(query), c(normals), c(lbs_offsets)
        A_big_c, A_pose_c = c(A_big).reshape(24, 16), c(A_pose).reshape(24, 16)
        ob, os_, op = c(off_big), c(off_shape), c(off_pose)
        R_c, Th_c, sv, w = c(R).reshape(3, 3), c(Th).reshape(3), c(smpl_verts), c(weights)
        vert_ids = torch.empty((P,), dtype=torch.int32, device=dev)
        # lean: the caller (render()) only consumes world points / transforms / normals: skip 120 B per point of stores
        bweights = None if lean else torch.empty((P, 24), dtype=f32, device=dev)
        smpl_pts = None if lean else torch.empty((P, 3), dtype=f32, device=dev)
        world_pts = torch.empty((P, 3), dtype=f32, device=dev)
        transforms = torch.empty((P, 3, 3), dtype=f32, device=dev)
        translation = None if lean else torch.empty((P, 3), dtype=f32, device=dev)
        world_normals = torch.empty((P, 3), dtype=f32, device=dev) if normals is not None else None
        args = (P, V, ptr(query_c), ptr(normals_c), ptr(sv), ptr(w), ptr(loff), ptr(A_big_c), ptr(A_pose_c), ptr(ob), ptr(os_),
                ptr(op), ptr(R_c), ptr(Th_c), ptr(vert_ids), ptr(bweights), ptr(smpl_pts), ptr(world_pts), ptr(transforms),
                ptr(translation), ptr(world_normals))
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if NEAREST_VERTEX_SEARCH == "grid":
                if sv.data_ptr() == smpl_verts.data_ptr():   # the caller's own storage (no conversion copy): remember its grid
                    ws, built = _GRIDS.get(smpl_verts)
                else:
                    ws, built = torch.empty((lib.gsr_lbs_workspace_bytes(V),), dtype=torch.uint8, device=dev), False
                if NN_TEMPORAL_CACHE and sv.data_ptr() == smpl_verts.data_ptr():
                    # frame-to-frame: a point that has not left the ball its entry vouches for keeps its vertex without a search
                    if not built:
                        check(lib.gsr_lbs_grid_build(V, ptr(sv), ptr(ws), ws.numel(), stream), "gsr_lbs_grid_build")
                    nn, valid = _GRIDS.nn_cache(smpl_verts, P)
                    check(lib.gsr_lbs_forward_cached(*args, ptr(ws), ws.numel(), ptr(nn), nn.numel(), int(valid), stream),
                          "gsr_lbs_forward_cached")
                else:
                    check(lib.gsr_lbs_forward_grid(*args, ptr(ws), ws.numel(), int(built), stream), "gsr_lbs_forward_grid")
            else:
                check(lib.gsr_lbs_forward(*args, stream), "gsr_lbs_forward")
        ctx.save_for_backward(query_c, normals_c, loff, A_big_c, A_pose_c, ob, os_, op, R_c, vert_ids, w)
        ctx.shapes = (A_pose.shape, off_pose.shape, V)
        e = torch.empty(0, device=dev)
        smpl_pts, bweights, translation = (e if smpl_pts is None else smpl_pts, e if bweights is None else bweights,
                                           e if translation is None else translation)
        ctx.mark_non_differentiable(vert_ids, bweights, smpl_pts, translation)
        return (world_pts, transforms, world_normals if world_normals is not None else torch.empty(0, device=dev), smpl_pts,
                bweights, translation, vert_ids)

    @staticmethod
    def backward(ctx, g_world, g_transforms, g_normals, *_unused):
        query, normals, loff, A_big, A_pose, ob, os_, op, R, vert_ids, w = ctx.saved_tensors
        A_shape, off_shape_, V = ctx.shapes
        dev, f32 = query.device, torch.float32
        P = query.shape[0]
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        g_world, g_transforms = c(g_world), c(g_transforms)
        g_normals = c(g_normals) if (normals is not None and g_normals is not None and g_normals.numel()) else None
        d_query = torch.empty((P, 3), dtype=f32, device=dev)
        d_normals = torch.empty((P, 3), dtype=f32, device=dev) if normals is not None else None
        d_loff = torch.empty((P, 24), dtype=f32, device=dev) if loff is not None else None
        # A_pose / off_pose gradients only when the pose path is trainable (pose-refinement MLP): their reductions are atomics
        # of every workgroup onto 288 + 3V addresses
        need_A, need_off = ctx.needs_input_grad[4], ctx.needs_input_grad[7]
        d_A = torch.zeros((24, 16), dtype=f32, device=dev) if need_A else None
        partials = torch.empty((lib.gsr_lbs_backward_workgroups(P), 24 * 12), dtype=f32, device=dev) if need_A else None
        d_off = torch.zeros((V, 3), dtype=f32, device=dev) if need_off else None
        with torch.cuda.device(dev):
            check(lib.gsr_lbs_backward(P, V, ptr(query), ptr(normals), ptr(vert_ids), ptr(w), ptr(loff), ptr(A_big), ptr(A_pose),
                                       ptr(ob), ptr(os_), ptr(op), ptr(R), ptr(g_world), ptr(g_transforms), ptr(g_normals),
                                       ptr(d_query), ptr(d_normals), ptr(d_loff), ptr(d_A), ptr(d_off), ptr(partials),
                                       torch.cuda.current_stream(dev).cuda_stream), "gsr_lbs_backward")
        if need_A:  # per-workgroup sums -> rows 0..2 of the 24 4x4 gradients (row 3 of A is constant)
            d_A.view(24, 4, 4)[:, :3, :] = partials.sum(0).view(24, 3, 4)
        return (d_query, d_normals, d_loff, None, None if d_A is None else d_A.view(A_shape), None, None,
                None if d_off is None else d_off.view(off_shape_), None, None, None, None, None)


def lbs_deform(query, normals, lbs_offsets, A_big, A_pose, off_big, off_shape, off_pose, R, Th, smpl_verts, weights, lean=False):
    """Per-point LBS (HIP). Returns dict(world_pts, transforms, world_normals, smpl_pts, bweights, translation, vert_ids);
    lean=True leaves smpl_pts / bweights / translation empty (not computed)."""
    o = _LBSDeform.apply(query, normals, lbs_offsets, A_big, A_pose, off_big, off_shape, off_pose, R, Th, smpl_verts, weights, lean)
    return dict(world_pts=o[0], transforms=o[1], world_normals=o[2] if normals is not None else None, smpl_pts=o[3],
                bweights=o[4], translation=o[5], vert_ids=o[6])


def coarse_deform_c2source(smpl, query_pts, params, t_params, t_vertices, lbs_weights=None, correct_Rs=None,
                           return_transl=False, normals=None, lean=False):
    """Drop-in for GaussianModel.coarse_deform_c2source (batch size 1): same arguments after `smpl`
    (= self.SMPL_NEUTRAL as device tensors) and the same 6-tuple
    (smpl_src_pts[1,P,3], world_src_pts[1,P,3], bweights[1,P,24], transforms[1,P,3,3], translation|None, world_normals)."""
    assert query_pts.shape[0] == 1, "batch size 1 (like every call site of the reference)"
    # big pose -> T pose, T pose -> target pose: one single-wave kernel each (csrc/pose.hip)
    # the big-pose side depends on the camera's big_pose_smpl_param alone, the shape offsets on the subject's betas: both are
    # the same tensors frame after frame (scene/cameras.py:44-74) and are computed once per tensor (pose kernel + 17 MB
    # pose-blend-shape product + two small rocBLAS products per frame otherwise)
    def big_pose():
        A, rot, _ = smpl_pose_transforms(smpl, t_params)
        return A, pose_offsets(smpl, rot)
    A_big, off_big = _CONSTANTS.get("big_pose", smpl, (t_params["poses"], t_params["shapes"]), big_pose,
                                    smpl_keys=("v_template", "shapedirs", "J_regressor", "posedirs"))
    A_pose, rot_mats, _ = smpl_pose_transforms(smpl, params, correct_Rs)
    R, Th = params["R"], params["Th"]
    # keyed on the caller's OWN tensor (before any .to(device): a host tensor would be a new device copy -- a new address,
    # never a hit -- every frame)
    shapes_in = params["shapes"]
    off_shape = _CONSTANTS.get("shape_offsets", smpl, (shapes_in,),
                               lambda: shape_offsets(smpl, shapes_in.to(query_pts.device)), smpl_keys=("shapedirs",))
    off_pose = pose_offsets(smpl, rot_mats)
    # (batch size 1: reshape, not [0] -- the backward of a select materialises a zero [1, P, 3] tensor and copies into it)
    P_ = query_pts.shape[1]
    o = lbs_deform(query_pts.reshape(P_, 3), None if normals is None else normals.reshape(P_, 3),
                   None if lbs_weights is None else lbs_weights.reshape(P_, -1),
                   A_big.reshape(24, 4, 4), A_pose.reshape(24, 4, 4), off_big, off_shape, off_pose, R.reshape(3, 3), Th.reshape(-1)[:3],
                   t_vertices.reshape(-1, 3), smpl["weights"], lean=lean and not return_transl)
    translation = o["translation"][None] if return_transl else None
    wn = None if o["world_normals"] is None else o["world_normals"][None]
    return o["smpl_pts"][None], o["world_pts"][None], o["bweights"][None], o["transforms"][None], translation, wn


def smplx_lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """The vendored smplx `lbs()` of the reference (smplx/lbs.py:156-252, fork returns 4 values): BASELINE config 1.
    Device-agnostic torch, batch-first: betas [B,NB], pose [B,72], v_template [1,V,3] or [V,3], shapedirs [V,3,NB],
    posedirs [207, V*3], J_regressor [24,V], parents [24], lbs_weights [V,24] -> (verts [B,V,3], J_transformed [B,24,3],
    A [B,24,4,4], T [B,V,4,4])."""
    B = max(betas.shape[0], pose.shape[0])
    vt = v_template if v_template.dim() == 3 else v_template[None]
    v_shaped = vt + torch.einsum("bl,mkl->bmk", betas, shapedirs[..., :betas.shape[-1]])
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)
    rot_mats = batch_rodrigues(pose.reshape(-1, 3)).view(B, -1, 3, 3)
    ident = torch.eye(3, dtype=pose.dtype, device=pose.device)
    pose_feature = (rot_mats[:, 1:] - ident).reshape(B, -1)
    v_posed = torch.matmul(pose_feature, posedirs).view(B, -1, 3) + v_shaped
    # kinematic chain (batch_rigid_transform, :349-405)
    rel = J.clone()
    rel[:, 1:] = rel[:, 1:] - J[:, parents[1:]]
    tm = torch.cat([torch.cat([rot_mats, rel[..., None]], dim=-1),
                    torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=pose.dtype, device=pose.device).expand(B, J.shape[1], 1, 4)], dim=-2)
    chain = [tm[:, 0]]
    for i in range(1, J.shape[1]):
        chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
    tr = torch.stack(chain, dim=1)
    J_transformed = tr[:, :, :3, 3]
    jh = torch.cat([J, torch.zeros_like(J[..., :1])], dim=-1)[..., None]
    A = tr - torch.nn.functional.pad(torch.matmul(tr, jh), [3, 0])
    T = torch.matmul(lbs_weights[None].expand(B, -1, -1), A.view(B, J.shape[1], 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones_like(v_posed[..., :1])], dim=2)
    verts = torch.matmul(T, vh[..., None])[:, :, :3, 0]
    return verts, J_transformed, A, T
