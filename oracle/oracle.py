"""ctypes/numpy front-end of the CPU oracle (oracle/gsr_oracle.c, oracle/lbs_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (mygauhuman_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSR_ORACLE_LIB selects another build of the same sources (the sanitizer build of the CPU test leg)
_LIB_PATH = os.environ.get("GSR_ORACLE_LIB") or os.path.join(_HERE, "libgsr_oracle.so")
_lib = None

f32 = np.float32
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_bp = C.POINTER(C.c_ubyte)


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    if os.environ.get("GSR_ORACLE_LIB"):
        return _LIB_PATH
    srcs = [os.path.join(_HERE, s) for s in ("gsr_oracle.c", "lbs_oracle.c")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_scan.restype = C.c_uint32
        _lib.oracle_higher_msb.restype = C.c_uint32
        _lib.oracle_get_max_threads.restype = C.c_int
    return _lib


def _p(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(t)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def set_threads(n):
    lib().oracle_set_threads(int(n))


def set_thresholds(alpha_min=1.0 / 255.0, T_min=0.0001):
    """Self-check only: smooth the renderer (0, 0); defaults are the reference constants."""
    lib().oracle_set_thresholds(C.c_float(alpha_min), C.c_float(T_min))


def max_threads():
    return lib().oracle_get_max_threads()


def higher_msb(n):
    return int(lib().oracle_higher_msb(C.c_uint32(n)))


# --------------------------------------------------------------------------- rasterizer
def preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, *, scales=None,
               rotations=None, scale_modifier=1.0, shs=None, degree=0, cov3D_precomp=None, colors_precomp=None):
    means3D = _c(means3D, f32)
    P = means3D.shape[0]
    shs = _c(shs, f32)
    M = 0 if shs is None else shs.shape[1]
    o = dict(
        radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), f32), depths=np.zeros(P, f32),
        cov3D=np.zeros((P, 6), f32), rgb=np.zeros((P, 3), f32), conic_opacity=np.zeros((P, 4), f32),
        tiles_touched=np.zeros(P, np.uint32), clamped=np.zeros((P, 3), np.uint8))
    scales, rotations = _c(scales, f32), _c(rotations, f32)
    cov3D_precomp, colors_precomp = _c(cov3D_precomp, f32), _c(colors_precomp, f32)
    lib().oracle_preprocess(
        C.c_int(P), C.c_int(degree), C.c_int(M), _p(means3D, _fp), _p(scales, _fp), C.c_float(scale_modifier),
        _p(rotations, _fp), _p(_c(opacities, f32), _fp), _p(shs, _fp), _p(cov3D_precomp, _fp),
        _p(colors_precomp, _fp), _p(_c(viewmatrix, f32), _fp), _p(_c(projmatrix, f32), _fp),
        _p(_c(campos, f32), _fp), C.c_int(W), C.c_int(H), C.c_float(tanfovx), C.c_float(tanfovy),
        _p(o["radii"], _ip), _p(o["means2D"], _fp), _p(o["depths"], _fp), _p(o["cov3D"], _fp), _p(o["rgb"], _fp),
        _p(o["conic_opacity"], _fp), _p(o["tiles_touched"], _up), _p(o["clamped"], _bp))
    if cov3D_precomp is not None:
        o["cov3D"] = cov3D_precomp
    if colors_precomp is not None:
        o["rgb"] = colors_precomp
    return o


def bin_tiles(pre, W, H):
    P = pre["radii"].shape[0]
    offsets = np.zeros(P, np.uint32)
    R = int(lib().oracle_scan(C.c_int(P), _p(pre["tiles_touched"], _up), _p(offsets, _up)))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    o = dict(offsets=offsets, R=R, keys_unsorted=np.zeros(R, np.uint64), values_unsorted=np.zeros(R, np.uint32),
             keys_sorted=np.zeros(R, np.uint64), point_list=np.zeros(R, np.uint32),
             ranges=np.zeros((tiles, 2), np.uint32))
    lib().oracle_bin(C.c_int(P), C.c_int(W), C.c_int(H), _p(pre["means2D"], _fp), _p(pre["depths"], _fp),
                     _p(pre["radii"], _ip), _p(offsets, _up), C.c_uint32(R), _p(o["keys_unsorted"], _u64p),
                     _p(o["values_unsorted"], _up), _p(o["keys_sorted"], _u64p), _p(o["point_list"], _up),
                     _p(o["ranges"], _up))
    return o


def blend_forward(pre, binned, W, H, bg):
    o = dict(color=np.zeros((3, H, W), f32), depth=np.zeros((1, H, W), f32), alpha=np.zeros((1, H, W), f32),
             final_T=np.zeros((H, W), f32), n_contrib=np.zeros((H, W), np.uint32),
             fragile=np.zeros((H, W), np.uint8))
    feats = _c(pre["rgb"], f32)
    lib().oracle_blend_forward(
        C.c_int(W), C.c_int(H), _p(binned["ranges"], _up), _p(binned["point_list"], _up), _p(pre["means2D"], _fp),
        _p(feats, _fp), _p(pre["depths"], _fp), _p(pre["conic_opacity"], _fp), _p(_c(bg, f32), _fp),
        _p(o["color"], _fp), _p(o["depth"], _fp), _p(o["alpha"], _fp), _p(o["final_T"], _fp),
        _p(o["n_contrib"], _up), _p(o["fragile"], _bp))
    return o


def rasterize_forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, bg, **kw):
    pre = preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, **kw)
    binned = bin_tiles(pre, W, H)
    img = blend_forward(pre, binned, W, H, bg)
    return dict(pre=pre, bin=binned, img=img, W=W, H=H, bg=_c(bg, f32), kw=kw,
                inputs=dict(means3D=_c(means3D, f32), viewmatrix=_c(viewmatrix, f32),
                            projmatrix=_c(projmatrix, f32), campos=_c(campos, f32), tanfovx=tanfovx,
                            tanfovy=tanfovy))


def rasterize_backward(fwd, dL_dcolor, dL_ddepth, dL_dalpha):
    pre, binned, img, W, H = fwd["pre"], fwd["bin"], fwd["img"], fwd["W"], fwd["H"]
    kw, inp = fwd["kw"], fwd["inputs"]
    P = pre["radii"].shape[0]
    g = dict(dL_dmean2D=np.zeros((P, 3), f32), dL_dconic=np.zeros((P, 4), f32), dL_dopacity=np.zeros((P, 1), f32),
             dL_dcolors=np.zeros((P, 3), f32))
    lib().oracle_blend_backward(
        C.c_int(P), C.c_int(W), C.c_int(H), _p(binned["ranges"], _up), _p(binned["point_list"], _up),
        _p(fwd["bg"], _fp), _p(pre["means2D"], _fp), _p(pre["conic_opacity"], _fp), _p(_c(pre["rgb"], f32), _fp),
        _p(pre["depths"], _fp), _p(img["final_T"], _fp), _p(img["n_contrib"], _up),
        _p(_c(dL_dcolor, f32), _fp), _p(_c(dL_ddepth, f32), _fp), _p(_c(dL_dalpha, f32), _fp),
        _p(g["dL_dmean2D"], _fp), _p(g["dL_dconic"], _fp), _p(g["dL_dopacity"], _fp), _p(g["dL_dcolors"], _fp))
    shs = _c(kw.get("shs"), f32)
    M = 0 if shs is None else shs.shape[1]
    scales, rotations = _c(kw.get("scales"), f32), _c(kw.get("rotations"), f32)
    g.update(dL_dmeans3D=np.zeros((P, 3), f32), dL_dcov3D=np.zeros((P, 6), f32), dL_dsh=np.zeros((P, M, 3), f32),
             dL_dscales=np.zeros((P, 3), f32), dL_drotations=np.zeros((P, 4), f32))
    lib().oracle_preprocess_backward(
        C.c_int(P), C.c_int(kw.get("degree", 0)), C.c_int(M), _p(inp["means3D"], _fp), _p(pre["radii"], _ip),
        _p(shs, _fp), _p(pre["clamped"], _bp), _p(scales, _fp), _p(rotations, _fp),
        C.c_float(kw.get("scale_modifier", 1.0)), _p(_c(pre["cov3D"], f32), _fp), _p(inp["viewmatrix"], _fp),
        _p(inp["projmatrix"], _fp), C.c_int(W), C.c_int(H), C.c_float(inp["tanfovx"]), C.c_float(inp["tanfovy"]),
        _p(inp["campos"], _fp), _p(g["dL_dmean2D"], _fp), _p(g["dL_dconic"], _fp), _p(g["dL_dcolors"], _fp),
        _p(g["dL_dmeans3D"], _fp), _p(g["dL_dcov3D"], _fp), _p(g["dL_dsh"], _fp), _p(g["dL_dscales"], _fp),
        _p(g["dL_drotations"], _fp))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _c(means3D, f32)
    out = np.zeros(means3D.shape[0], np.uint8)
    lib().oracle_mark_visible(C.c_int(means3D.shape[0]), _p(means3D, _fp), _p(_c(viewmatrix, f32), _fp),
                              _p(_c(projmatrix, f32), _fp), _p(out, _bp))
    return out.astype(bool)


# --------------------------------------------------------------------------- simple-knn
def dist2_brute(points):
    points = _c(points, f32)
    out = np.zeros(points.shape[0], f32)
    lib().oracle_dist2_brute(C.c_int(points.shape[0]), _p(points, _fp), _p(out, _fp))
    return out


def dist2_morton(points):
    points = _c(points, f32)
    P = points.shape[0]
    out, codes, order = np.zeros(P, f32), np.zeros(P, np.uint32), np.zeros(P, np.uint32)
    lib().oracle_dist2_morton(C.c_int(P), _p(points, _fp), _p(out, _fp), _p(codes, _up), _p(order, _up))
    return out, codes, order


def knn_self(points, k):
    """(idx [P,k] int32, dist [P,k]) of the k <= 3 nearest among the same points, self included, brute force."""
    points = _c(points, f32)
    P = points.shape[0]
    idx, dist = np.zeros((P, k), np.int32), np.zeros((P, k), f32)
    lib().oracle_knn_self(C.c_int(P), _p(points, _fp), C.c_int(k), _p(idx, _ip), _p(dist, _fp))
    return idx, dist


def knn_self_boxes(points, k):
    """knn_self's result through Morton-sorted boxes (exact, seconds at 500k points)."""
    points = _c(points, f32)
    P = points.shape[0]
    idx, dist = np.zeros((P, k), np.int32), np.zeros((P, k), f32)
    lib().oracle_knn_self_boxes(C.c_int(P), _p(points, _fp), C.c_int(k), _p(idx, _ip), _p(dist, _fp))
    return idx, dist


def nearest_dist(query, verts, idx):
    query, verts, idx = _c(query, f32), _c(verts, f32), np.ascontiguousarray(idx, np.int32)
    out = np.zeros(query.shape[0], f32)
    lib().oracle_nearest_dist(C.c_int(query.shape[0]), _p(query, _fp), C.c_int(verts.shape[0]), _p(verts, _fp), _p(idx, _ip),
                              _p(out, _fp))
    return out


def nearest_vertex(query, verts):
    query, verts = _c(query, f32), _c(verts, f32)
    out = np.zeros(query.shape[0], np.int32)
    lib().oracle_nearest_vertex(C.c_int(query.shape[0]), _p(query, _fp), C.c_int(verts.shape[0]), _p(verts, _fp),
                                _p(out, _ip))
    return out


# --------------------------------------------------------------------------- LBS
def rodrigues(rotvec):
    rotvec = _c(rotvec, f32).reshape(-1, 3)
    out = np.zeros((rotvec.shape[0], 3, 3), f32)
    lib().oracle_rodrigues(C.c_int(rotvec.shape[0]), _p(rotvec, _fp), _p(out, _fp))
    return out


def joint_transforms(smpl, betas, rot_mats):
    """smpl: dict(v_template[V,3], shapedirs[V,3,NB], J_regressor[24,V], parents int32[24])."""
    vt, sd = _c(smpl["v_template"], f32), _c(smpl["shapedirs"], f32)
    betas = _c(betas, f32).reshape(-1)
    sd = np.ascontiguousarray(sd[..., :betas.shape[0]])
    A, joints = np.zeros((24, 4, 4), f32), np.zeros((24, 3), f32)
    lib().oracle_joint_transforms(C.c_int(vt.shape[0]), C.c_int(betas.shape[0]), _p(vt, _fp), _p(sd, _fp),
                                  _p(_c(smpl["J_regressor"], f32), _fp), _p(_c(smpl["parents"], np.int32), _ip),
                                  _p(betas, _fp), _p(_c(rot_mats, f32), _fp), _p(A, _fp), _p(joints, _fp))
    return A, joints


def pose_offsets(posedirs_vk, rot_mats):
    pd = _c(posedirs_vk, f32).reshape(-1, 207)
    V = pd.shape[0] // 3
    out = np.zeros((V, 3), f32)
    lib().oracle_pose_offsets(C.c_int(V), _p(pd, _fp), _p(_c(rot_mats, f32), _fp), _p(out, _fp))
    return out


def shape_offsets(shapedirs, betas):
    betas = _c(betas, f32).reshape(-1)
    sd = np.ascontiguousarray(_c(shapedirs, f32)[..., :betas.shape[0]])
    out = np.zeros((sd.shape[0], 3), f32)
    lib().oracle_shape_offsets(C.c_int(sd.shape[0]), C.c_int(betas.shape[0]), _p(sd, _fp), _p(betas, _fp),
                               _p(out, _fp))
    return out


def lbs_deform(query, normals, vert_ids, weights, A_big, A_pose, off_big, off_shape, off_pose, R, Th, lbs_off=None):
    query = _c(query, f32)
    P = query.shape[0]
    o = dict(smpl_src=np.zeros((P, 3), f32), world_src=np.zeros((P, 3), f32), bweights=np.zeros((P, 24), f32),
             transforms=np.zeros((P, 3, 3), f32), translation=np.zeros((P, 3), f32),
             world_normals=np.zeros((P, 3), f32))
    lib().oracle_lbs_deform(
        C.c_int(P), _p(query, _fp), _p(_c(normals, f32), _fp), _p(_c(vert_ids, np.int32), _ip),
        _p(_c(weights, f32), _fp), _p(_c(lbs_off, f32), _fp), _p(_c(A_big, f32), _fp), _p(_c(A_pose, f32), _fp),
        _p(_c(off_big, f32), _fp), _p(_c(off_shape, f32), _fp), _p(_c(off_pose, f32), _fp), _p(_c(R, f32), _fp),
        _p(_c(Th, f32).reshape(-1), _fp), _p(o["smpl_src"], _fp), _p(o["world_src"], _fp), _p(o["bweights"], _fp),
        _p(o["transforms"], _fp), _p(o["translation"], _fp), _p(o["world_normals"], _fp))
    return o


def smpl_lbs(betas, pose, v_template, shapedirs, posedirs_kv, J_regressor, parents, lbs_weights):
    vt = _c(v_template, f32).reshape(-1, 3)
    V = vt.shape[0]
    betas = _c(betas, f32).reshape(-1)
    sd = np.ascontiguousarray(_c(shapedirs, f32)[..., :betas.shape[0]])
    verts, J, A, T = np.zeros((V, 3), f32), np.zeros((24, 3), f32), np.zeros((24, 4, 4), f32), np.zeros((V, 4, 4), f32)
    lib().oracle_smpl_lbs(C.c_int(V), C.c_int(betas.shape[0]), _p(betas, _fp), _p(_c(pose, f32).reshape(-1), _fp),
                          _p(vt, _fp), _p(sd, _fp), _p(_c(posedirs_kv, f32), _fp), _p(_c(J_regressor, f32), _fp),
                          _p(_c(parents, np.int32), _ip), _p(_c(lbs_weights, f32), _fp), _p(verts, _fp), _p(J, _fp),
                          _p(A, _fp), _p(T, _fp))
    return verts, J, A, T


def project(points, full_proj):
    points = _c(points, f32)
    out = np.zeros_like(points)
    lib().oracle_project(C.c_int(points.shape[0]), _p(points, _fp), _p(_c(full_proj, f32), _fp), _p(out, _fp))
    return out
