"""Generate golden vectors from the IMPORTED reference Python (runs only in the build container).

Usage:  python tests/golden/make_golden.py          (needs /root/reference on disk)

Only the pieces of the hot path that exist as importable, device-agnostic reference Python are covered:
  utils/sh_utils.py:eval_sh                     -> sh_eval.npz      (pins the SH polynomial of computeColorFromSH)
  smplx/lbs.py: lbs, batch_rodrigues,
                batch_rigid_transform            -> lbs_smpl.npz     (pins Rodrigues, the kinematic chain, skinning)
  utils/graphics_utils.py: getWorld2View2,
        getProjectionMatrix_refine,
        geom_transform_points, focal2fov         -> camera.npz       (pins the matrix conventions + projection)
The rasterizer / simple-knn are CUDA-only and cannot run here: no reference outputs exist for them
("parity unpinned", see oracle/gsr_oracle.c).  The files written are data (inputs + expected outputs), not code.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from smplx.lbs import batch_rodrigues, lbs  # noqa: E402
from utils.graphics_utils import (focal2fov, geom_transform_points, getProjectionMatrix_refine,  # noqa: E402
                                  getWorld2View2)
from utils.sh_utils import eval_sh  # noqa: E402

PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int64)


def synthetic_smpl(V, seed):
    """Seeded SMPL-shaped model: the real SMPL pkl is not redistributable/offline."""
    rng = np.random.default_rng(seed)
    v_template = rng.uniform(-1, 1, (V, 3)).astype(np.float32) * np.array([0.9, 0.9, 0.15], np.float32)
    shapedirs = (rng.normal(0, 0.01, (V, 3, 10))).astype(np.float32)
    posedirs = (rng.normal(0, 0.001, (207, V * 3))).astype(np.float32)
    J_regressor = rng.uniform(0, 1, (24, V)).astype(np.float32)
    J_regressor /= J_regressor.sum(1, keepdims=True)
    weights = rng.uniform(0, 1, (V, 24)).astype(np.float32) ** 4
    weights /= weights.sum(1, keepdims=True)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor,
                weights=weights.astype(np.float32), parents=PARENTS)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    rng = np.random.default_rng(0)

    # ---- eval_sh
    N = 257
    sh = rng.normal(0, 0.5, (N, 3, 16)).astype(np.float32)
    dirs = rng.normal(0, 1, (N, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out = {"sh": sh, "dirs": dirs}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = eval_sh(deg, torch.from_numpy(sh), torch.from_numpy(dirs)).numpy()
    np.savez_compressed(os.path.join(HERE, "sh_eval.npz"), **out)

    # ---- smplx lbs
    V = 431
    m = synthetic_smpl(V, 1)
    betas = rng.normal(0, 1, (1, 10)).astype(np.float32)
    pose = rng.normal(0, 0.2, (1, 72)).astype(np.float32)
    verts, Jt, A, T = lbs(torch.from_numpy(betas), torch.from_numpy(pose), torch.from_numpy(m["v_template"])[None],
                          torch.from_numpy(m["shapedirs"]), torch.from_numpy(m["posedirs"]),
                          torch.from_numpy(m["J_regressor"]), torch.from_numpy(PARENTS),
                          torch.from_numpy(m["weights"]))
    rot = batch_rodrigues(torch.from_numpy(pose).view(-1, 3))
    tiny = np.array([[0, 0, 0], [1e-9, 0, 0], [0, 1e-4, 0], [3.0, -2.0, 1.0]], np.float32)
    np.savez_compressed(os.path.join(HERE, "lbs_smpl.npz"), betas=betas, pose=pose, verts=verts.numpy()[0],
                        J_transformed=Jt.numpy()[0], A=A.numpy()[0], T=T.numpy()[0], rot_mats=rot.numpy(),
                        rodrigues_in=tiny, rodrigues_out=batch_rodrigues(torch.from_numpy(tiny)).numpy(),
                        smpl_seed=np.int64(1), smpl_V=np.int64(V),
                        **{"smpl_" + k: v for k, v in m.items()})

    # ---- camera conventions + CPU projection
    W, H = 640, 480
    K = np.array([[700.5, 0.3, 322.25], [0, 690.25, 236.5], [0, 0, 1]], np.float32)
    ang = 0.3
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    Tt = np.array([0.1, -0.2, 3.0])
    w2v = getWorld2View2(R, Tt, np.array([0.0, 0.0, 0.0]), 1.0)
    w2v_ts = getWorld2View2(R, Tt, np.array([0.5, -0.25, 0.125]), 1.5)
    proj = getProjectionMatrix_refine(torch.from_numpy(K), H, W, 0.001, 1000).numpy()
    view_T = torch.tensor(w2v).transpose(0, 1)
    proj_T = torch.from_numpy(proj).transpose(0, 1)
    full = view_T.unsqueeze(0).bmm(proj_T.unsqueeze(0)).squeeze(0)
    center = view_T.inverse()[3, :3]
    pts = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    ndc = geom_transform_points(torch.from_numpy(pts), full).numpy()
    np.savez_compressed(os.path.join(HERE, "camera.npz"), W=np.int64(W), H=np.int64(H), K=K, R=R, T=Tt, w2v=w2v,
                        w2v_translate_scale=w2v_ts, proj=proj, full_proj=full.numpy(), camera_center=center.numpy(),
                        fovx=np.float64(focal2fov(float(K[0, 0]), W)), fovy=np.float64(focal2fov(float(K[1, 1]), H)),
                        pts=pts, ndc=ndc)
    # ---- PLY reader / writer: the reference's own data file check/points3d.ply (an SMPL-shaped initial point cloud written by
    # plyfile through storePly, scene/dataset_readers.py:138-153) pins the on-disk format.  Committed: its first 256 vertices
    # as a valid PLY (header count rewritten) + size / sha256 / bounding box / mean colour of the whole file.
    import hashlib
    import json
    src = os.path.join(REF, "check", "points3d.ply")
    data = open(src, "rb").read()
    end = data.find(b"end_header\n") + len(b"end_header\n")
    header = data[:end].decode("ascii")
    n_total = int(header.split("element vertex ")[1].split()[0])
    with open(os.path.join(HERE, "points3d_head256.ply"), "wb") as f:
        f.write(header.replace(f"element vertex {n_total}", "element vertex 256").encode("ascii") + data[end:end + 256 * 27])
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from mygauhuman_amd import ply_io
    v = ply_io.read_ply(src)
    json.dump(dict(source="check/points3d.ply of the reference (binary_little_endian, written by plyfile through storePly)",
                   vertices=int(len(v)), bytes=len(data), sha256=hashlib.sha256(data).hexdigest(), header=header,
                   bbox_min=[float(v[k].min()) for k in "xyz"], bbox_max=[float(v[k].max()) for k in "xyz"],
                   rgb_mean=[float(v[k].mean()) for k in ("red", "green", "blue")]),
              open(os.path.join(HERE, "points3d_meta.json"), "w"), indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
