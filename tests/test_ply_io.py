"""PLY I/O against the reference's own data file (check/points3d.ply, committed as tests/golden/points3d.ply) and round trips
of the Gaussian checkpoint format (attribute order of scene/gaussian_model.py:309-326)."""
import os

import numpy as np
import torch

from mygauhuman_amd import ply_io
from mygauhuman_amd.scene_model import HumanGaussianModel

import json  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "points3d_head256.ply")
META = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "points3d_meta.json")))


def test_reference_point_cloud_reads_and_rewrites_byte_for_byte(tmp_path):
    """tests/golden/points3d_head256.ply = the first 256 vertices of the reference's check/points3d.ply (written by plyfile),
    points3d_meta.json = header / size / bounding box of the whole file (tests/golden/make_golden.py)."""
    pts, cols, nrm = ply_io.fetch_ply(GOLD)
    assert pts.shape == (256, 3) and cols.shape == (256, 3) and nrm.shape == (256, 3)
    assert pts.dtype == np.float32 and 0.0 <= cols.min() and cols.max() <= 1.0 and np.isfinite(pts).all()
    assert np.all(pts >= np.array(META["bbox_min"]) - 1e-6) and np.all(pts <= np.array(META["bbox_max"]) + 1e-6)
    out = tmp_path / "again.ply"
    ply_io.store_ply(str(out), pts, np.round(cols * 255.0), nrm)
    assert out.read_bytes() == open(GOLD, "rb").read()                       # same header text, same records
    # the full file: 6890 vertices (one per SMPL vertex), header + 27 bytes per vertex, a human body in metres
    assert META["vertices"] == 6890 and META["bytes"] == len(META["header"]) + 27 * META["vertices"]
    hdr = open(GOLD, "rb").read().split(b"end_header\n")[0].decode() + "end_header\n"
    assert hdr == META["header"].replace("element vertex 6890", "element vertex 256")
    ext = np.array(META["bbox_max"]) - np.array(META["bbox_min"])
    assert 1.0 < ext.max() < 2.5


def test_gaussian_checkpoint_round_trip_and_layout(tmp_path):
    P, deg = 37, 3
    rng = np.random.default_rng(0)
    g = dict(means3D=rng.normal(0, 1, (P, 3)).astype(np.float32), scales=np.exp(rng.normal(-4, 0.3, (P, 3))).astype(np.float32),
             rotations=rng.normal(0, 1, (P, 4)).astype(np.float32), opacities=rng.uniform(0.05, 0.95, (P, 1)).astype(np.float32),
             shs=rng.normal(0, 1, (P, 16, 3)).astype(np.float32))
    m = HumanGaussianModel.from_arrays(g, deg, device="cpu")
    path = str(tmp_path / "sub" / "point_cloud.ply")
    ply_io.save_gaussians_ply(m, path)
    v = ply_io.read_ply(path)
    names = ply_io.gaussian_attribute_names(3, 45)
    assert list(v.dtype.names) == names and len(names) == 3 + 3 + 3 + 1 + 3 + 45 + 1 + 3 + 4 and len(v) == P
    assert os.path.getsize(path) == len(open(path, "rb").read().split(b"end_header\n")[0]) + len(b"end_header\n") + P * 4 * len(names)
    # SH coefficients are stored channel-major: f_rest_k = features_rest[:, k % 15, k // 15]
    np.testing.assert_array_equal(v["f_rest_16"], m._features_rest.detach().numpy()[:, 1, 1])
    np.testing.assert_array_equal(v["f_dc_2"], m._features_dc.detach().numpy()[:, 0, 2])
    m2 = ply_io.load_gaussians_ply(HumanGaussianModel(deg, device="cpu"), path)
    for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_normal", "_albedo", "_roughness"):
        assert torch.equal(getattr(m, a).detach(), getattr(m2, a).detach()), a
    # nn.Parameter(t.requires_grad_(False)) is trainable again: the reference's opacity quirk (:398) is reproduced
    assert m2._features_rest.shape == (P, 15, 3) and m2._opacity.requires_grad and m2._xyz.requires_grad


def test_ascii_and_error_paths(tmp_path):
    p = tmp_path / "a.ply"
    p.write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                 "end_header\n0.5 1 2 255\n-1 0 3.25 7\n")
    v = ply_io.read_ply(str(p))
    assert v["z"].tolist() == [2.0, 3.25] and v["red"].tolist() == [255, 7]
    bad = tmp_path / "b.ply"
    bad.write_text("not a ply")
    try:
        ply_io.read_ply(str(bad))
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
