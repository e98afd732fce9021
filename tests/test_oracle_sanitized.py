"""the CPU oracle once under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU pool has no sanitizer
runs, so the checker itself is what gets sanitized).  A child process preloads libasan, loads oracle/libgsr_oracle_san.so
through GSR_ORACLE_LIB and replays a small forward / backward / binning / LBS / k-NN scene; any report aborts the child."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import numpy as np
from oracle import oracle as orc
from tests import util
orc.set_threads(2)
for (P, W, H, seed, deg, scale, behind) in [(300, 80, 48, 1, 3, 0.05, 0.1), (40, 16, 16, 2, 0, 0.3, 0.0), (0, 32, 32, 3, 1, 0.02, 0.0)]:
    cam, g = util.make_scene(P, W, H, seed, deg, scale, behind)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    for mode in ("sh", "precomp"):
        ref = util.oracle_forward(orc, cam, g, bg, mode)
        rng = np.random.default_rng(seed)
        dc, dd, da = (rng.normal(0, 1, s).astype(np.float32) for s in ((3, H, W), (1, H, W), (1, H, W)))
        if P:
            orc.rasterize_backward(ref, dc, dd, da)
pts = np.random.default_rng(0).normal(0, 1, (500, 3)).astype(np.float32)
orc.dist2_brute(pts)
orc.dist2_morton(pts)
orc.dist2_morton(pts[:2])
orc.knn_self(pts, 3)
verts = np.random.default_rng(1).normal(0, 1, (64, 3)).astype(np.float32)
orc.nearest_vertex(pts, verts)
orc.rodrigues(np.array([0.1, -0.2, 0.3], np.float32))
import bench
m = bench.synthetic_smpl(V=200)
rng = np.random.default_rng(4)
v, _, _, _ = orc.smpl_lbs(rng.normal(0, 1, 10).astype(np.float32), rng.normal(0, 0.2, 72).astype(np.float32), m["v_template"],
                          m["shapedirs"], m["posedirs"], m["J_regressor"], m["parents"], m["weights"])
orc.project(v, np.eye(4, dtype=np.float32))
print("sanitized oracle run OK")
"""


def test_oracle_under_asan_ubsan(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "san"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               GSR_ORACLE_LIB=os.path.join(ROOT, "oracle", "libgsr_oracle_san.so"), PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "sanitized oracle run OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
