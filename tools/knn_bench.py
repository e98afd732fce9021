"""Time the k-NN service: distCUDA2, self k-NN with indices, nearest SMPL vertex (grid vs brute force inside the LBS kernel)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd import knn_cuda  # noqa: E402
from mygauhuman_amd.simple_knn._C import distCUDA2  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    rng = np.random.default_rng(0)
    V = 6890
    verts = (rng.uniform(-1, 1, (V, 3)) * np.array([0.45, 0.9, 0.15])).astype(np.float32)
    for P in (6890, 200_000, 500_000):
        pts = (verts[rng.integers(0, V, P)] + rng.normal(0, 0.01, (P, 3))).astype(np.float32)
        x, v = torch.from_numpy(pts).cuda(), torch.from_numpy(verts).cuda()
        print(f"P={P}: distCUDA2 {timeit(lambda: distCUDA2(x)):.3f} ms | knn_self k=2 {timeit(lambda: knn_cuda.knn_self(x, 2)):.3f} ms | "
              f"k=3 {timeit(lambda: knn_cuda.knn_self(x, 3)):.3f} ms | nearest of {V} vertices {timeit(lambda: knn_cuda.knn_nearest(v, x)):.3f} ms",
              flush=True)


if __name__ == "__main__":
    main()
