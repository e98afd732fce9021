"""Forward-only render() (the render.py use case): ms per frame under torch.no_grad()."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402
from tools.train_demo import build  # noqa: E402


def main():
    model, cam, _ = build(200_000, 6890, 1024, 1024)
    bg = torch.zeros(3, device="cuda")
    for sync_free in (False, True):
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, sync_free_raster=sync_free)
        with torch.no_grad():
            for _ in range(10):
                render(1, cam, model, pipe, bg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                render(1, cam, model, pipe, bg)
            torch.cuda.synchronize()
        print(f"render() forward only, 200k Gaussians, 1024x1024, sync_free_raster={sync_free}: "
              f"{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/frame", flush=True)


if __name__ == "__main__":
    main()
