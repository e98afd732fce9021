"""Time prune_points / densify_and_prune: one row plan + one fused gather (mygauhuman_amd.densify) against the reference's
structure (boolean-mask indexing and torch.cat per tensor, scene/gaussian_model.py:421-487) written with plain torch ops."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd import densify  # noqa: E402
from tests.test_densify_cpu import make_state  # noqa: E402
from tests.test_gpu_densify import to_model  # noqa: E402


def reference_structure_prune(m, mask):
    valid = ~mask
    for group in m.optimizer.param_groups:
        p = group["params"][0]
        st = m.optimizer.state.get(p, None)
        if st is not None:
            st["exp_avg"] = st["exp_avg"][valid]
            st["exp_avg_sq"] = st["exp_avg_sq"][valid]
            del m.optimizer.state[p]
            group["params"][0] = torch.nn.Parameter(p[valid].requires_grad_(True))
            m.optimizer.state[group["params"][0]] = st
        else:
            group["params"][0] = torch.nn.Parameter(p[valid].requires_grad_(True))
        setattr(m, densify.ATTR[group["name"]], group["params"][0])
    m.xyz_gradient_accum = m.xyz_gradient_accum[valid]
    m.denom = m.denom[valid]
    m.max_radii2D = m.max_radii2D[valid]


def main():
    for P in (200_000, 500_000):
        st = make_state(P, 0)
        mask = torch.from_numpy(np.random.default_rng(1).uniform(0, 1, P) < 0.1).cuda()
        res = {}
        for name, fn in (("fused row plan", densify.prune_points), ("reference structure (torch ops)", reference_structure_prune)):
            ts = []
            for _ in range(6):
                m = to_model(st)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn(m, mask)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            res[name] = min(ts)
        print(f"prune_points, P={P}, 10% pruned: " + " | ".join(f"{k}: {v:.2f} ms" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
