"""Time 1 - ssim(render, gt) forward + backward: fused HIP kernels vs the reference's grouped-conv2d formulation."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd import loss_utils  # noqa: E402
from tests.torch_reference import ssim_torch  # noqa: E402


def main():
    for shape in ((1, 3, 512, 512), (1, 3, 1024, 1024)):
        gt = torch.rand(shape, device="cuda")
        res = {}
        for name, fn in (("fused", loss_utils.ssim), ("conv2d formulation", ssim_torch)):
            x = (gt + 0.1 * torch.randn(shape, device="cuda")).requires_grad_(True)

            def step():
                x.grad = None
                (1.0 - fn(x, gt)).backward()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 50 * 1e3
        print(f"ssim fwd+bwd {shape}: " + " | ".join(f"{k}: {v:.3f} ms" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
