"""Time of the SH-gradient reconstruction of the view-parallel step at N views (GPU box):  python tools/sh_reconstruct_bench.py [views]
The kernel runs after the exchange of every N > 1 step (parallel.CompactShExchange.reconstruct); here with random view blocks."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd import _lib  # noqa: E402

lib, check = _lib.lib, _lib.check


def main(P=200_000, M=16):
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for n_views in ([int(a) for a in sys.argv[1:]] or [2, 4, 8]):
        means = torch.randn(P, 3, device=dev)
        # static Gaussians: [P*3 packed dL_dRGB | campos]
        stride = (P * 3 + 3 + 63) // 64 * 64
        views = torch.randn(n_views, stride, device=dev)
        grad = torch.empty(P, M, 3, device=dev)

        def static():
            check(lib.gsr_sh_grad_from_views(P, 3, M, n_views, means.data_ptr(), views.data_ptr(), stride, 1.0 / n_views, None,
                                             grad.data_ptr(), stream), "static")
        # articulated: [P*3 dRGB | P*3 posed positions | campos | P radii]
        stride_p = (7 * P + 4 + 63) // 64 * 64
        views_p = torch.randn(n_views, stride_p, device=dev)
        g_dc, g_rest = torch.empty(P, 1, 3, device=dev), torch.empty(P, 15, 3, device=dev)

        def posed():
            check(lib.gsr_sh_grad_from_views_posed(P, 3, n_views, views_p.data_ptr(), stride_p, 3 * P, 6 * P, 1.0 / n_views, None,
                                                   g_dc.data_ptr(), g_rest.data_ptr(), stream), "posed")
        for name, fn in (("static", static), ("posed", posed)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            print(f"{n_views} views, {name}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per reconstruction (P = {P})", flush=True)


if __name__ == "__main__":
    main()
