"""The skinning-weight offset network at 200k points: fused forward (csrc/mlp.hip) vs the same module in torch ops, no_grad.
python tools/mlp_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd.nets import FusedLBSOffsetDecoder  # noqa: E402

dec = FusedLBSOffsetDecoder().cuda()
for P in (200_000,):
    pts = torch.rand(1, P, 3, device="cuda") - 0.5
    with torch.no_grad():
        from mygauhuman_amd._lib import lib, ptr

        def bf16x3():
            x = pts[0].contiguous()
            out = torch.empty((P, 24), device="cuda")
            lib.gsr_debug_lbs_offset_mlp_forward_bf16x3(P, ptr(x), ptr(dec._packed_weights(x.device)), ptr(out), torch.cuda.current_stream().cuda_stream)
            return out.t()[None]
        ref = dec.forward_torch(pts.double()) if False else None
        d64 = FusedLBSOffsetDecoder().cuda().double()
        d64.load_state_dict({k: v.double() for k, v in dec.state_dict().items()})
        want = d64.forward_torch(pts.double())
        for name, fn in (("fused", lambda: dec(pts)), ("bf16x3", bf16x3), ("torch ops", lambda: dec.forward_torch(pts))):
            print(f"P={P:7d} {name:10s} max error vs float64 / max |out| = {float((fn().double() - want).abs().max() / want.abs().max()):.2e}", flush=True)
        for name, fn in (("fused", lambda: dec(pts)), ("bf16x3", bf16x3), ("torch ops", lambda: dec.forward_torch(pts))):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"P={P:7d} {name:10s} {dt * 1e3:8.3f} ms   {137e3 * P / dt / 1e12:6.1f} TFLOP/s", flush=True)
    w = torch.randn(1, 24, P, device="cuda")
    for name, fn in (("fused", lambda: dec(pts)), ("torch ops", lambda: dec.forward_torch(pts))):
        def step():
            for p in dec.parameters():
                p.grad = None
            (fn() * w).sum().backward()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        print(f"P={P:7d} {name:10s} forward + backward {(time.perf_counter() - t0) / n * 1e3:8.3f} ms", flush=True)
