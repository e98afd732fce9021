"""Is the attribute kernel pair (csrc/attributes.hip) quantised by resident workgroups?  Times forward and backward over P around
768 x 256 = 196,608 (three 53 KB workgroups per CU x 256 CUs).   python tools/attr_quant_probe.py

Measured (round 4, buffers cache-hot): forward 18.7 us at P = 196,608 (768 workgroups, one round), 24.3 at 200,000 (782: a second
round of 14); backward 26.2 / 34.3.  Tried against it, all slower or equal at 200k: 64-thread workgroups (23.7: the LDS granule
then admits 11, not 12, per CU), SH rows staged half a row at a time to fit five workgroups per CU -- second half re-read (26.8)
or held in registers (25.1; 22.1 vs 16.8 at 150k: the extra barrier pair and element-wise LDS writes cost more than the second
round), the direct (non-staged) instantiation (37.8 / 63.0).  The backward is held to 768 threads per CU by its 168 VGPRs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd.attributes import frame_attributes  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for P in (150_000, 196_608, 200_000, 230_000, 262_144, 300_000, 500_000):
    r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    means, T, wn = r(P, 3).requires_grad_(), r(P, 3, 3).requires_grad_(), r(P, 3).requires_grad_()
    scales, rc, ra = r(P, 3).abs().requires_grad_(), r(P, 4).requires_grad_(), r(P, 4).requires_grad_()
    alb, occ = r(P, 3).requires_grad_(), r(P, 3).requires_grad_()
    dc, rest = r(P, 1, 3).requires_grad_(), r(P, 15, 3).requires_grad_()
    campos, view = r(3), torch.eye(4, device=dev)

    from mygauhuman_amd._lib import lib, ptr
    c = lambda t: t.detach().contiguous()  # noqa: E731
    ins = [c(means), c(T).reshape(P, 9), c(wn), c(scales), c(rc), c(ra), c(alb), c(alb), c(occ), c(dc), c(rest), campos, view.reshape(16)]
    cov, col, feat = torch.empty(P, 6, device=dev), torch.empty(P, 3, device=dev), torch.empty(P, 18, device=dev)
    gcov, gcol, gfeat = torch.randn_like(cov), torch.randn_like(col), torch.randn_like(feat)
    d = [torch.empty_like(t) for t in (ins[0], ins[1], ins[2], ins[3], ins[4], ins[5], ins[6], ins[6], ins[8], ins[9], ins[10])]
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        lib.gsr_frame_attributes_forward_split(P, 3, 16, ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), ptr(ins[3]), 1.0, ptr(ins[4]), ptr(ins[5]),
                                               ptr(ins[6]), ptr(ins[7]), ptr(ins[8]), ptr(ins[9]), ptr(ins[10]), ptr(ins[11]), ptr(ins[12]),
                                               ptr(cov), ptr(col), ptr(feat), st)

    def bwd():
        lib.gsr_frame_attributes_backward_split(P, 3, 16, ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), ptr(ins[3]), 1.0, ptr(ins[4]), ptr(ins[5]),
                                                ptr(ins[6]), ptr(ins[7]), ptr(ins[8]), ptr(ins[9]), ptr(ins[10]), ptr(ins[11]), ptr(ins[12]),
                                                ptr(gcov), ptr(gcol), ptr(gfeat), *[ptr(t) for t in d], st)
    big = torch.randn(P * 48 + 1, device=dev, generator=g)
    sh_un = big[1:]          # 4-byte aligned only: the library takes the non-staged instantiation
    d_un = torch.empty(P * 48 + 1, device=dev)[1:]

    def fwd_direct():
        lib.gsr_frame_attributes_forward_split(P, 3, 16, ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), ptr(ins[3]), 1.0, ptr(ins[4]), ptr(ins[5]),
                                               ptr(ins[6]), ptr(ins[7]), ptr(ins[8]), sh_un.data_ptr(), None, ptr(ins[11]), ptr(ins[12]),
                                               ptr(cov), ptr(col), ptr(feat), st)

    def bwd_direct():
        lib.gsr_frame_attributes_backward_split(P, 3, 16, ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), ptr(ins[3]), 1.0, ptr(ins[4]), ptr(ins[5]),
                                                ptr(ins[6]), ptr(ins[7]), ptr(ins[8]), sh_un.data_ptr(), None, ptr(ins[11]), ptr(ins[12]),
                                                ptr(gcov), ptr(gcol), ptr(gfeat), *[ptr(t) for t in d[:9]], d_un.data_ptr(), None, st)
    tf, tb, tfd, tbd = [], [], [], []
    for fn, acc in ((fwd, tf), (bwd, tb), (fwd_direct, tfd), (bwd_direct, tbd)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            acc.append(e0.elapsed_time(e1) / 50 * 1e3)
    tf.sort(), tb.sort(), tfd.sort(), tbd.sort()
    print(f"P={P:7d}  workgroups={(P + 255) // 256:5d}  forward {tf[len(tf) // 2]:7.1f} us  backward {tb[len(tb) // 2]:7.1f} us   not staged: {tfd[len(tfd) // 2]:7.1f} / {tbd[len(tbd) // 2]:7.1f} us", flush=True)
