"""render() forward + loss + backward as ONE hipGraph (torch.cuda.graph whole-step capture): the per-frame host work drops from
~110 eager launches to one graph launch, so the frame time is the GPU time.  Per-frame inputs (camera matrices, SMPL pose,
targets) live in static device tensors that are updated in place before a replay; image size and field of view are by-value
kernel arguments, i.e. one graph per camera intrinsics; a densification step changes the shapes and needs a new capture.
The overflow / prefilter flags of captured forwards are read after the replays (AsyncCapacity.check_graph_status()).

  python tools/render_graph_bench.py        (GPU box)
"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd.diff_gaussian_rasterization import _C  # noqa: E402
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402
from tools.render_bench import PHASE1_KEYS  # noqa: E402
from tools.train_demo import build  # noqa: E402


def main(P=200_000, V=6890, W=1024, H=1024):
    model, cam, _ = build(P, V, W, H)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    bg = torch.zeros(3, device="cuda")
    params = list(model.parameters())

    def step():
        o = render(1, cam, model, pipe, bg)
        loss = sum(o[k].mean() for k in PHASE1_KEYS)
        loss.backward()
        return o

    def timed(fn, n=50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def eager():
        for p in params:
            p.grad = None
        step()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on a side stream, as whole-network capture asks
        for _ in range(30):
            eager()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    _C.AsyncCapacity.check_all()
    eager_ms = [timed(eager) for _ in range(3)]
    eager()
    torch.cuda.synchronize()
    ref_img = render(1, cam, model, pipe, bg)["render"].detach().clone()
    ref_grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
    print(f"eager: {eager_ms} ms/frame; capturing ...", flush=True)

    for p in params:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    print("captured; first replay ...", flush=True)
    g.replay()
    torch.cuda.synchronize()
    _C.AsyncCapacity.check_graph_status()
    img_err = float((out["render"].detach() - ref_img).abs().max())
    gerrs = []
    for p, r in zip(params, ref_grads):
        if r is None:
            continue
        gerrs.append(float((p.grad - r).abs().max()) / (float(r.abs().max()) + 1e-20))
    graph_ms = [timed(g.replay) for _ in range(3)]
    _C.AsyncCapacity.check_graph_status()
    print(f"render() fwd+bwd as one graph, P={P}, {W}x{H}: eager {min(eager_ms):.2f} ms/frame -> graph replay {min(graph_ms):.2f} ms/frame "
          f"(all runs: eager {eager_ms}, graph {graph_ms}); image max abs diff {img_err:.2e}, "
          f"gradient max rel diff {max(gerrs):.2e}", flush=True)


if __name__ == "__main__":
    main()
