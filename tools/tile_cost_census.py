"""Would a cost-accurate visiting order help the render() frame's blend kernels?  (GPU box.)  Renders the tools/render_bench.py body
scene once through the blocking rasterizer, reads the per-tile list ranges and the per-pixel n_contrib back (gsr_query_state), and
compares, in a greedy list-scheduling model of W workgroup slots, the makespan of the CURRENT order (descending list length) with the
order by WALKED length (max n_contrib of the tile = what the backward really walks; the forward walks a little further)."""
import heapq
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mygauhuman_amd.diff_gaussian_rasterization import _C  # noqa: E402
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402


def makespan(costs, slots):
    h = [0.0] * slots
    for c in costs:
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)


def main():
    import render_bench
    captured = {}
    real = _C.rasterize_gaussians

    def spy(*a, **k):
        out = real(*a, **k)
        captured["out"], captured["P"], captured["HW"] = out, a[1].shape[0], (int(a[12]), int(a[13]))
        return out
    _C.rasterize_gaussians = spy
    real_async = _C.rasterize_gaussians_async

    def spy_async(*a, **k):
        out = real_async(*a, **k)
        captured["out"], captured["P"], captured["HW"] = out, a[1].shape[0], (int(a[12]), int(a[13]))
        return out
    _C.rasterize_gaussians_async = spy_async
    import mygauhuman_amd.diff_gaussian_rasterization as dgr
    if hasattr(dgr, "_C"):
        dgr._C.rasterize_gaussians = spy
    model, cam, bg = render_bench.scene()
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, separate_feature_passes=False,
                                 sync_free_raster=os.environ.get("SYNC_FREE", "0") == "1")
    with torch.no_grad():
        render(1, cam, model, pipe, bg)
    o, P, (H, W) = captured["out"], captured["P"], captured["HW"]
    rng = _C.query_state("RANGES", P, o[0], W, H, o[5], o[6], o[7]).cpu().numpy().astype(np.int64)
    ncon = _C.query_state("N_CONTRIB", P, o[0], W, H, o[5], o[6], o[7]).cpu().numpy().astype(np.int64)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    order = _C.query_state("ORDER", P, o[0], W, H, o[5], o[6], o[7]).cpu().numpy().view(np.uint32)
    ent = order[2:2 + int(order[1])]
    nseg = ((ent >> 25) & 7) + 1
    print(f"order: mode {int(order[0] & 0xFF)}, longest {int(order[0] >> 8)}, slots {int(order[1])} for {gx * gy} tiles; tiles cut into 2/3/4 segments: "
          f"{[int(((nseg == k) & (((ent >> 22) & 7) == 0)).sum()) for k in (2, 3, 4)]}")
    L = (rng[:, 1] - rng[:, 0]).reshape(gy, gx)
    pad = np.zeros((gy * 16, gx * 16), np.int64)
    pad[:H, :W] = ncon.reshape(H, W)
    walked_q = pad.reshape(gy, 2, 8, gx, 2, 8).max(axis=(2, 5))            # [gy, 2, gx, 2]: per 8x8 quadrant
    walked = walked_q.max(axis=(1, 3))
    busy = L > 0
    print(f"tiles {gx * gy}, busy {int(busy.sum())}; list mean {L[busy].mean():.0f} max {L.max()}; walked mean {walked[busy].mean():.0f} "
          f"max {walked.max()}; walked/list mean {np.mean(walked[busy] / L[busy]):.2f}; corr {np.corrcoef(L[busy], walked[busy])[0, 1]:.3f}")
    print(f"quadrant walked: mean {walked_q[walked_q > 0].mean():.0f}; sum over quadrants {int(walked_q.sum())} vs 4 x list {4 * int(L.sum())}")
    Lb, Wb = L[busy].astype(float), walked[busy].astype(float)
    for slots in (1280, 1024, 768):
        cur = makespan(Wb[np.argsort(-Lb, kind="stable")], slots)
        best = makespan(np.sort(Wb)[::-1], slots)
        rnd = makespan(Wb, slots)
        print(f"slots {slots}: makespan (entries walked by the busiest slot) raster order {rnd:.0f} | by list length {cur:.0f} | by walked "
              f"{best:.0f} | bounds: mean {Wb.sum() / slots:.0f}, longest {Wb.max():.0f}")


if __name__ == "__main__":
    main()
