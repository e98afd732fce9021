"""Stage times of the render() frame under a tuning set (GPU box):  python tools/render_stage_ab.py key=value ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mygauhuman_amd import _lib  # noqa: E402

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_tuning(k, int(v))
from mygauhuman_amd.diff_gaussian_rasterization import _C  # noqa: E402
_cap = {}
_real = _C.rasterize_gaussians_async


def _spy(*a, **k):
    out = _real(*a, **k)
    if not _cap:  # (the first, eager call only: tensors of the warm-up stream must not be kept alive across the graph capture)
        _cap["o"], _cap["P"], _cap["HW"] = out, a[1].shape[0], (int(a[12]), int(a[13]))
    return out


_C.rasterize_gaussians_async = _spy
_lib.profile_enable(_lib.PROF_STAGES)
r = bench.render_extra(torch.device("cuda", 0), steps=40, warmup=20)
torch.cuda.synchronize()
prof = _lib.profile_read()
if _cap:
    o, P, (H, W) = _cap["o"], _cap["P"], _cap["HW"]
    order = _C.query_state("ORDER", P, o[0], W, H, o[5], o[6], o[7]).cpu().numpy().view("uint32")
    rng = _C.query_state("RANGES", P, o[0], W, H, o[5], o[6], o[7]).cpu().numpy().astype("int64")
    L = rng[:, 1] - rng[:, 0]
    ent = order[2:2 + int(order[1])]
    nseg = ((ent >> 25) & 7) + 1
    print(f"lists: busy {int((L > 0).sum())} mean {L[L > 0].mean():.0f} max {L.max()}; slots {int(order[1])}; tiles cut into 2/3/4: "
          f"{[int(((nseg == k) & (((ent >> 22) & 7) == 0)).sum()) for k in (2, 3, 4)]}", flush=True)
print(" ".join(sys.argv[1:]) or "(defaults)", {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in prof.items()}, "eager",
      r["eager"]["ms_per_step"], "graph", r.get("one_graph", {}).get("ms_per_step"), flush=True)
