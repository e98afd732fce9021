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
_lib.profile_enable(_lib.PROF_STAGES)
r = bench.render_extra(torch.device("cuda", 0), steps=40, warmup=20)
torch.cuda.synchronize()
prof = _lib.profile_read()
print(" ".join(sys.argv[1:]) or "(defaults)", {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in prof.items()}, "eager",
      r["eager"]["ms_per_step"], "graph", r.get("one_graph", {}).get("ms_per_step"), flush=True)
