#!/usr/bin/env python3
"""How long does a freshly acquired MI355X take to reach its sustained shader clock, and what does a bench step cost on the way?
Prints (ms since the first GPU work, GHz) from back-to-back clock probes (gsr_debug_clock_probe), then the per-step times of C3
steps run right after a pause.  Measurement only (round 4: why the driver's 25-step run read 10 % below a 300-step run)."""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mygauhuman_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.zeros(1, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
rows = []
for i in range(int(os.environ.get("PROBES", "400"))):
    ghz, mt = _lib.clock_probe(1024, 1 << 19, dev)
    rows.append(((time.perf_counter() - t0) * 1e3, ghz, mt))
for k, (ms, ghz, mt) in enumerate(rows):
    if k < 20 or k % 20 == 0:
        print(f"probe {k:4d}  t={ms:8.1f} ms  clock_fma={ghz:.4f} GHz  s_memtime_per_ns={mt}")
print("settle_clock:", _lib.settle_clock(dev)[-6:])
# idle, then again: does the clock fall back?
for pause in (0.05, 0.5, 2.0):
    time.sleep(pause)
    t1 = time.perf_counter()
    r = [(round((time.perf_counter() - t1) * 1e3, 1), round(_lib.clock_probe(1024, 1 << 19, dev)[0], 4)) for _ in range(12)]
    print(f"after {pause} s idle:", r)
