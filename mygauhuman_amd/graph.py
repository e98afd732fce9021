"""Whole-step hipGraph capture of a frame (forward + loss + backward) over the sync-free rasterizer.

The reference's render() is ~110 eager launches per training frame and therefore host-bound once the kernels are fast
(gaussian_renderer/__init__.py:53-295).  Every kernel of libgsr.so is capture-safe (no host reads, no allocations, options
resolved per stream), so the whole step can be recorded once and replayed with ONE launch per frame:

    frame = GraphedFrame(step_fn)       # step_fn(): render(...), loss, loss.backward(); returns anything (kept as .result)
    ...copy this frame's camera / pose / targets into the static input tensors...
    frame.replay()                      # .result and the parameters' .grad now hold this frame's values
    frame.check()                       # every so often: raises if a captured forward overflowed its binning capacity

Rules of a captured region (violations bake a stale pointer or an unqueryable object into the graph):
  * per-frame inputs are device tensors updated IN PLACE (camera matrices, SMPL pose, ground truth); image size and field of
    view are by-value kernel arguments: one GraphedFrame per camera intrinsics;
  * no pinned device-to-host copies / event records: rasterize_gaussians_async notices the capture and leaves its deferred
    overflow watch out (AsyncCapacity.graph_status keeps the status tensors for check());
  * the number of Gaussians is baked in: capture again after a densification / pruning step;
  * stage profiling (gsr_profile_enable) must be off;
  * ROCm 7.2: the HIP runtime's graph packet capture must be off (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, which importing this
    package sets if the HIP runtime is not up yet): with it, replays go wrong as soon as other GPU work runs between them.
"""
import torch

from . import GRAPH_REPLAY_SAFE
from .diff_gaussian_rasterization import _C


class GraphedFrame:
    def __init__(self, step_fn, warmup=3, zero_grads=None, debug_dump=None):
        """step_fn is called `warmup` times eagerly on a side stream (allocator / caches settle), then once under capture.
        zero_grads: optional iterable of parameters whose .grad is set to None before the capture, so that the captured
        backward allocates them from the graph's pool (they stay valid between replays)."""
        if not GRAPH_REPLAY_SAFE:
            raise RuntimeError("GraphedFrame: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in the environment before the HIP runtime "
                               "starts (import mygauhuman_amd before the first torch.cuda call, or export it): on ROCm 7.2 graph "
                               "replays over torch allocations return wrong results otherwise")
        self.step_fn = step_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if zero_grads is not None:
                    for p in zero_grads:
                        p.grad = None
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _C.AsyncCapacity.check_all()
        if zero_grads is not None:
            for p in zero_grads:
                p.grad = None
        n0 = len(_C.AsyncCapacity.graph_status)
        self.graph = torch.cuda.CUDAGraph()
        if debug_dump:
            self.graph.enable_debug_mode()
        # capture on the SAME stream the warm-up ran on: autograd runs a leaf's gradient accumulation on the stream its
        # AccumulateGrad node was created on, and a node that survives from the warm-up would otherwise fork the captured
        # backward onto a second stream (legal, but the graph pool's block reuse is only ordered along one stream)
        with torch.cuda.graph(self.graph, stream=side):
            self.result = step_fn()
        if debug_dump:
            self.graph.debug_dump(debug_dump)
        self._status = _C.AsyncCapacity.graph_status[n0:]
        del _C.AsyncCapacity.graph_status[n0:]
        # the gradients the captured backward writes: tensors of the graph's pool, re-attached at every replay
        self._params = list(zero_grads) if zero_grads is not None else []
        self._grads = [p.grad for p in self._params]

    def replay(self):
        self.graph.replay()
        for p, g in zip(self._params, self._grads):
            p.grad = g
        return self.result

    def check(self):
        """Synchronises; raises RuntimeError if the last replay overflowed a captured forward's binning capacity."""
        saved, _C.AsyncCapacity.graph_status = _C.AsyncCapacity.graph_status, list(self._status)
        try:
            _C.AsyncCapacity.check_graph_status()
        finally:
            _C.AsyncCapacity.graph_status = saved
