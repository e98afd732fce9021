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
  * ROCm 7.2: under the HIP runtime's graph packet capture (the default) a memset NODE on memory of the graph's pool replays wrong
    as soon as other GPU work runs between two replays (profiles/r3_graph_bisect.txt).  libgsr records none; torch may.  Export
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the process touches the GPU (bench.py, the rank launcher and tests/conftest.py do).
    Whether that took effect cannot be known from Python, so a GraphedFrame VERIFIES ITSELF once after the capture: one replay,
    unrelated eager GPU work (the condition under which the bad mode fails), a second replay, both compared with the eager
    warm-up step; a mismatch raises -- naming the variable -- instead of handing wrong gradients to an optimizer.
"""
import os

import torch

from .diff_gaussian_rasterization import _C


def _tensors_of(result):
    """The floating-point tensors inside whatever step_fn returned (a tensor, a dict / list / tuple of them, or nothing)."""
    if isinstance(result, torch.Tensor):
        return [result] if result.is_floating_point() else []
    if isinstance(result, dict):
        result = list(result.values())
    if isinstance(result, (list, tuple)):
        return [t for t in result if isinstance(t, torch.Tensor) and t.is_floating_point()]
    return []


class GraphedFrame:
    def __init__(self, step_fn, warmup=3, zero_grads=None, debug_dump=None, verify=True, verify_rtol=2e-3):
        """step_fn is called `warmup` times eagerly on a side stream (allocator / caches settle), then once under capture.
        zero_grads: optional iterable of parameters whose .grad is set to None before the capture, so that the captured
        backward allocates them from the graph's pool (they stay valid between replays).
        verify: replay twice with unrelated eager GPU work in between and compare the gradients (and the tensors step_fn returns)
        with the eager warm-up step to verify_rtol of each tensor's largest magnitude (float atomics reorder sums; the failure
        this guards against is off by many orders of magnitude); raises RuntimeError on a mismatch.
        verify=True REQUIRES a deterministic step_fn without side effects: the constructor runs it `warmup` + 1 times and replays
        it twice, so per-call randomness (the reference's random_background) fails the comparison spuriously and anything the step
        accumulates in place (statistics, an optimizer step inside it) is applied that many extra times.  Such a step is captured
        with verify=False, which in turn demands DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment."""
        # Without the self-check there is nothing to stand on unless the runtime's graph packet capture is known to be off
        if not verify and os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0":
            raise RuntimeError("GraphedFrame(verify=False): DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in the environment before the HIP "
                               "runtime starts: on ROCm 7.2 a graph that contains a memset node on graph-pool memory (torch records "
                               "them) replays wrong otherwise, and nothing would notice")
        self.step_fn = step_fn
        params = list(zero_grads) if zero_grads is not None else []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        eager_result = None
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                for p in params:
                    p.grad = None
                eager_result = step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _C.AsyncCapacity.check_all()
        # what the eager step produced: the reference of the self-check below
        eager_ref = [None if p.grad is None else p.grad.detach().clone() for p in params]
        eager_ref += [t.detach().clone() for t in _tensors_of(eager_result)]
        del eager_result
        for p in params:
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
        self._params = params
        self._grads = [p.grad for p in self._params]
        if verify:
            self._verify(eager_ref, verify_rtol)

    def _verify(self, eager_ref, rtol):
        """One replay, unrelated eager GPU work (a fill, a reduction with a device-to-host read, a fresh allocation: what an
        optimizer step or a logging call does between frames), a second replay; both must reproduce the eager step."""
        live = [g for g in self._grads] + _tensors_of(self.result)
        if len(live) != len(eager_ref):
            return  # step_fn returns something else per call: nothing comparable

        def compare(tag):
            torch.cuda.synchronize()
            for k, (got, want) in enumerate(zip(live, eager_ref)):
                if got is None or want is None or got.shape != want.shape:
                    continue
                if want.numel() == 0:
                    continue
                scale = float(want.detach().abs().max())
                err = float((got.detach().float() - want.float()).abs().max())
                if not (err <= rtol * scale + 1e-30) or not bool(torch.isfinite(got.detach()).all()):
                    raise RuntimeError(
                        f"GraphedFrame self-check failed ({tag}, tensor {k}: max error {err:.3e} against a magnitude of {scale:.3e}): "
                        "the replay differs from the eager step.  Possible causes: (a) step_fn is not deterministic or has side "
                        "effects -- per-call randomness such as a random background, parameters or statistics updated in place "
                        "inside the step (verify=True replays the step twice more than you asked for: pass verify=False for such "
                        "a step); (b) on ROCm 7.2, the HIP runtime's graph packet capture replays memset nodes on graph-pool "
                        "memory wrongly -- DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in the environment BEFORE the first HIP call "
                        "of the process (torch.cuda.is_available() counts).")
        self.replay()
        compare("first replay")
        dev = live[0].device if live else torch.device("cuda")
        junk = torch.empty(1 << 22, device=dev)
        junk.fill_(2.0)
        float(junk.sum())
        junk2 = torch.empty(1 << 24, device=dev)
        junk2.fill_(1.0)
        del junk, junk2
        self.replay()
        compare("replay after unrelated eager GPU work")

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
