"""View-parallel multi-GPU layer (new work: the reference is single-GPU, utils/general_utils.py:140).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests).  Every
rank holds a full replica of the Gaussian parameters and renders its own camera view.  Per step the ranks exchange
  * a SUM all-reduce of ONE flat fp32 bucket with the per-Gaussian gradients of means3D 3 + opacity 1 + scales 3 +
    rotations 4 floats (44 B per Gaussian), divided by the world size afterwards -- the rasterizer backward writes straight
    into views of that bucket, no gather / flatten copy;
  * the SH-coefficient gradient (3M floats, 192 B at M = 16 -- 81 % of a naive payload) in COMPACT form: it is rank one per
    view, dL_dsh[k][c] = w_k(view direction) * dL_dRGB[c], so every rank all-gathers its clamp-masked dL_dRGB (12 B per
    Gaussian) plus its camera position and rebuilds  mean_v w_k(dir_v) dL_dRGB_v  locally (csrc/sh_exchange.hip; fixed
    summation order, replicas stay bit-identical).  Bytes on the xGMI links per rank at 8 GPUs and 200k Gaussians:
    ~82 MB for a plain 47 MB all-reduce -> ~32 MB.  `compact_sh=False` selects the plain all-reduce of everything.
Densification statistics use a second, tiny bucket (sum of gradient norms + visibility counts, max of radii:
scene/gaussian_model.py:764-766, train.py:403).
"""
import os
from collections import OrderedDict, deque

import torch
import torch.distributed as dist

from ._lib import check, lib
from .launch import free_port, spawn_ranks  # noqa: F401  (re-exported)


def _staged(t, group=None):
    """gloo rehearsals with device tensors: the collective runs on a host copy (ProcessGroupGloo builds without device
    support reject device tensors); RCCL takes the device tensor itself."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_(t, op=dist.ReduceOp.SUM, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def init_distributed(device_type="cuda"):
    """Initialise the default process group from the torchrun environment. Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # rehearsal knobs (single-GPU boxes): GSR_DIST_BACKEND=gloo, GSR_SINGLE_DEVICE=1 puts every rank on cuda:0
        backend = os.environ.get("GSR_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if os.environ.get("GSR_SINGLE_DEVICE") == "1":
            local = 0
        if device_type == "cuda":
            torch.cuda.set_device(local)
        if device_type == "cuda" and backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradientBucket:
    """One flat fp32 buffer with named, shaped views (the all-reduce payload)."""

    def __init__(self, shapes, device):
        self.slices = OrderedDict()
        off = 0
        for name, shape in shapes.items():
            n = 1
            for s in shape:
                n *= int(s)
            self.slices[name] = (off, n, tuple(int(s) for s in shape))
            off += (n + 63) // 64 * 64  # 256-byte aligned views (the HIP kernels use 16-byte vector stores)
        self.flat = torch.zeros(max(off, 1), dtype=torch.float32, device=device)
        self.views = {k: self.flat[o:o + n].view(shape) for k, (o, n, shape) in self.slices.items()}

    def __getitem__(self, name):
        return self.views[name]

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def all_reduce_mean(self, group=None):
        """SUM all-reduce + divide by world size (no-op for a single process)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            all_reduce_(self.flat, dist.ReduceOp.SUM, group)
            self.flat.div_(dist.get_world_size(group))
        return self.flat


def gaussian_gradient_shapes(P, M, mode="sh"):
    if mode == "sh":
        return OrderedDict(means3D=(P, 3), sh=(P, M, 3), opacity=(P, 1), scales=(P, 3), rotations=(P, 4))
    if mode == "sh_compact":  # the SH gradient travels as masked dL_dRGB through an all-gather instead
        return OrderedDict(means3D=(P, 3), opacity=(P, 1), scales=(P, 3), rotations=(P, 4))
    return OrderedDict(means3D=(P, 3), colors=(P, 3), opacity=(P, 1), cov3D=(P, 6))


class CompactShExchange:
    """All-gather of (masked dL_dRGB [P,3] | campos [3]) per rank + local reconstruction of the mean SH gradient."""

    def __init__(self, P, M, device, group=None):
        self.P, self.M, self.group = int(P), int(M), group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.stride = (self.P * 3 + 3 + 63) // 64 * 64
        self.gathered = torch.zeros((self.world, self.stride), dtype=torch.float32, device=device)
        self.mine = self.gathered[dist.get_rank(group) if self.world > 1 else 0]
        self.grad = torch.empty((self.P, self.M, 3), dtype=torch.float32, device=device)

    def pack(self, session, campos):
        """Fill this rank's block from the session's last backward (its raw dL_dcolor and the forward's clamp bits)."""
        dev = self.grad.device
        check(lib.gsr_sh_view_pack(self.P, session.geom.data_ptr(), session.dL_dcolors.data_ptr(), self.mine.data_ptr(),
                                   torch.cuda.current_stream(dev).cuda_stream), "gsr_sh_view_pack")
        self.mine[self.P * 3:self.P * 3 + 3].copy_(campos.reshape(3))

    def exchange(self):
        if self.world <= 1:
            return
        if dist.get_backend() == "nccl":
            # in-place all-gather: RCCL accepts a send buffer that IS this rank's slot of the receive buffer
            # (sendbuff == recvbuff + rank * count), so no per-step copy of the 2.4 MB block is made
            dist.all_gather_into_tensor(self.gathered.view(-1), self.mine, group=self.group)
            return
        # gloo (CPU tests, one-GPU rehearsals): host-staged, list form
        src = self.mine.cpu() if self.mine.is_cuda else self.mine.clone()
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src, group=self.group)
        self.gathered.copy_(torch.stack(parts))

    def reconstruct(self, means3D, sh_degree, dev_scale=None):
        """grad[P,M,3] = mean over the views of w_k(dir_v) * dL_dRGB_v; dev_scale: optional device float that REPLACES the
        1 / world factor (0 for a step every replica must skip)."""
        dev = self.grad.device
        check(lib.gsr_sh_grad_from_views(self.P, int(sh_degree), self.M, self.world, means3D.data_ptr(), self.gathered.data_ptr(),
                                         self.stride, 1.0 if dev_scale is not None else 1.0 / self.world,
                                         None if dev_scale is None else dev_scale.data_ptr(), self.grad.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream), "gsr_sh_grad_from_views")
        return self.grad


def all_reduce_densify_stats(grad_norm_accum, denom, max_radii2D, group=None):
    """Sum the per-view densification statistics and max the screen radii across ranks, in place
    (add_densification_stats, scene/gaussian_model.py:764-766; max_radii2D update, train.py:403)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    P = grad_norm_accum.shape[0]
    packed = torch.cat([grad_norm_accum.reshape(P), denom.reshape(P).to(grad_norm_accum.dtype)])
    all_reduce_(packed, dist.ReduceOp.SUM, group)
    grad_norm_accum.copy_(packed[:P].view_as(grad_norm_accum))
    denom.copy_(packed[P:].view_as(denom).to(denom.dtype))
    all_reduce_(max_radii2D, dist.ReduceOp.MAX, group)


def view_for_step(step, rank, world):
    """Step s assigns view (s * world + rank) to this rank."""
    return step * world + rank


class BinningOverflow(RuntimeError):
    """A step needed more (Gaussian, tile) instances than its session's binning buffer holds: that view rendered only the
    background.  The step's gradients were zeroed on EVERY rank (replicas stay identical) and the capacity has been raised."""


class ViewParallelStep:
    """fwd + loss gradient + bwd of one camera view on this rank, gradients into the flat bucket, then the all-reduce.

    Loss = L1(color, gt) + 0.1 * MSE(alpha, mask) (the bench's "alpha-mask loss", train.py:261-262); its gradient is
    formed by one fused HIP kernel, no autograd graph is built.  Uses a sync-free RasterSession (fastpath.py): the
    host never waits for the GPU inside a step.

    Capacity overflow (the session was sized from one calibration view; cameras and Gaussians change) is handled on the
    device and reported late, never silently:
      * the session's overflow flag rides in one slot of the all-reduced bucket, so after the reduction every rank holds the
        number of ranks that overflowed; the gradients of such a step are multiplied by zero ON THE DEVICE on every rank
        (a consistent skipped step) instead of averaging a blank view into the mean;
      * (ranks overflowed, R, own flag) are written into pinned host memory by the same one-thread bookkeeping kernel that
        forms the scale (gsr_step_status) and are examined once the step is `max_in_flight` calls old (by then its event has
        long fired: the host does not stall) and by check(): an overflowed step raises BinningOverflow there -- at the same
        call on every rank, so the collectives stay matched -- after this rank's session has been regrown to 2 x its R."""

    def __init__(self, params, sh_degree, cam, bg, group=None, slack=1.3, compact_sh=None, max_in_flight=2, capacity=None):
        from .fastpath import RasterSession
        self.p = params  # dict: means3D, shs, opacities, scales, rotations (device tensors)
        self.deg = sh_degree
        self.group = group
        P, M = params["means3D"].shape[0], params["shs"].shape[1]
        dev = params["means3D"].device
        self.world = world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if compact_sh is None:
            compact_sh = world > 1 and os.environ.get("GSR_COMPACT_SH", "1") != "0"
        self.compact = CompactShExchange(P, M, dev, group) if compact_sh else None
        shapes = gaussian_gradient_shapes(P, M, "sh_compact" if compact_sh else "sh")
        shapes["overflow"] = (1,)  # number of ranks whose binning buffer overflowed in this step (SUM all-reduced)
        self.bucket = GradientBucket(shapes, dev)
        self.grads = {k: v for k, v in self.bucket.views.items() if k != "overflow"}
        if compact_sh:
            self.grads["sh"] = self.compact.grad  # the local backward writes this view's dL_dsh here; reconstruct() replaces it
        if capacity is None:
            self.session = RasterSession.calibrated(params, cam, bg, sh_degree, slack=slack)
        else:
            self.session = RasterSession(P, cam["W"], cam["H"], M, dev, capacity)
        self.max_in_flight = int(max_in_flight)
        self.steps = 0
        self.pending = deque()   # (step index, pinned uint32 [ranks overflowed, R, own flag], event)
        self._pinned, self._events = [], []
        self._scale = torch.zeros(1, dtype=torch.float32, device=dev)

    @property
    def payload_bytes(self):
        """Bytes this rank contributes to the collectives of one step."""
        return self.bucket.nbytes + (self.compact.stride * 4 if self.compact else 0)

    # ---- deferred overflow checks
    def _status(self, phase, report=None):
        dev = self._scale.device
        check(lib.gsr_step_status(phase, self.session.status.data_ptr(), self.bucket["overflow"].data_ptr(), 1.0 / self.world,
                                  self._scale.data_ptr(), None if report is None else report.data_ptr(),
                                  torch.cuda.current_stream(dev).cuda_stream), "gsr_step_status")

    def _report_buffer(self):
        # pinned host memory is mapped into the device's address space: the bookkeeping kernel writes the three words
        # straight into it (no copy engine involved), the event tells the host when they are there
        return self._pinned.pop() if self._pinned else torch.zeros(3, dtype=torch.int32).pin_memory()

    def _examine(self, block_older_than):
        while self.pending:
            step, host, ev = self.pending[0]
            if step > block_older_than:  # (never by ev.query(): ranks must raise at the same call)
                return
            ev.synchronize()
            self.pending.popleft()
            self._events.append(ev)
            if self.world > 1:   # [ranks that overflowed, R, own flag], written by gsr_step_status after the all-reduce
                ranks, R, own = int(host[0]), int(host[1]) & 0xFFFFFFFF, int(host[2]) != 0
            else:                # the forward's own status words [R, flags], written straight into this pinned buffer
                R, own = int(host[0]) & 0xFFFFFFFF, (int(host[1]) & 1) != 0
                ranks = int(own)
            self._pinned.append(host)
            if ranks > 0:
                old = self.session.capacity
                if own:  # regrow this rank's buffers; queued kernels keep the old ones alive (stream-ordered allocator)
                    self.session = self.session.regrown(2 * R + 4096)
                raise BinningOverflow(
                    f"step {step}: {ranks} rank(s) exceeded their binning capacity"
                    + (f" (this rank: R = {R} > {old}; capacity raised to {self.session.capacity})" if own else "")
                    + "; the step's gradients were zeroed on every rank -- repeat it")

    def check(self):
        """Block until every issued step has been examined (raises BinningOverflow for an overflowed one)."""
        self._examine(block_older_than=self.steps)

    def __call__(self, cam, bg, gt, mask, reduce=True):
        """Returns (color, alpha, radii); afterwards self.grads[name] holds the (mean over ranks, if reduce) gradients."""
        self._examine(block_older_than=self.steps - self.max_in_flight)
        s, b = self.session, self.bucket
        host = self._report_buffer()
        if self.world == 1:
            # single process: the forward writes its status words (R, flags) straight into this step's pinned buffer (device-
            # mapped host memory) -- no bookkeeping kernel, no copy; the event below tells the host when they are there
            s.status = host
        s.forward(self.p, cam, bg, self.deg)
        # the loss gradient is formed inside the blend-backward kernel (fastpath.backward_alpha_mask_loss): no loss kernel, no
        # gradient images
        s.backward_alpha_mask_loss(self.p, cam, bg, self.deg, gt, mask, 0.1, self.grads)
        if reduce and self.world > 1:
            self._status(0)
            if self.compact is not None:
                self.compact.pack(s, cam["campos"])
                self.compact.exchange()
            all_reduce_(b.flat, dist.ReduceOp.SUM, self.group)
            self._status(1, host)
            # scale = 1 / world, or 0 if any rank overflowed: a skipped step is skipped by every replica
            b.flat.mul_(self._scale)
            if self.compact is not None:
                self.compact.reconstruct(self.p["means3D"], self.deg, self._scale)
        elif self.world > 1:  # an un-reduced step of a multi-rank job: local bookkeeping only
            self._status(2, host)
        # (a single process needs no scaling: its own overflowed step has exactly zero gradients already)
        ev = self._events.pop() if self._events else torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self._scale.device))
        self.steps += 1
        self.pending.append((self.steps - 1, host, ev))
        return s.color, s.alpha, s.radii
