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


class ExchangeTimer:
    """HIP events around the collectives of every step (recorded on the compute stream the collectives are ordered on; no
    synchronisation until read): what `exchange_ms` in the bench line is made of."""

    def __init__(self):
        self.pairs, self.pool = [], []

    def begin(self, device):
        e0, e1 = self.pool.pop() if self.pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e0.record(torch.cuda.current_stream(device))
        self.pairs.append((e0, e1))

    def end(self, device):
        self.pairs[-1][1].record(torch.cuda.current_stream(device))

    def reset(self):
        self.pool.extend(self.pairs)
        self.pairs = []

    def read_ms(self):
        """Per-step exchange times of the steps since reset() (synchronises on the last one)."""
        if not self.pairs:
            return []
        self.pairs[-1][1].synchronize()
        return [a.elapsed_time(b) for a, b in self.pairs]


_SELFTEST_DONE = set()


def collective_selftest(device, group=None):
    """Run the two collectives of a step once on 64 floats and CHECK the results, at construction time: the first use of a
    backend in a job (RCCL on a multi-GPU node) fails here, loudly and with its own message, not in the middle of a timed
    loop.  Returns the all-gather form to use: "tensor" (all_gather_into_tensor, out of place) or "list" (the list form, if
    the tensor form is refused by this backend)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    key = (id(group), str(device))
    form = "tensor"
    send = torch.full((64,), float(rank + 1), dtype=torch.float32, device=device)
    recv = torch.zeros((world * 64,), dtype=torch.float32, device=device)
    want = torch.arange(1, world + 1, dtype=torch.float32, device=device).repeat_interleave(64)
    staged = _staged(send, group)

    def agree(ok):
        """MIN over the ranks of a success flag, through host memory (gloo-style staging works on every backend and does not
        depend on the collective under test): every rank takes the same branch below, so no rank posts a collective its peers
        never post."""
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32)
        objs = [None] * world
        dist.all_gather_object(objs, float(flag[0]), group=group)
        return min(objs) >= 1.0

    err = None
    if staged:
        ok_here = False   # host-staged rehearsal: list form by construction (the same on every rank)
    else:
        try:
            dist.all_gather_into_tensor(recv, send, group=group)
            torch.cuda.synchronize(device) if torch.device(device).type == "cuda" else None  # (asynchronous RCCL errors surface here)
            ok_here = bool(torch.equal(recv, want))
            if not ok_here:
                err = "all_gather_into_tensor returned wrong data"
        except RuntimeError as ex:   # (a refused collective; anything else is a bug and propagates)
            ok_here, err = False, str(ex)
    if staged or not agree(ok_here):
        form = "list"
        if key not in _SELFTEST_DONE and not staged and rank == 0:
            print(f"[mygauhuman_amd.parallel] all_gather_into_tensor unusable on backend {dist.get_backend(group)} on at least one "
                  f"rank ({err or 'a peer reported it'}); every rank falls back to the list form", flush=True)
        h = send.cpu() if staged else send
        parts = [torch.empty_like(h) for _ in range(world)]
        ok_list, got = True, None
        try:
            dist.all_gather(parts, h, group=group)
            got = torch.cat(parts).to(device)
            ok_list = bool(torch.equal(got, want))
        except RuntimeError:
            ok_list = False
        if not agree(ok_list):   # every rank raises, none is left waiting in a collective
            raise RuntimeError(f"all_gather self-test failed on backend {dist.get_backend(group)} (tensor and list form) on at least one rank")
    red = send.clone()
    all_reduce_(red, dist.ReduceOp.SUM, group)
    if not agree(bool(torch.equal(red, torch.full_like(red, world * (world + 1) / 2.0)))):
        raise RuntimeError(f"all_reduce self-test failed on backend {dist.get_backend(group)} on at least one rank")
    _SELFTEST_DONE.add(key)
    return form


class CompactShExchange:
    """All-gather of one block per rank + local reconstruction of the mean SH gradient.

    static Gaussians (ViewParallelStep):   block = [P*3 masked dL_dRGB | campos 3]; every replica holds the same positions
    posed Gaussians (ViewParallelRender):  block = [P*3 masked dL_dRGB | P*3 positions of THIS view | campos 3 | P radii]:
                                           every view poses the Gaussians differently (LBS), so the positions the view's SH
                                           colours were evaluated at travel with it; the screen radii ride along for the
                                           max-reduction of the densification statistics (train.py:403).
    The send buffer is a tensor of its own (the pack kernel writes it; nothing is copied) and the receive buffer is
    [world][stride]: a plain out-of-place all_gather_into_tensor -- no aliasing of send and receive memory."""

    def __init__(self, P, M, device, group=None, posed=False, side_stream=None):
        """side_stream: None = a side stream + communicator of its own when world > 1 on a device backend; True forces them (the
        one-rank RCCL smoke test); False = everything on the compute stream."""
        self.P, self.M, self.group, self.posed = int(P), int(M), group, bool(posed)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        P = self.P
        self.means_off = P * 3 if posed else 0
        self.cam_off = P * 6 if posed else P * 3
        self.radii_off = self.cam_off + 4 if posed else 0
        used = self.radii_off + P if posed else P * 3 + 3
        self.stride = (used + 63) // 64 * 64
        self.gathered = torch.zeros((self.world, self.stride), dtype=torch.float32, device=device)
        self.mine = torch.zeros((self.stride,), dtype=torch.float32, device=device)
        if posed:
            self.grad_dc = torch.empty((P, 1, 3), dtype=torch.float32, device=device)
            self.grad_rest = torch.empty((P, 15, 3), dtype=torch.float32, device=device)
            self.grad = None
        else:
            self.grad = torch.empty((P, self.M, 3), dtype=torch.float32, device=device)
        self.form = "tensor"
        # The all-gather runs on a SIDE stream through a communicator of its own (SURVEY.md section 8e "overlap"): its payload is
        # final long before the all-reduce's -- after the attribute kernel's backward in the articulated path, after the backward
        # preprocess in the static one -- so it is issued there (exchange_async) and travels under what the compute stream still has
        # to do (LBS / activation backward, the bucket copy) and under the all-reduce itself; two collectives in flight at once need
        # two communicators.  wait() orders the compute stream behind it right before the reconstruction reads the blocks.
        # Host-staged backends (gloo rehearsals, CPU tests) are synchronous by nature: exchange_async() completes before it returns.
        self.side, self.ag_group, self._ag_done, self._ag_timing = None, group, None, []
        if self.world > 1:
            self.form = collective_selftest(torch.device(device), group)
        # (GSR_EXCHANGE_SIDE_STREAM=0: everything on the compute stream through the one communicator, as in round 3)
        want_side = ((self.world > 1 and os.environ.get("GSR_EXCHANGE_SIDE_STREAM", "1") != "0") if side_stream is None
                     else bool(side_stream))
        if want_side and dist.is_available() and dist.is_initialized():
            if torch.device(device).type == "cuda" and not _staged(self.mine, group):
                self.side = torch.cuda.Stream(device=device)
                self.ag_group = dist.new_group(ranks=None if group is None else dist.get_process_group_ranks(group),
                                               backend=dist.get_backend(group))

    def pack(self, session, campos):
        """Static path: fill this rank's block from the session's last backward (its raw dL_dcolor + the forward's clamp bits)."""
        dev = self.mine.device
        check(lib.gsr_sh_view_pack(self.P, session.geom.data_ptr(), session.dL_dcolors.data_ptr(), self.mine.data_ptr(),
                                   torch.cuda.current_stream(dev).cuda_stream), "gsr_sh_view_pack")
        self.mine[self.P * 3:self.P * 3 + 3].copy_(campos.reshape(3))

    def pack_posed(self, colors, g_colors, means_view, campos, radii=None):
        """Articulated path: masked dL_dRGB, this view's posed positions and camera position in one launch (+ the radii)."""
        dev = self.mine.device
        check(lib.gsr_sh_view_pack_posed(self.P, colors.data_ptr(), g_colors.data_ptr(), means_view.data_ptr(), campos.data_ptr(),
                                         self.mine.data_ptr(), self.means_off, self.cam_off,
                                         torch.cuda.current_stream(dev).cuda_stream), "gsr_sh_view_pack_posed")
        if radii is not None:
            self.mine[self.radii_off:self.radii_off + self.P].copy_(radii)  # int32 -> float32 (exact below 2^24 pixels)

    def local(self):
        """No exchange (single process, or an un-reduced step): this rank's block becomes view 0; reconstruct with n_views=1."""
        self.gathered[0].copy_(self.mine)

    def exchange_async(self, timing=False):
        """Issue the all-gather of this rank's block NOW (the block must be packed): on the side stream, behind everything the
        current stream has queued so far.  Follow with wait() before reconstruct() / max_radii()."""
        if self.side is None:
            self.exchange()
            return
        cur = torch.cuda.current_stream(self.mine.device)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            e0 = e1 = None
            if timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.side)
            if self.form == "tensor":
                dist.all_gather_into_tensor(self.gathered.view(-1), self.mine, group=self.ag_group)
            else:
                parts = [torch.empty_like(self.mine) for _ in range(self.world)]
                dist.all_gather(parts, self.mine, group=self.ag_group)
                self.gathered.copy_(torch.stack(parts))
            if timing:
                e1.record(self.side)
                self._ag_timing.append((e0, e1))
            self._ag_done = torch.cuda.Event()
            self._ag_done.record(self.side)
        # (the block and the receive buffer are used by the side stream: the caching allocator must not hand them on early --
        # both are owned by this object for its lifetime, nothing to record)

    def wait(self):
        """Order the current stream behind the all-gather issued by exchange_async()."""
        if self._ag_done is not None:
            torch.cuda.current_stream(self.mine.device).wait_event(self._ag_done)
            self._ag_done = None

    def allgather_ms(self):
        """Per-step all-gather times on the side stream since the last call (synchronises on the last one)."""
        t, self._ag_timing = self._ag_timing, []
        if not t:
            return []
        t[-1][1].synchronize()
        return [a.elapsed_time(b) for a, b in t]

    def exchange(self):
        if self.world <= 1:
            self.local()
            return
        if self.form == "tensor" and not _staged(self.mine, self.group):
            dist.all_gather_into_tensor(self.gathered.view(-1), self.mine, group=self.group)
            return
        # gloo (CPU tests, one-GPU rehearsals) or a backend without the tensor form: list form, host-staged for device tensors
        src = self.mine.cpu() if _staged(self.mine, self.group) else self.mine
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src, group=self.group)
        self.gathered.copy_(torch.stack(parts))

    def reconstruct(self, means3D, sh_degree, dev_scale=None, n_views=None):
        """grad[P,M,3] (or grad_dc / grad_rest) = mean over the views of w_k(dir_v) * dL_dRGB_v; dev_scale: optional device float
        that REPLACES the 1 / n_views factor (0 for a step every replica must skip); n_views: views 0 .. n_views-1 of the
        receive buffer (default: all ranks)."""
        dev = self.gathered.device
        n_views = self.world if n_views is None else int(n_views)
        scale = 1.0 if dev_scale is not None else 1.0 / n_views
        ds = None if dev_scale is None else dev_scale.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self.posed:
            check(lib.gsr_sh_grad_from_views_posed(self.P, int(sh_degree), n_views, self.gathered.data_ptr(), self.stride,
                                                   self.means_off, self.cam_off, scale, ds, self.grad_dc.data_ptr(),
                                                   self.grad_rest.data_ptr(), stream), "gsr_sh_grad_from_views_posed")
            return self.grad_dc, self.grad_rest
        check(lib.gsr_sh_grad_from_views(self.P, int(sh_degree), self.M, n_views, means3D.data_ptr(), self.gathered.data_ptr(),
                                         self.stride, scale, ds, self.grad.data_ptr(), stream), "gsr_sh_grad_from_views")
        return self.grad

    def max_radii(self, n_views=None):
        """Max over the views of the screen radii that rode along (posed layout)."""
        n_views = self.world if n_views is None else int(n_views)
        return self.gathered[:n_views, self.radii_off:self.radii_off + self.P].max(dim=0).values.to(torch.int32)


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
        self.timer = ExchangeTimer()      # the whole exchange as the compute stream sees it (all-reduce + waiting for the all-gather)
        self.ar_timer = ExchangeTimer()   # the all-reduce alone; the all-gather times itself on its side stream (compact.allgather_ms)
        if world > 1:
            collective_selftest(dev, group)

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
        dev = self._scale.device
        host = self._report_buffer()
        if self.world == 1:
            # single process: the forward writes its status words (R, flags) straight into this step's pinned buffer (device-
            # mapped host memory) -- no bookkeeping kernel, no copy; the event below tells the host when they are there
            s.status = host
        s.forward(self.p, cam, bg, self.deg)
        # the loss gradient is formed inside the blend-backward kernel (fastpath.backward_alpha_mask_loss): no loss kernel, no
        # gradient images
        s.backward_alpha_mask_loss(self.p, cam, bg, self.deg, gt, mask, 0.1, self.grads)
        if self.world > 1:
            self._status(0)  # this rank's overflow flag -> its slot of the bucket (0 / 1)
            self.timer.begin(dev)
            if reduce:
                if self.compact is not None:
                    # the all-gather (side stream, own communicator) and the all-reduce (this stream) travel at the same time
                    self.compact.pack(s, cam["campos"])
                    self.compact.exchange_async(timing=True)
                self.ar_timer.begin(dev)
                all_reduce_(b.flat, dist.ReduceOp.SUM, self.group)
                self.ar_timer.end(dev)
                if self.compact is not None:
                    self.compact.wait()
            else:
                # an un-reduced step of a multi-rank job keeps its gradients local, but the ranks still AGREE on the overflow word
                # (one 4-byte all-reduce): every rank raises BinningOverflow at the same call, or none does -- a rank that raised
                # alone would leave its peers waiting in their next collective
                all_reduce_(b["overflow"], dist.ReduceOp.SUM, self.group)
            self.timer.end(dev)
            # scale = 1 / world, or 0 if any rank overflowed (a skipped step is skipped by every replica), the whole bucket times
            # scale and the report words: one launch (gsr_step_finish)
            check(lib.gsr_step_finish(s.status.data_ptr(), b.flat.data_ptr(), b.flat.numel(), b.slices["overflow"][0],
                                      (1.0 / self.world) if reduce else 1.0, self._scale.data_ptr(), host.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream), "gsr_step_finish")
            if reduce and self.compact is not None:
                self.compact.reconstruct(self.p["means3D"], self.deg, self._scale)
        # (a single process needs no scaling: its own overflowed step has exactly zero gradients already)
        ev = self._events.pop() if self._events else torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self._scale.device))
        self.steps += 1
        self.pending.append((self.steps - 1, host, ev))
        return s.color, s.alpha, s.radii


class _DetachedShModel:
    """The model as render() reads it, with the two SH parameter tensors detached: the attribute kernel's backward then skips
    the 192-byte-per-Gaussian SH gradient (ViewParallelRender rebuilds its mean over the views from the compact exchange)."""

    def __init__(self, model):
        object.__setattr__(self, "_m", model)

    def __getattr__(self, name):
        m = object.__getattribute__(self, "_m")
        if name in ("_features_dc", "_features_rest"):
            return getattr(m, name).detach()
        return getattr(m, name)


class _ShSink:
    """What the attribute kernel's backward hands to the compact exchange (attributes.sh_gradient_sink).  With `early` set (the
    exchange object, the camera position and the radii of this view) the block is packed and its all-gather issued RIGHT THERE, in
    the middle of autograd's backward: it travels under the LBS / activation backward kernels that follow."""

    def __init__(self):
        self.got, self.early, self.sent = None, None, False

    def collect(self, colors, g_colors, means_view):
        self.got = (colors, g_colors, means_view)
        if self.early is not None:
            ex, campos, radii = self.early
            ex.pack_posed(colors, g_colors, means_view, campos, radii)
            ex.exchange_async(timing=True)
            self.sent = True


class ViewParallelRender:
    """One view-parallel TRAINING step of the articulated model -- what train.py:212-224,401-417 does for one camera, for `world`
    cameras at once: every rank renders its own camera / pose of the shared model through gaussian_renderer.render() (LBS deform ->
    attributes -> fused rasterizer), evaluates the loss, runs autograd's backward, and then ONE exchange leaves on every rank

      * the MEAN over the views of the gradient of every model leaf -- `_xyz, _features_dc, _features_rest, _opacity, _scaling,
        _rotation, _normal, _albedo, _roughness` (scene/gaussian_model.py:55-127; a leaf the loss does not reach contributes
        zeros) and every parameter of `pose_decoder` / `lweight_offset_decoder` (gaussian_renderer/__init__.py:100-106) -- in
        `p.grad`, as views of one flat fp32 bucket (SUM all-reduce, divided on the device);
      * the densification statistics of the step (scene/gaussian_model.py:764-766, train.py:403): the sum over the views of the
        screen-space gradient norms of the visible Gaussians, the number of views that saw each Gaussian, the max screen radius
        -- in `.stat_grad_norm`, `.stat_visible`, `.max_radii` (they ride in the same bucket / the same all-gather: no
        collective of their own in the compact mode).

    compact_sh (default for world > 1): the SH coefficients are 48 of the 65 gradient floats per Gaussian, and their gradient is
    rank one per view, dL_dsh[k][c] = w_k(dir) dL_dRGB[c].  The ranks all-gather (masked dL_dRGB | posed position | radius) --
    28 B per Gaussian -- and rebuild the mean locally with the positions EACH VIEW posed the Gaussians at (CompactShExchange,
    posed layout); 68 B per Gaussian stay in the all-reduce instead of 260.

    A rank whose binning buffer overflowed (rare: _C.AsyncCapacity sizes it generously) contributes zeros and raises the
    overflow count in the bucket; every replica then multiplies the step by zero and raises BinningOverflow at the same later
    call (examined `max_in_flight` steps late through pinned memory, like ViewParallelStep: no host stall)."""

    MODEL_LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_normal", "_albedo", "_roughness")

    def __init__(self, model, pipe, bg, group=None, compact_sh=None, max_in_flight=2):
        self.model, self.pipe, self.bg, self.group = model, pipe, bg, group
        dev = model.get_xyz.device
        self.device = dev
        self.world = world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.P = P = int(model.get_xyz.shape[0])
        leaves = OrderedDict()
        for n in self.MODEL_LEAVES:
            t = getattr(model, n, None)
            if isinstance(t, torch.Tensor) and t.numel() and t.requires_grad:
                leaves[n] = t
        for mod_name in ("pose_decoder", "lweight_offset_decoder"):
            mod = getattr(model, mod_name, None)
            if isinstance(mod, torch.nn.Module):
                for n, p_ in mod.named_parameters():
                    if p_.requires_grad:
                        leaves[f"{mod_name}.{n}"] = p_
        self.leaves = leaves
        dc, rest = leaves.get("_features_dc"), leaves.get("_features_rest")
        can_compact = (dc is not None and rest is not None and tuple(dc.shape) == (P, 1, 3) and tuple(rest.shape) == (P, 15, 3)
                       and getattr(pipe, "convert_SHs_python", False) and not getattr(pipe, "separate_feature_passes", False))
        if compact_sh is None:
            compact_sh = world > 1 and os.environ.get("GSR_COMPACT_SH", "1") != "0"
        self.compact = CompactShExchange(P, 16, dev, group, posed=True) if (compact_sh and can_compact) else None
        shapes = OrderedDict((n, tuple(t.shape)) for n, t in leaves.items()
                             if not (self.compact is not None and n in ("_features_dc", "_features_rest")))
        shapes["stat_grad_norm"] = (P, 1)   # xyz_gradient_accum contribution of this step (scene/gaussian_model.py:765)
        shapes["stat_visible"] = (P, 1)     # denom contribution (:766)
        shapes["overflow"] = (1,)
        self.bucket = GradientBucket(shapes, dev)
        self._bucket_leaves = [n for n in shapes if n in leaves]
        self.max_radii = torch.zeros((P,), dtype=torch.int32, device=dev)
        self.max_in_flight = int(max_in_flight)
        self.steps = 0
        self.pending = deque()
        self._pinned, self._events = [], []
        self._scale = torch.zeros(1, dtype=torch.float32, device=dev)
        self.timer = ExchangeTimer()      # the whole exchange as the compute stream sees it (all-reduce + waiting for the all-gather)
        self.ar_timer = ExchangeTimer()   # the all-reduce alone; the all-gather times itself on its side stream (compact.allgather_ms)
        self._view = _DetachedShModel(model) if self.compact is not None else model
        if world > 1 and self.compact is None:
            collective_selftest(dev, group)

    @property
    def stat_grad_norm(self):
        return self.bucket["stat_grad_norm"]

    @property
    def stat_visible(self):
        return self.bucket["stat_visible"]

    @property
    def payload_bytes(self):
        return self.bucket.nbytes + (self.compact.stride * 4 if self.compact is not None else 0)

    def accumulate_densification_stats(self):
        """add_densification_stats + the max_radii2D update of the step (scene/gaussian_model.py:764-766, train.py:403) from the
        statistics the exchange left: sums over ALL views of the step."""
        m = self.model
        m.xyz_gradient_accum += self.stat_grad_norm
        m.denom += self.stat_visible
        m.max_radii2D = torch.maximum(m.max_radii2D, self.max_radii.to(m.max_radii2D.dtype))

    def _examine(self, block_older_than):
        while self.pending:
            step, host, ev = self.pending[0]
            if step > block_older_than:
                return
            ev.synchronize()
            self.pending.popleft()
            self._events.append(ev)
            ranks = int(host[0])
            self._pinned.append(host)
            if ranks > 0:
                raise BinningOverflow(f"step {step}: {ranks} rank(s) exceeded their binning capacity; the step's gradients were "
                                      "zeroed on every rank (the capacity of the overflowing rank has been raised) -- repeat it")

    def check(self):
        self._examine(block_older_than=self.steps)

    def _current_leaves(self):
        cur = OrderedDict()
        for n in self.MODEL_LEAVES:
            t = getattr(self.model, n, None)
            if isinstance(t, torch.Tensor) and t.numel() and t.requires_grad:
                cur[n] = t
        for mod_name in ("pose_decoder", "lweight_offset_decoder"):
            mod = getattr(self.model, mod_name, None)
            if isinstance(mod, torch.nn.Module):
                for n, p_ in mod.named_parameters():
                    if p_.requires_grad:
                        cur[f"{mod_name}.{n}"] = p_
        return cur

    def _rebind_leaves(self):
        """The model may REPLACE a leaf between two steps without changing its shape -- densify.reset_opacity (train.py:412) installs
        a new nn.Parameter for `_opacity`, optimizer surgery does the same for every group -- and render() then backpropagates into
        the new tensor.  A step object that kept the old one would zero, reduce and hand back the gradient of a dead tensor while
        the live one accumulated its own un-reduced gradient: replicas diverge silently.  So every call compares the
        model's leaves with the ones this object is bound to: same names and shapes -> the new tensors are adopted (the bucket is
        sized by shapes, nothing else changes); anything else -- densify / prune changed P, a leaf appeared or vanished -- raises and
        asks for a new ViewParallelRender, whose bucket is sized for the new model."""
        cur = self._current_leaves()
        if int(self.model.get_xyz.shape[0]) != self.P or list(cur) != list(self.leaves) or any(
                tuple(cur[n].shape) != tuple(self.leaves[n].shape) for n in cur):
            raise RuntimeError(
                "ViewParallelRender: the model's leaves changed (Gaussian count %d -> %d, or a leaf appeared / vanished / changed "
                "shape): build a new ViewParallelRender after densify / prune" % (self.P, int(self.model.get_xyz.shape[0])))
        for n, t in cur.items():
            if t is not self.leaves[n]:
                self.leaves[n] = t

    def __call__(self, iteration, camera, loss_fn, reduce=True, **render_kw):
        """loss_fn(out) -> scalar loss of this rank's view.  Returns (out, loss); afterwards p.grad of every leaf holds the mean
        over the ranks (reduce=True) and the statistics of the step are in .stat_grad_norm / .stat_visible / .max_radii."""
        from . import attributes
        from .diff_gaussian_rasterization import _C as _RasterC
        from .gaussian_renderer import render
        self._examine(block_older_than=self.steps - self.max_in_flight)
        self._rebind_leaves()
        dev, b = self.device, self.bucket
        for t in self.leaves.values():
            t.grad = None
        sink = _ShSink() if self.compact is not None else None
        overflowed = False
        out = loss = None
        try:
            with attributes.sh_gradient_sink(sink):
                out = render(iteration, camera, self._view, self.pipe, self.bg, **render_kw)
                loss = loss_fn(out)
                if sink is not None and self.world > 1 and reduce:
                    sink.early = (self.compact, camera.camera_center, out["radii"])
                loss.backward()
        except _RasterC.BinningCapacityExceeded:
            overflowed = True   # this view rendered only the background: it contributes zeros and one count
        # ---- this rank's contribution into the flat bucket (one multi-tensor copy) + the statistics of the step
        if overflowed or out is None:
            b.flat.zero_()
            b["overflow"].fill_(1.0)
            if self.compact is not None:
                self.compact.mine.zero_()
        else:
            srcs, dsts = [], []
            for n in self._bucket_leaves:
                g = self.leaves[n].grad
                if g is None:
                    b[n].zero_()     # a leaf the loss did not reach (e.g. _roughness: get_roughness reads _albedo)
                else:
                    srcs.append(g)
                    dsts.append(b[n])
            torch._foreach_copy_(dsts, srcs)
            vis = out["visibility_filter"]
            vg = out["viewspace_points"].grad
            if vg is not None:   # add_densification_stats (scene/gaussian_model.py:764-766)
                torch.mul(torch.norm(vg[:, :2], dim=-1, keepdim=True), vis[:, None], out=b["stat_grad_norm"])
            else:
                b["stat_grad_norm"].zero_()
            b["stat_visible"].copy_(vis[:, None])
            b["overflow"].zero_()
            if self.compact is not None:
                if sink.got is None:
                    raise RuntimeError("ViewParallelRender: the compact SH exchange needs render()'s python SH path "
                                       "(pipe.convert_SHs_python = True, no override_color)")
                if not sink.sent:
                    colors, g_colors, means_view = sink.got
                    self.compact.pack_posed(colors, g_colors, means_view, camera.camera_center, out["radii"])
            else:
                self.max_radii.copy_(out["radii"])
        host = self._pinned.pop() if self._pinned else torch.zeros(3, dtype=torch.int32).pin_memory()
        # ---- the exchange
        n_views = self.world if (self.world > 1 and reduce) else 1
        if n_views > 1:
            self.timer.begin(dev)
            if self.compact is not None and not (sink is not None and sink.sent):
                self.compact.exchange_async(timing=True)   # (an overflowed view, or nothing reached the sink: zeros / late pack)
            self.ar_timer.begin(dev)
            all_reduce_(b.flat, dist.ReduceOp.SUM, self.group)
            self.ar_timer.end(dev)
            if self.compact is None:
                all_reduce_(self.max_radii, dist.ReduceOp.MAX, self.group)
            if self.compact is not None:
                self.compact.wait()
            self.timer.end(dev)
        else:
            if self.world > 1:
                # an un-reduced step of a multi-rank job: gradients stay local, but the ranks agree on the overflow word (4 bytes)
                all_reduce_(b["overflow"], dist.ReduceOp.SUM, self.group)
            if self.compact is not None:
                self.compact.local()
        # scale = 1 / n_views (0 if any rank overflowed), the bucket times scale, the report words: one launch
        check(lib.gsr_step_finish(None, b.flat.data_ptr(), b.flat.numel(), b.slices["overflow"][0], 1.0 / n_views,
                                  self._scale.data_ptr(), host.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
              "gsr_step_finish")
        if n_views > 1:  # the statistics are SUMS over the views, not means
            b["stat_grad_norm"].mul_(float(n_views))
            b["stat_visible"].mul_(float(n_views))
        if self.compact is not None:
            self.compact.reconstruct(None, self.model.active_sh_degree, self._scale, n_views=n_views)
            self.max_radii.copy_(self.compact.max_radii(n_views))
        # ---- p.grad = views of the bucket (and of the reconstructed SH gradient)
        for n, t in self.leaves.items():
            if self.compact is not None and n == "_features_dc":
                t.grad = self.compact.grad_dc
            elif self.compact is not None and n == "_features_rest":
                t.grad = self.compact.grad_rest
            else:
                t.grad = b[n]
        ev = self._events.pop() if self._events else torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.steps += 1
        self.pending.append((self.steps - 1, host, ev))
        return out, loss
