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
from collections import OrderedDict

import torch
import torch.distributed as dist

from ._lib import check, lib


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
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
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
        if self.world > 1:
            mine = self.mine.clone()  # the send buffer must not alias the receive buffer
            try:
                dist.all_gather_into_tensor(self.gathered.view(-1), mine, group=self.group)
            except (RuntimeError, NotImplementedError):  # back-ends without the flat variant (gloo rehearsals)
                dist.all_gather(list(self.gathered.unbind(0)), mine, group=self.group)

    def reconstruct(self, means3D, sh_degree):
        """grad[P,M,3] = mean over the views of w_k(dir_v) * dL_dRGB_v."""
        dev = self.grad.device
        check(lib.gsr_sh_grad_from_views(self.P, int(sh_degree), self.M, self.world, means3D.data_ptr(), self.gathered.data_ptr(),
                                         self.stride, 1.0 / self.world, self.grad.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream), "gsr_sh_grad_from_views")
        return self.grad


def all_reduce_densify_stats(grad_norm_accum, denom, max_radii2D, group=None):
    """Sum the per-view densification statistics and max the screen radii across ranks, in place
    (add_densification_stats, scene/gaussian_model.py:764-766; max_radii2D update, train.py:403)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    P = grad_norm_accum.shape[0]
    packed = torch.cat([grad_norm_accum.reshape(P), denom.reshape(P).to(grad_norm_accum.dtype)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    grad_norm_accum.copy_(packed[:P].view_as(grad_norm_accum))
    denom.copy_(packed[P:].view_as(denom).to(denom.dtype))
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def view_for_step(step, rank, world):
    """Step s assigns view (s * world + rank) to this rank."""
    return step * world + rank


class ViewParallelStep:
    """fwd + loss gradient + bwd of one camera view on this rank, gradients into the flat bucket, then the all-reduce.

    Loss = L1(color, gt) + 0.1 * MSE(alpha, mask) (the bench's "alpha-mask loss", train.py:261-262); its gradient is
    formed by one fused HIP kernel, no autograd graph is built.  Uses a sync-free RasterSession (fastpath.py): the
    host never waits for the GPU inside a step."""

    def __init__(self, params, sh_degree, cam, bg, group=None, slack=1.3, compact_sh=None):
        from .fastpath import RasterSession
        self.p = params  # dict: means3D, shs, opacities, scales, rotations (device tensors)
        self.deg = sh_degree
        self.group = group
        P, M = params["means3D"].shape[0], params["shs"].shape[1]
        dev = params["means3D"].device
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if compact_sh is None:
            compact_sh = world > 1 and os.environ.get("GSR_COMPACT_SH", "1") != "0"
        self.compact = CompactShExchange(P, M, dev, group) if compact_sh else None
        self.bucket = GradientBucket(gaussian_gradient_shapes(P, M, "sh_compact" if compact_sh else "sh"), dev)
        self.grads = dict(self.bucket.views)
        if compact_sh:
            self.grads["sh"] = self.compact.grad  # the local backward writes this view's dL_dsh here; reconstruct() replaces it
        self.session = RasterSession.calibrated(params, cam, bg, sh_degree, slack=slack)

    @property
    def payload_bytes(self):
        """Bytes this rank contributes to the collectives of one step."""
        return self.bucket.nbytes + (self.compact.stride * 4 if self.compact else 0)

    def __call__(self, cam, bg, gt, mask, reduce=True):
        """Returns (color, alpha, radii); afterwards self.grads[name] holds the (mean over ranks, if reduce) gradients."""
        s, b = self.session, self.bucket
        s.forward(self.p, cam, bg, self.deg)
        dc, da = s.alpha_mask_loss_backward(gt, mask, 0.1)
        s.backward(self.p, cam, bg, self.deg, dc, s.dL_ddepth, da, self.grads)
        if reduce:
            if self.compact is not None:
                self.compact.pack(s, cam["campos"])
                self.compact.exchange()
                b.all_reduce_mean(self.group)
                self.compact.reconstruct(self.p["means3D"], self.deg)
            else:
                b.all_reduce_mean(self.group)
        return s.color, s.alpha, s.radii
