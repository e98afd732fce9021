"""The per-Gaussian skinning-weight offset network of render() with a fused forward (csrc/mlp.hip).

    dec = FusedLBSOffsetDecoder()                      # same parameter names / shapes as nets/mlp_delta_weight_lbs.py:5-32
    dec.load_state_dict(reference_module.state_dict())  # bw_linears.{0..3}.{weight,bias}, bw_fc.{weight,bias}
    pc.lweight_offset_decoder = dec                      # render() calls it with xyz [1, P, 3] and permutes the [1, 24, P] result

The reference runs this network on every Gaussian every frame when motion_offset_flag is set (gaussian_renderer/__init__.py:100-106):
a 63-d positional embedding through 63-128-128-128-(63+128)-128-24 with ReLU -- 27 GFLOP forward at 200k points, which plain torch
spends in skinny fp32 GEMMs and 100-MB elementwise kernels.  Here forward() is ONE kernel on the matrix cores (f32 MFMA, activations
in registers, the accuracy of the torch ops) and the backward two (the forward again + dh = W^T dZ chained the same way; the
weight gradients as products over the points).  render() hands the positions in DETACHED (:104): a `pts` that requires grad takes
the same arithmetic in torch ops instead (forward_torch).  Tensors must live on the GPU: there is no CPU path for the fused kernels.
"""
import ctypes as C

import torch

from ._lib import check, lib, ptr

_OCTAVES = 10


def set_precision(mode):
    """"bf16x3" (default: the bf16 matrix instruction with both operands split in two bf16 terms -- 7e-6 of the output's scale against
    float64, 2.7 x the speed) or "f32" (the f32 matrix instruction, 6e-7) for the forward and the backward's chain; process-wide."""
    check(lib.gsr_lbs_offset_mlp_set_precision({"f32": 0, "bf16x3": 1}[mode]), "gsr_lbs_offset_mlp_set_precision")


def positional_embedding(x):
    """[P, 3] -> [P, 63]: (x, sin(2^o x), cos(2^o x), o = 0..9) in the order of get_embedder(10) (nets/mlp_delta_weight_lbs.py:34-77)."""
    freqs = 2.0 ** torch.arange(_OCTAVES, dtype=x.dtype, device=x.device)
    ang = x[:, None, :] * freqs[:, None]
    return torch.cat((x, torch.stack((torch.sin(ang), torch.cos(ang)), dim=2).reshape(x.shape[0], -1)), dim=1)


class FusedLBSOffsetDecoder(torch.nn.Module):
    def __init__(self, total_bones=24):
        super().__init__()
        if total_bones != 24:
            raise ValueError("FusedLBSOffsetDecoder: built for the 24 SMPL joints")
        self.total_bones = total_bones
        E, W = 3 + 3 * 2 * _OCTAVES, 128
        self.bw_linears = torch.nn.ModuleList([torch.nn.Conv1d(E, W, 1), torch.nn.Conv1d(W, W, 1), torch.nn.Conv1d(W, W, 1),
                                               torch.nn.Conv1d(W + E, W, 1)])
        self.bw_fc = torch.nn.Conv1d(W, total_bones, 1)
        self._packed, self._packed_key = None, None
        self.use_fused = True   # False: the same arithmetic in torch ops (forward_torch), for comparison

    def _layers(self):
        return list(self.bw_linears) + [self.bw_fc]

    def _packed_weights(self, dev):
        """The A fragments of the five layers, re-packed whenever a parameter changed (in-place updates bump `_version`)."""
        ts = [t for m in self._layers() for t in (m.weight, m.bias)]
        key = tuple((t.data_ptr(), t._version) for t in ts) + (str(dev),)
        if self._packed is None or self._packed_key != key:
            for t in ts:
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise RuntimeError("FusedLBSOffsetDecoder: parameters must be contiguous float32 tensors on a HIP device")
            self._packed = _pack(ts, dev, None if (self._packed is None or self._packed.device != dev) else self._packed)
            self._packed_key = key
        return self._packed

    def forward_torch(self, pts):
        """The reference's arithmetic in torch ops (row-major linear layers on [P, C]); differentiable."""
        emb = positional_embedding(pts[0])
        h = emb
        for i, m in enumerate(self.bw_linears):
            if i == 3:
                h = torch.cat((emb, h), dim=1)
            h = torch.relu(torch.addmm(m.bias, h, m.weight[:, :, 0].t()))
        return torch.addmm(self.bw_fc.bias, h, self.bw_fc.weight[:, :, 0].t()).t()[None]

    def forward(self, pts):
        """pts [1, P, 3] -> [1, 24, P]."""
        if not self.use_fused or (torch.is_grad_enabled() and pts.requires_grad):
            return self.forward_torch(pts)   # (an input gradient is not built: render() detaches the positions)
        if not pts.is_cuda:
            raise RuntimeError("FusedLBSOffsetDecoder: tensors must live on a HIP device (no CPU path)")
        x = pts[0].detach().contiguous().float()
        params = [t for m in self._layers() for t in (m.weight, m.bias)]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _FusedOffsetNet.apply(x, *params).t()[None]
        return _forward_fused(x, self._packed_weights(x.device)).t()[None]


def _pack(params, dev, out=None):
    n = int(lib.gsr_lbs_offset_mlp_packed_floats())
    packed = torch.empty(n, dtype=torch.float32, device=dev) if out is None else out
    mk = lambda xs: (C.c_void_p * 5)(*[x.data_ptr() for x in xs])  # noqa: E731  (host arrays of device pointers)
    with torch.cuda.device(dev):
        check(lib.gsr_lbs_offset_mlp_pack(mk(params[0::2]), mk(params[1::2]), ptr(packed), torch.cuda.current_stream(dev).cuda_stream),
              "gsr_lbs_offset_mlp_pack")
    return packed


def _forward_fused(x, packed):
    out = torch.empty((x.shape[0], 24), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.gsr_lbs_offset_mlp_forward(x.shape[0], ptr(x), ptr(packed), ptr(out), torch.cuda.current_stream(x.device).cuda_stream),
              "gsr_lbs_offset_mlp_forward")
    return out


class _FusedOffsetNet(torch.autograd.Function):
    """x [P, 3] (no gradient), the ten parameter tensors -> [P, 24]; backward = gsr_lbs_offset_mlp_backward."""

    @staticmethod
    def forward(ctx, x, *params):
        ps = [p.detach() for p in params]
        for t in ps:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise RuntimeError("FusedLBSOffsetDecoder: parameters must be contiguous float32 tensors on a HIP device")
        packed = _pack(ps, x.device)   # a buffer of its own: it must still describe THESE parameters when backward runs
        ctx.save_for_backward(x, packed)
        ctx.shapes = [tuple(p.shape) for p in params]
        return _forward_fused(x, packed)

    @staticmethod
    def backward(ctx, g):
        x, packed = ctx.saved_tensors
        dev, P = x.device, x.shape[0]
        g = g.contiguous().float()
        # one zero fill for the ten gradient tensors (the kernels ADD into them): views of one flat buffer
        sizes = [int(torch.Size(s).numel()) for s in ctx.shapes]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        grads, off = [], 0
        for shp, n in zip(ctx.shapes, sizes):
            grads.append(flat[off:off + n].view(shp))
            off += n
        ws = torch.empty(int(lib.gsr_lbs_offset_mlp_backward_workspace_floats(P)), dtype=torch.float32, device=dev)
        mk = lambda xs: (C.c_void_p * 5)(*[t.data_ptr() for t in xs])  # noqa: E731
        with torch.cuda.device(dev):
            check(lib.gsr_lbs_offset_mlp_backward(P, ptr(x), ptr(packed), ptr(g), ptr(ws), mk(grads[0::2]), mk(grads[1::2]),
                                                  torch.cuda.current_stream(dev).cuda_stream), "gsr_lbs_offset_mlp_backward")
        return (None, *grads)


LBSOffsetDecoder = FusedLBSOffsetDecoder   # the reference's class name (`from nets.mlp_delta_weight_lbs import LBSOffsetDecoder`)
