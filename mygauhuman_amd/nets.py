"""The per-Gaussian skinning-weight offset network of render() with a fused forward (csrc/mlp.hip).

    dec = FusedLBSOffsetDecoder()                      # same parameter names / shapes as nets/mlp_delta_weight_lbs.py:5-32
    dec.load_state_dict(reference_module.state_dict())  # bw_linears.{0..3}.{weight,bias}, bw_fc.{weight,bias}
    pc.lweight_offset_decoder = dec                      # render() calls it with xyz [1, P, 3] and permutes the [1, 24, P] result

The reference runs this network on every Gaussian every frame when motion_offset_flag is set (gaussian_renderer/__init__.py:100-106):
a 63-d positional embedding through 63-128-128-128-(63+128)-128-24 with ReLU -- 27 GFLOP forward at 200k points, which plain torch
spends in skinny fp32 GEMMs and 100-MB elementwise kernels.  When no gradient is being recorded (render.py,
evaluation, every eval_*.sh of the reference) forward() is ONE kernel on the matrix cores: f32 MFMA, activations in registers, the
same accuracy as the torch ops.  While gradients are recorded the same arithmetic runs in torch ops (the fused backward is not built).
Tensors must live on the GPU for the fused path; there is no CPU path for it.
"""
import ctypes as C

import torch

from ._lib import check, lib, ptr

_OCTAVES = 10


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
            n = int(lib.gsr_lbs_offset_mlp_packed_floats())
            if self._packed is None or self._packed.device != dev:
                self._packed = torch.empty(n, dtype=torch.float32, device=dev)
            mk = lambda xs: (C.c_void_p * 5)(*[x.data_ptr() for x in xs])  # noqa: E731  (host arrays of device pointers)
            with torch.cuda.device(dev):
                check(lib.gsr_lbs_offset_mlp_pack(mk([m.weight for m in self._layers()]), mk([m.bias for m in self._layers()]),
                                                  ptr(self._packed), torch.cuda.current_stream(dev).cuda_stream), "gsr_lbs_offset_mlp_pack")
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
        needs_grad = torch.is_grad_enabled() and (pts.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            return self.forward_torch(pts)
        if not pts.is_cuda:
            raise RuntimeError("FusedLBSOffsetDecoder: tensors must live on a HIP device (no CPU path)")
        dev = pts.device
        x = pts[0].detach().contiguous().float()
        P = x.shape[0]
        out = torch.empty((P, self.total_bones), dtype=torch.float32, device=dev)
        packed = self._packed_weights(dev)
        with torch.cuda.device(dev):
            check(lib.gsr_lbs_offset_mlp_forward(P, ptr(x), ptr(packed), ptr(out), torch.cuda.current_stream(dev).cuda_stream),
                  "gsr_lbs_offset_mlp_forward")
        return out.t()[None]
