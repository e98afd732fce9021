"""k-NN service with the call surface of KNN_CUDA 0.2 (`from knn_cuda import KNN`, scene/gaussian_model.py:23,87-89), the
un-vendored CUDA-only dependency of the reference.

    KNN(k, transpose_mode=True)(ref [B,N,3], query [B,M,3]) -> (dist [B,M,k] float32, idx [B,M,k] int64)

Two exact HIP paths cover every call site of the reference:
  * ref IS query (self k-NN, k <= 3; :176 k = 3, :573/:621/:671 k = 2): Morton-ordered 1024-point boxes with bound pruning
    (csrc/knn.hip, the structure simple-knn uses for distCUDA2), the point itself comes back first at distance 0;
  * k = 1 with any ref (:727 distance to the SMPL vertices, :775 nearest vertex): uniform grid over ref (csrc/lbs.hip).
Distances are Euclidean, ascending; ties resolve to the lowest index.  KNN_CUDA's own tie-breaking is not documented and the
package cannot be installed here: parity unpinned, semantics as stated.  No CPU path: CPU tensors raise.
"""
import torch

from ._lib import check, lib, ptr


def _is_same(a, b):
    return a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()


def knn_self(points, k):
    """points [P,3] -> (dist [P,k], idx [P,k] int32)."""
    if not points.is_cuda:
        raise RuntimeError("knn_self: tensors must live on a HIP device (no CPU path)")
    pts = points.detach().contiguous().float()
    P, dev = pts.shape[0], pts.device
    idx = torch.empty((P, k), dtype=torch.int32, device=dev)
    dist = torch.empty((P, k), dtype=torch.float32, device=dev)
    if P:
        ws = torch.empty((lib.gsr_dist2_workspace_bytes(P),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(lib.gsr_knn_self(P, ptr(pts), int(k), ptr(idx), ptr(dist), ptr(ws), ws.numel(),
                                   torch.cuda.current_stream(dev).cuda_stream), "gsr_knn_self")
    return dist, idx


def knn_nearest(ref, query):
    """ref [N,3], query [M,3] -> (dist [M], idx [M] int32) of the nearest reference point."""
    if not query.is_cuda:
        raise RuntimeError("knn_nearest: tensors must live on a HIP device (no CPU path)")
    r, q = ref.detach().contiguous().float(), query.detach().contiguous().float()
    M, N, dev = q.shape[0], r.shape[0], q.device
    idx = torch.empty((M,), dtype=torch.int32, device=dev)
    dist = torch.empty((M,), dtype=torch.float32, device=dev)
    if M:
        ws = torch.empty((lib.gsr_lbs_workspace_bytes(N),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(lib.gsr_knn_nearest(M, ptr(q), N, ptr(r), ptr(idx), ptr(dist), ptr(ws), ws.numel(),
                                      torch.cuda.current_stream(dev).cuda_stream), "gsr_knn_nearest")
    return dist, idx


class KNN(torch.nn.Module):
    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = int(k)
        self._t = bool(transpose_mode)

    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        with torch.no_grad():
            if not self._t:  # [B, dim, N] layout of the default mode
                ref, query = ref.transpose(1, 2), query.transpose(1, 2)
            D, idxs = [], []
            for b in range(ref.shape[0]):
                if self.k <= 3 and _is_same(ref[b], query[b]):
                    d, i = knn_self(ref[b], self.k)
                elif self.k == 1:
                    d, i = knn_nearest(ref[b], query[b])
                    d, i = d[:, None], i[:, None]
                else:
                    raise NotImplementedError("KNN: k > 1 is implemented for ref is query (k <= 3); k = 1 for any ref")
                D.append(d)
                idxs.append(i.long())
            D, idxs = torch.stack(D), torch.stack(idxs)
            if not self._t:
                D, idxs = D.transpose(1, 2).contiguous(), idxs.transpose(1, 2).contiguous()
        return D, idxs
