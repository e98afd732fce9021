"""`simple_knn._C.distCUDA2` (SK/spatial.cu:15-26) over the C ABI of libgsr.so."""
import torch

from .._lib import check, lib


def distCUDA2(points):
    """points [P,3] float32 on a HIP device -> float32 [P]: mean squared distance to the 3 nearest neighbours."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must live on a HIP device (no CPU path)")
    if points.ndimension() != 2 or points.size(1) != 3:
        raise RuntimeError("distCUDA2: points must have dimensions (num_points, 3)")
    P = points.size(0)
    pts = points.contiguous().float()
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)
    if P:
        ws_bytes = lib.gsr_dist2_workspace_bytes(P)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=points.device)
        with torch.cuda.device(points.device):
            check(lib.gsr_dist2(P, pts.data_ptr(), means.data_ptr(), ws.data_ptr(), ws_bytes,
                                torch.cuda.current_stream(points.device).cuda_stream), "gsr_dist2")
    return means
