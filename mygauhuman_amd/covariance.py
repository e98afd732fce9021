"""3D covariance of the Gaussians on the host side (torch, device-agnostic):
  build_rotation / build_scaling_rotation / strip_symmetric    utils/general_utils.py:64-117 (quaternion normalised)
  build_covariance_from_scaling_rotation                        scene/gaussian_model.py:35-42  (Sigma = L L^T, then T Sigma T^T)
  get_minimum_axis / flip_align_view                            utils/general_utils.py:144-157
Unlike the reference helpers (hard-coded device="cuda") these follow the device of their inputs."""
import torch


def bmm3(A, B):
    """Batched [.,3,3] x [.,3,k] product as broadcast multiply + sum: rocBLAS' strided-batched GEMM needs ~1-3 ms for 200k
    3x3 products, these three elementwise kernels a few tens of microseconds."""
    return (A.unsqueeze(-1) * B.unsqueeze(-3)).sum(-2)


def build_rotation(r):
    q = r / torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=-1).view(-1, 3, 3)


def build_scaling_rotation(s, r):
    return build_rotation(r) * s[:, None, :]  # R @ diag(s)


def strip_symmetric(sym):
    return torch.stack([sym[:, 0, 0], sym[:, 0, 1], sym[:, 0, 2], sym[:, 1, 1], sym[:, 1, 2], sym[:, 2, 2]], dim=-1)


def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation, transform=None):
    L = build_scaling_rotation(scaling_modifier * scaling, rotation)
    cov = bmm3(L, L.transpose(1, 2))
    if transform is not None:
        cov = bmm3(bmm3(transform, cov), transform.transpose(1, 2))
    return strip_symmetric(cov)


def get_minimum_axis(scales, rotations):
    idx = torch.argsort(scales, descending=False, dim=-1, stable=True)
    R = build_rotation(rotations)
    R_sorted = torch.gather(R, dim=2, index=idx[:, None, :].repeat(1, 3, 1))
    # NB: the reference takes ROW 0 of the column-sorted matrix (utils/general_utils.py:148, `R_sorted[:,0,:]`), not the
    # column of the smallest scale; reproduced as is.
    return R_sorted[:, 0, :]


def flip_align_view(normal, viewdir):
    non_flip = torch.sum(normal * -viewdir, dim=-1, keepdim=True) >= 0
    return normal * torch.where(non_flip, 1.0, -1.0), non_flip


def transformVector3x3(v, matrix):
    """transform.py:9-17: v [N,3] times the upper-left 3x3 of a row-vector-convention matrix."""
    return v[:, 0:1] * matrix[0, :3] + v[:, 1:2] * matrix[1, :3] + v[:, 2:3] * matrix[2, :3]
