"""Seeded synthetic scenes of SURVEY.md §8(d) ("S-uniform" / "S-human").

Everything is generated on the CPU with numpy's default_rng so every device and the oracle see
identical bits; callers upload the arrays.
"""
import math

import numpy as np

from .cameras import make_camera


def uniform_gaussians(P, seed=0, sh_degree=3, fov_deg=50.0, log_scale_mean=math.log(0.01), zmin=2.5, zmax=4.5):
    rng = np.random.default_rng(seed)
    t = math.tan(math.radians(fov_deg) * 0.5)
    z = rng.uniform(zmin, zmax, P)
    x = rng.uniform(-0.9, 0.9, P) * z * t
    y = rng.uniform(-0.9, 0.9, P) * z * t
    means = np.stack([x, y, z], 1).astype(np.float32)
    scales = np.exp(rng.normal(log_scale_mean, 0.3, (P, 3))).astype(np.float32)
    q = rng.normal(0, 1, (P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0, 1.5, (P, 1))))).astype(np.float32)
    M = (sh_degree + 1) ** 2
    shs = rng.normal(0, 0.1, (P, M, 3))
    shs[:, 0, :] = rng.normal(0, 1.0, (P, 3))
    colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    return dict(means3D=means, scales=scales, rotations=q.astype(np.float32), opacities=opac,
                shs=shs.astype(np.float32), colors=colors, sh_degree=sh_degree)


def loss_targets(W, H, seed=0):
    rng = np.random.default_rng(seed + 1000)
    gt = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    mask = (rng.uniform(0, 1, (1, H, W)) > 0.5).astype(np.float32)
    return gt, mask


def uniform_scene(P, W, H, seed=0, sh_degree=3, **kw):
    cam = make_camera(W, H, 50.0)
    g = uniform_gaussians(P, seed, sh_degree, **kw)
    return cam, g
