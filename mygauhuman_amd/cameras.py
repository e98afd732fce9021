"""Camera matrix conventions of the hot path (host side, numpy).

Mirrors the matrices the reference hands to the rasterizer:
  world_view_transform = getWorld2View2(R, T)^T           utils/graphics_utils.py:38-49, scene/cameras.py:63
  projection_matrix    = getProjectionMatrix_refine(K)^T  utils/graphics_utils.py:82-102, scene/cameras.py:65
  full_proj_transform  = world_view_transform @ projection_matrix            scene/cameras.py:66
  camera_center        = inverse(world_view_transform)[3, :3]                scene/cameras.py:67
Memory order is what the kernels index (row-vector convention: x-row = m[0],m[4],m[8],m[12]).
"""
import math

import numpy as np


def world2view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """getWorld2View2: R is stored transposed (camera-to-world rotation), t is the W2C translation."""
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = np.asarray(R, np.float64).T
    Rt[:3, 3] = np.asarray(t, np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, np.float64)) * scale
    return np.linalg.inv(C2W).astype(np.float32)


def projection_from_K(K, H, W, znear=0.001, zfar=1000.0):
    """getProjectionMatrix_refine (fp32 arithmetic like the torch original)."""
    K = np.asarray(K, np.float32)
    fx, fy, cx, cy, s = K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1]
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = np.float32(2) * fx / np.float32(W)
    P[0, 1] = np.float32(2) * s / np.float32(W)
    P[0, 2] = np.float32(-1) + np.float32(2) * (cx / np.float32(W))
    P[1, 1] = np.float32(2) * fy / np.float32(H)
    P[1, 2] = np.float32(-1) + np.float32(2) * (cy / np.float32(H))
    P[2, 2] = np.float32((zfar + znear) / (zfar - znear))
    P[2, 3] = np.float32(-1.0 * 2 * zfar * znear / (zfar - znear))
    P[3, 2] = np.float32(1.0)
    return P


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def make_camera(W, H, fov_deg=50.0, R=None, T=None):
    """Pinhole camera with fx = fy = W / (2 tan(fov/2)), principal point at the image centre."""
    R = np.eye(3) if R is None else np.asarray(R, np.float64)
    T = np.zeros(3) if T is None else np.asarray(T, np.float64)
    f = W / (2.0 * math.tan(math.radians(fov_deg) * 0.5))
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
    view_T = world2view(R, T).T.copy()  # world_view_transform
    proj_T = projection_from_K(K, H, W).T.copy()
    full = (view_T.astype(np.float32) @ proj_T.astype(np.float32)).astype(np.float32)
    campos = np.linalg.inv(view_T.astype(np.float64))[3, :3].astype(np.float32)
    fovx, fovy = focal2fov(float(K[0, 0]), W), focal2fov(float(K[1, 1]), H)
    return dict(W=W, H=H, K=K, viewmatrix=np.ascontiguousarray(view_T, np.float32),
                projmatrix=np.ascontiguousarray(full, np.float32), campos=campos,
                tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5), FoVx=fovx, FoVy=fovy)


def look_at_camera(W, H, eye, target, fov_deg=50.0):
    """Camera at `eye` looking at `target`, image y pointing down (-world y is up)."""
    eye, target = np.asarray(eye, np.float64), np.asarray(target, np.float64)
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    down_hint = np.array([0.0, 1.0, 0.0])  # camera y (image down) = world +y, as in the identity camera
    right = np.cross(down_hint, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rc2w = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world
    Tw2c = -Rc2w.T @ eye
    return make_camera(W, H, fov_deg, R=Rc2w, T=Tw2c)


def ring_camera(W, H, k, n, radius=3.0, fov_deg=50.0):
    """k-th of n cameras on a horizontal ring of `radius` looking at the origin (config 4 views)."""
    ang = 2.0 * math.pi * k / n
    return look_at_camera(W, H, [radius * math.sin(ang), 0.0, -radius * math.cos(ang)], [0.0, 0.0, 0.0], fov_deg)


def orbit_camera(W, H, yaw_deg, center=(0.0, 0.0, 3.5), fov_deg=50.0):
    """Camera orbiting `center` at the distance of the origin, yaw 0 = the identity view of the S-uniform scene."""
    c = np.asarray(center, np.float64)
    d = np.linalg.norm(c)
    a = math.radians(yaw_deg)
    eye = c + d * np.array([-math.sin(a), 0.0, -math.cos(a)])
    return look_at_camera(W, H, eye, c, fov_deg)


class ViewCamera:
    """The fields of scene.cameras.Camera (scene/cameras.py:17-74) that render() reads, as device tensors."""

    def __init__(self, cam, device, smpl_param=None, big_pose_smpl_param=None, big_pose_world_vertex=None, occlusion=None):
        import torch
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
        self.image_width, self.image_height = cam["W"], cam["H"]
        self.FoVx, self.FoVy = cam["FoVx"], cam["FoVy"]
        self.world_view_transform = t(cam["viewmatrix"])
        self.full_proj_transform = t(cam["projmatrix"])
        self.camera_center = t(cam["campos"])
        self.smpl_param = smpl_param
        self.big_pose_smpl_param = big_pose_smpl_param
        self.big_pose_world_vertex = big_pose_world_vertex
        self.occlusion = occlusion
