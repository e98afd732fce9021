"""End-to-end use of the operator API the way train.py uses it: Adam on the raw Gaussian parameters through
GaussianRasterizer + l1 / ssim losses, with densification statistics, a densify-and-prune step and an opacity reset in the
loop.  The loss must go down; every tensor must stay finite."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def test_fit_to_target_image_with_densification():
    from mygauhuman_amd import cameras, densify, loss_utils
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from mygauhuman_amd.scene_model import HumanGaussianModel
    from mygauhuman_amd.synthetic import uniform_gaussians
    torch.manual_seed(0)
    W = H = 128
    cam = cameras.make_camera(W, H, 50.0)
    bg = torch.zeros(3, device="cuda")
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg,
                                       scale_modifier=1.0, viewmatrix=util.to_dev(cam["viewmatrix"]),
                                       projmatrix=util.to_dev(cam["projmatrix"]), sh_degree=0, campos=util.to_dev(cam["campos"]),
                                       prefiltered=False, debug=False)
    rast = GaussianRasterizer(rs)

    def render(m):
        screen = torch.zeros_like(m.get_xyz, requires_grad=True)
        screen.retain_grad()
        color, radii, depth, alpha = rast(means3D=m.get_xyz, means2D=screen, opacities=m.get_opacity, shs=m.get_features,
                                          scales=m.get_scaling, rotations=m.get_rotation)
        return color, radii, screen

    target_model = HumanGaussianModel.from_arrays(uniform_gaussians(1500, seed=3, sh_degree=0, log_scale_mean=float(np.log(0.05))), 0)
    with torch.no_grad():
        target = render(target_model)[0].clamp(0, 1)
    m = HumanGaussianModel.from_arrays(uniform_gaussians(600, seed=9, sh_degree=0, log_scale_mean=float(np.log(0.05))), 0)
    densify.training_setup(m, dict(xyz=2e-3, f_dc=2e-2, f_rest=1e-3, opacity=5e-2, scaling=5e-3, rotation=1e-3), percent_dense=0.01)
    losses, counts = [], []
    for it in range(1, 241):
        color, radii, screen = render(m)
        loss = 0.8 * loss_utils.l1_loss(color, target) + 0.2 * (1.0 - loss_utils.ssim(color[None], target[None]))
        loss.backward()
        with torch.no_grad():
            vis = radii > 0
            m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis].float())
            densify.add_densification_stats(m, screen, vis)
            if it % 60 == 0 and it < 200:
                densify.densify_and_prune(m, 2e-5, 0.005, 4.0, 0)
            if it == 150:
                densify.reset_opacity(m)
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
        counts.append(m.get_xyz.shape[0])
    assert all(np.isfinite(losses))
    assert np.mean(losses[-10:]) < 0.6 * np.mean(losses[:10]), (losses[:3], losses[-3:])
    assert counts[-1] != counts[0]                       # densification changed the set
    for p in m.parameters():
        assert torch.isfinite(p).all()
    assert m.optimizer.state[m._xyz]["exp_avg"].shape == m._xyz.shape
