"""Compact SH-gradient exchange of the view-parallel layer (csrc/sh_exchange.hip), emulated on one GPU: the SH gradient
rebuilt from each view's clamp-masked dL_dRGB and camera position must equal the mean of the views' own dL_dsh."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deg,n_views", [(3, 3), (2, 2), (0, 1), (3, 8)])
def test_compact_sh_exchange_equals_mean_of_view_gradients(deg, n_views):
    from mygauhuman_amd import _lib, cameras, parallel
    P, W, H = 5000, 160, 112
    cam0, g = util.make_scene(P, W, H, 17, deg)
    bg = util.to_dev(np.array([0.1, 0.2, 0.3], np.float32))
    params = dict(means3D=util.to_dev(g["means3D"]), shs=util.to_dev(g["shs"]), opacities=util.to_dev(g["opacities"]),
                  scales=util.to_dev(g["scales"]), rotations=util.to_dev(g["rotations"]))
    rng = np.random.default_rng(3)
    gt = util.to_dev(rng.uniform(0, 1, (3, H, W)).astype(np.float32))
    mask = util.to_dev((rng.uniform(0, 1, (1, H, W)) > 0.5).astype(np.float32))
    M = g["shs"].shape[1]

    def camd(c):
        return dict(c, viewmatrix=util.to_dev(c["viewmatrix"]), projmatrix=util.to_dev(c["projmatrix"]), campos=util.to_dev(c["campos"]))

    cams = [camd(cameras.orbit_camera(W, H, 4.0 * (v - (n_views - 1) / 2)) if n_views > 1 else cam0) for v in range(n_views)]
    step = parallel.ViewParallelStep(params, deg, cams[0], bg, compact_sh=False)
    ex = parallel.CompactShExchange(P, M, "cuda")
    stride = ex.stride
    gathered = torch.zeros((n_views, stride), device="cuda")
    want = torch.zeros((P, M, 3), device="cuda", dtype=torch.float64)
    clamped_any = 0
    for v, c in enumerate(cams):
        step(c, bg, gt, mask, reduce=False)
        want += step.grads["sh"].double()
        ex.pack(step.session, c["campos"])
        gathered[v].copy_(ex.mine)
        clamped_any += int((ex.mine[:P * 3] == 0).sum())
    want /= n_views
    got = torch.empty((P, M, 3), device="cuda")
    _lib.check(_lib.lib.gsr_sh_grad_from_views(P, deg, M, n_views, params["means3D"].data_ptr(), gathered.data_ptr(), stride,
                                               1.0 / n_views, None, got.data_ptr(), torch.cuda.current_stream().cuda_stream), "from_views")
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert scale > 0 and clamped_any > 0
    err = float((got.double() - want).abs().max()) / scale
    assert err < 2e-6, err
    if M > (deg + 1) ** 2:  # inactive bands stay exactly zero
        assert float(got[:, (deg + 1) ** 2:].abs().max()) == 0.0


def test_view_parallel_step_compact_single_process():
    """world size 1 with compact_sh forced on: the exchange degenerates to pack + reconstruct of the own view."""
    from mygauhuman_amd import parallel
    P, W, H = 3000, 96, 64
    cam, g = util.make_scene(P, W, H, 5, 3)
    bg = util.to_dev(np.zeros(3, np.float32))
    params = dict(means3D=util.to_dev(g["means3D"]), shs=util.to_dev(g["shs"]), opacities=util.to_dev(g["opacities"]),
                  scales=util.to_dev(g["scales"]), rotations=util.to_dev(g["rotations"]))
    camd = dict(cam, viewmatrix=util.to_dev(cam["viewmatrix"]), projmatrix=util.to_dev(cam["projmatrix"]), campos=util.to_dev(cam["campos"]))
    rng = np.random.default_rng(1)
    gt = util.to_dev(rng.uniform(0, 1, (3, H, W)).astype(np.float32))
    mask = util.to_dev((rng.uniform(0, 1, (1, H, W)) > 0.5).astype(np.float32))
    plain = parallel.ViewParallelStep(params, 3, camd, bg, compact_sh=False)
    compact = parallel.ViewParallelStep(params, 3, camd, bg, compact_sh=True)
    plain(camd, bg, gt, mask)
    compact(camd, bg, gt, mask)
    assert compact.payload_bytes < plain.payload_bytes / 3
    for k in ("means3D", "opacity", "scales", "rotations"):
        util.assert_close(k, compact.grads[k].cpu().numpy(), plain.grads[k].cpu().numpy(), tol=2e-5, max_bad_frac=1e-4)
    util.assert_close("sh", compact.grads["sh"].cpu().numpy(), plain.grads["sh"].cpu().numpy(), tol=2e-6)
