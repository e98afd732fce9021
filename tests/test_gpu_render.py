"""GPU test of gaussian_renderer.render(): the whole articulated path (LBS deform -> covariance -> SH -> rasterizer)
against the composition of the CPU oracle pieces, result-dict contract, and gradient flow to every parameter group."""
import os
import types

import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_lbs import BIG_POSE, PARENTS, make_smpl

pytestmark = pytest.mark.gpu


def _human_scene(oracle, P=4000, V=1200, W=160, H=128, seed=0, motion=False):
    from mygauhuman_amd import cameras
    from mygauhuman_amd.scene_model import HumanGaussianModel
    rng = np.random.default_rng(seed)
    m = make_smpl(V, seed)
    d = util.to_dev
    smpl = dict(v_template=d(m["v_template"]), shapedirs=d(m["shapedirs"]), posedirs=d(m["posedirs"]),
                J_regressor=d(m["J_regressor"]), weights=d(m["weights"]),
                kintree_table=torch.from_numpy(np.stack([PARENTS, np.arange(24)]).astype(np.int64)).cuda())
    betas, pose = rng.normal(0, 0.5, 10).astype(np.float32), rng.normal(0, 0.15, 72).astype(np.float32)
    Rw, Th = np.eye(3, dtype=np.float32), np.array([0.0, 0.0, 0.0], np.float32)
    big_verts = m["v_template"].copy()
    pts = (big_verts[rng.integers(0, V, P)] + rng.normal(0, 0.01, (P, 3))).astype(np.float32)
    g = dict(means3D=pts, scales=np.exp(rng.normal(np.log(0.02), 0.3, (P, 3))).astype(np.float32),
             rotations=rng.normal(0, 1, (P, 4)).astype(np.float32),
             opacities=(1 / (1 + np.exp(-rng.normal(0, 1.5, (P, 1))))).astype(np.float32),
             shs=np.concatenate([rng.normal(0, 1, (P, 1, 3)), rng.normal(0, 0.1, (P, 15, 3))], 1).astype(np.float32))
    model = HumanGaussianModel.from_arrays(g, 3, smpl=smpl, motion_offset_flag=motion, seed=seed)
    cam_np = cameras.look_at_camera(W, H, eye=[0.3, -0.1, -2.6], target=[0.0, -0.1, 0.0], fov_deg=50.0)
    smpl_param = dict(poses=d(pose[None]), shapes=d(betas[None]), R=d(Rw), Th=d(Th[None]))
    big_param = dict(poses=d(BIG_POSE[None]), shapes=d(np.zeros((1, 10), np.float32)), R=d(np.eye(3, dtype=np.float32)),
                     Th=d(np.zeros((1, 3), np.float32)))
    cam = cameras.ViewCamera(cam_np, "cuda", smpl_param, big_param, d(big_verts))
    return types.SimpleNamespace(m=m, g=g, model=model, cam=cam, cam_np=cam_np, betas=betas, pose=pose, R=Rw, Th=Th,
                                 big_verts=big_verts)


def _oracle_render(oracle, s, bg):
    """Compose the oracle pieces: nearest vertex -> LBS -> covariance -> SH colours -> rasterizer."""
    m, g = s.m, s.g
    rot_big, rot_pose = oracle.rodrigues(BIG_POSE), oracle.rodrigues(s.pose)
    A_big, _ = oracle.joint_transforms(m, np.zeros(10, np.float32), rot_big)
    A_pose, _ = oracle.joint_transforms(m, s.betas, rot_pose)
    ids = oracle.nearest_vertex(g["means3D"], s.big_verts)
    nrm = s.model.get_normal.detach().cpu().numpy()
    o = oracle.lbs_deform(g["means3D"], nrm, ids, m["weights"], A_big, A_pose, oracle.pose_offsets(m["posedirs"], rot_big),
                          oracle.shape_offsets(m["shapedirs"], s.betas), oracle.pose_offsets(m["posedirs"], rot_pose), s.R, s.Th)
    q = g["rotations"] / np.linalg.norm(g["rotations"], axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                  1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                  1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3).astype(np.float64)
    L = R * g["scales"][:, None, :].astype(np.float64)
    T = o["transforms"].astype(np.float64)
    cov = T @ (L @ L.transpose(0, 2, 1)) @ T.transpose(0, 2, 1)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1).astype(np.float32)
    from mygauhuman_amd.sh_utils import eval_sh
    dirs = o["world_src"] - s.cam_np["campos"]
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    rgb = eval_sh(3, torch.from_numpy(np.ascontiguousarray(g["shs"].transpose(0, 2, 1))), torch.from_numpy(dirs)).numpy()
    colors = np.maximum(rgb + 0.5, 0).astype(np.float32)
    c = s.cam_np
    return oracle.rasterize_forward(o["world_src"], g["opacities"], c["viewmatrix"], c["projmatrix"], c["campos"], c["W"], c["H"],
                                    c["tanfovx"], c["tanfovy"], bg, cov3D_precomp=cov6, colors_precomp=colors), o


def test_render_matches_oracle_composition_and_contract(oracle):
    from mygauhuman_amd.gaussian_renderer import RESULT_KEYS, render
    s = _human_scene(oracle)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    out = render(1, s.cam, s.model, pipe, util.to_dev(bg))
    assert tuple(out.keys()) == RESULT_KEYS
    H, W, P = s.cam_np["H"], s.cam_np["W"], s.g["means3D"].shape[0]
    for k in ("render", "normal", "albedo", "occlusion", "roughness", "world_normal", "render_axis"):
        assert out[k].shape == (3, H, W) and torch.isfinite(out[k]).all(), k
    assert out["render_depth"].shape == (1, H, W) and out["render_alpha"].shape == (1, H, W)
    assert out["radii"].shape == (P,) and out["visibility_filter"].dtype == torch.bool
    assert out["transforms"].shape == (1, P, 3, 3) and out["translation"] is None and out["correct_Rs"] is None
    ref, lbs_o = _oracle_render(oracle, s, bg)
    vis_frac = float((ref["pre"]["radii"] > 0).mean())
    assert vis_frac > 0.9
    # radii may differ where fp32 covariance roundings (torch vs float64 numpy) move ceil(3 sigma) across an integer
    assert float((out["radii"].cpu().numpy() == ref["pre"]["radii"]).mean()) > 0.995
    np.testing.assert_allclose(out["transforms"][0].detach().cpu().numpy(), lbs_o["transforms"], rtol=1e-4, atol=1e-4)
    diff = np.abs(out["render"].detach().cpu().numpy() - ref["img"]["color"])
    assert np.percentile(diff, 99.9) < 2e-3 and diff.mean() < 5e-5, (diff.max(), diff.mean())
    adiff = np.abs(out["render_alpha"].detach().cpu().numpy() - ref["img"]["alpha"])
    assert np.percentile(adiff, 99.9) < 2e-3 and adiff.mean() < 5e-5
    # gradients reach every parameter group (xyz through LBS + projection, SH, opacity, scale/rotation through the
    # python covariance, normals/albedo through the feature passes)
    loss = sum(out[k].mean() for k in ("render", "normal", "albedo", "occlusion", "roughness", "world_normal", "render_axis"))
    (loss + out["render_alpha"].mean()).backward()
    for name, p in zip(("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "normal", "albedo"), s.model.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0, name
    assert out["viewspace_points"].grad is not None and float(out["viewspace_points"].grad.abs().sum()) > 0


def test_render_kernel_sh_and_cov_modes_agree(oracle):
    """compute_cov3D_python / convert_SHs_python = False route scale/rotation and SHs through the rasterizer kernels.
    With identity LBS transforms both settings describe the same image."""
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle, seed=3)
    # rest pose = big pose, zero shape -> transforms = I, so the in-kernel covariance (no LBS transform) is comparable
    s.cam.smpl_param = dict(s.cam.big_pose_smpl_param)
    bg = util.to_dev(np.zeros(3, np.float32))
    a = render(1, s.cam, s.model, types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True), bg)
    b = render(1, s.cam, s.model, types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False), bg)
    np.testing.assert_allclose(a["transforms"][0].detach().cpu().numpy(), np.broadcast_to(np.eye(3), (4000, 3, 3)), atol=2e-4)
    d = (a["render"] - b["render"]).abs().detach()
    assert float(d.mean()) < 1e-4 and float(torch.quantile(d.flatten()[::7], 0.999)) < 5e-3


def test_render_with_cached_transforms_and_motion_decoders(oracle):
    """motion_offset_flag=True: pose / LBS-weight decoders feed the deform; cached transforms skip it (render.py:169-195)."""
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle, seed=5, motion=True)
    P = s.g["means3D"].shape[0]

    class PoseDec(torch.nn.Module):
        def forward(self, posevec):
            return {"Rs": torch.eye(3, device=posevec.device)[None, None].repeat(1, 23, 1, 1)}

    class WDec(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1, 24, 1, device="cuda"))

        def forward(self, pts):
            return self.w.expand(1, 24, pts.shape[1])

    s.model.pose_decoder, s.model.lweight_offset_decoder = PoseDec(), WDec()
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    bg = util.to_dev(np.zeros(3, np.float32))
    out = render(1, s.cam, s.model, pipe, bg, return_smpl_rot=True)
    assert out["translation"].shape == (1, P, 3) and out["correct_Rs"].shape == (1, 23, 3, 3)
    out["render"].mean().backward()
    assert s.model.lweight_offset_decoder.w.grad is not None
    cached = render(1, s.cam, s.model, pipe, bg, transforms=out["transforms"].detach(), translation=out["translation"].detach())
    # same image up to cut-off flips of a few pixels (the cached path recomputes the means as transforms . p + translation)
    d = (cached["render"] - out["render"]).abs().detach().flatten()
    assert float(d.mean()) < 1e-5 and float((d > 2e-4).float().mean()) < 1e-3


def test_render_pipe_knobs_agree(oracle):
    """pipe.sync_free_raster = False (the reference's blocking read of num_rendered) and override_color: same images (to fp32
    rounding) and gradients as the default (sync-free) path."""
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle)
    bg = util.to_dev(np.array([0.3, 0.1, 0.2], np.float32))
    keys = ("render", "normal", "albedo", "occlusion", "roughness", "world_normal", "render_axis", "render_alpha", "render_depth")

    def run(**kw):
        for p in s.model.parameters():
            p.grad = None
        extra = {k: kw.pop(k) for k in ("override_color",) if k in kw}
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, **kw)
        o = render(1, s.cam, s.model, pipe, bg, **extra)
        sum(o[k].mean() for k in keys).backward()
        return {k: o[k].detach().clone() for k in keys}, [None if p.grad is None else p.grad.detach().clone() for p in s.model.parameters()]

    base, gbase = run()
    for variant in (dict(sync_free_raster=False),):  # the default is the sync-free entry; the blocking one must agree
        out, grads = run(**variant)
        for k in keys:
            d = (out[k] - base[k]).abs()
            assert float(d.mean()) < 2e-5 and float((d > 2e-3).float().mean()) < 1e-3, (variant, k, float(d.max()))
        for ga, gb in zip(grads, gbase):
            if gb is None:
                assert ga is None
                continue
            scale = float(gb.abs().max()) + 1e-12
            assert float((ga - gb).abs().mean()) / scale < 1e-4, variant
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    _C.AsyncCapacity.check_all()
    col = torch.rand((s.g["means3D"].shape[0], 3), device="cuda")
    out, grads = run(override_color=col)
    assert float((out["render"] - base["render"]).abs().max()) > 1e-3            # different colours ...
    assert float((out["normal"] - base["normal"]).abs().max()) < 1e-5             # ... same geometry and feature images
    assert grads[1] is None or float(grads[1].abs().sum()) == 0.0                  # SH coefficients unused


def test_pose_refinement_gradient_hip_chain_equals_torch_chain(oracle):
    """A trainable pose-refinement module (correct_Rs with a parameter): the gradient that reaches it through render() ->
    rasterizer -> attributes -> LBS (dA_pose partials, pose-offset GEMV backward) -> pose kernel must equal the one through the
    torch formulation of the pose chain (tests/torch_reference.py, patched in for lbs.smpl_pose_transforms)."""
    from mygauhuman_amd import lbs
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle, seed=7, motion=True)

    class PoseDec(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(3)
            self.delta = torch.nn.Parameter(0.02 * torch.randn((23, 3, 3), generator=g).cuda())

        def forward(self, posevec):
            return {"Rs": (torch.eye(3, device="cuda")[None] + self.delta)[None]}

    class WDec(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1, 24, 1, device="cuda"))

        def forward(self, pts):
            return self.w.expand(1, 24, pts.shape[1])

    dec, wdec = PoseDec(), WDec()
    s.model.pose_decoder, s.model.lweight_offset_decoder = dec, wdec
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    bg = util.to_dev(np.zeros(3, np.float32))
    w_img = torch.rand((3, s.cam_np["H"], s.cam_np["W"]), device="cuda")
    grads = {}
    from tests.torch_reference import smpl_pose_transforms_torch
    hip_chain = lbs.smpl_pose_transforms
    for chain in ("hip", "torch"):
        lbs.smpl_pose_transforms = hip_chain if chain == "hip" else smpl_pose_transforms_torch
        try:
            dec.delta.grad = None
            wdec.w.grad = None
            out = render(1, s.cam, s.model, pipe, bg)
            ((out["render"] * w_img).sum() + out["normal"].mean()).backward()
            grads[chain] = (dec.delta.grad.clone(), wdec.w.grad.clone())
        finally:
            lbs.smpl_pose_transforms = hip_chain
    for a, b in zip(grads["hip"], grads["torch"]):
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) < 2e-3 * scale, (float((a - b).abs().max()), scale)


def test_render_training_loop_with_densification(oracle):
    """render() in a short Adam loop with densification statistics, a densify-and-prune (SMPL-distance prior included) and an
    opacity reset between frames: shapes follow the changing Gaussian count, everything stays finite, the loss goes down."""
    from mygauhuman_amd import densify, loss_utils
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle, seed=9)
    model = s.model
    densify.training_setup(model, dict(xyz=2e-4, f_dc=5e-3, f_rest=2.5e-4, opacity=0.05, scaling=5e-3, rotation=1e-3, normal=1e-3,
                                       albedo=0.02, roughness=0.02))
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    bg = util.to_dev(np.zeros(3, np.float32))
    with torch.no_grad():
        target = render(1, s.cam, model, pipe, bg)["render"].clone()
        model._features_dc.add_(0.5 * torch.randn_like(model._features_dc))   # perturb the colours, then fit them back
    verts = s.cam.big_pose_world_vertex
    losses, counts = [], []
    for it in range(1, 61):
        o = render(it, s.cam, model, pipe, bg)
        loss = loss_utils.l1_loss(o["render"], target) + 0.2 * (1.0 - loss_utils.ssim(o["render"][None], target[None]))
        loss.backward()
        with torch.no_grad():
            densify.update_max_radii(model, o["radii"], o["visibility_filter"])
            densify.add_densification_stats(model, o["viewspace_points"], o["visibility_filter"])
            if it == 30:
                densify.densify_and_prune(model, 1e-7, 0.005, 2.0, 20, t_vertices=verts)
            if it == 45:
                densify.reset_opacity(model)
        model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
        counts.append(model.get_xyz.shape[0])
        assert o["radii"].shape[0] == counts[-1] or it == 30
    assert all(np.isfinite(losses)) and np.mean(losses[25:29]) < 0.7 * np.mean(losses[:3])
    assert counts[-1] != counts[0]
    for p in model.parameters():
        assert torch.isfinite(p).all()


def test_render_step_as_one_graph_equals_eager(oracle):
    """graph.GraphedFrame: render() forward + loss + backward recorded into one hipGraph; a replay gives the eager step's image
    bits and gradients, and follows in-place updates of its static inputs (a new pose)."""
    from mygauhuman_amd.gaussian_renderer import render
    from mygauhuman_amd.graph import GraphedFrame
    s = _human_scene(oracle, seed=11)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    bg = util.to_dev(np.array([0.2, 0.3, 0.1], np.float32))
    params = [p for p in s.model.parameters()]
    keys = ("render", "render_alpha", "normal", "render_axis")

    def step():
        o = render(1, s.cam, s.model, pipe, bg)
        sum(o[k].mean() for k in keys).backward()
        return o

    def eager():
        for p in params:
            p.grad = None
        o = step()
        return o["render"].detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in params]

    frame = GraphedFrame(step, warmup=3, zero_grads=params)
    for trial in range(2):
        if trial == 1:  # a different pose through the same graph: update the static input in place
            s.cam.smpl_param["poses"].add_(0.05 * torch.randn_like(s.cam.smpl_param["poses"]))
        img_e, grads_e = eager()
        out = frame.replay()   # re-attaches the .grad tensors the captured backward writes
        torch.cuda.synchronize()
        frame.check()
        assert torch.equal(out["render"].detach(), img_e), trial
        for p, ge in zip(params, grads_e):
            if ge is None:
                continue
            scale = float(ge.abs().max()) + 1e-20
            assert float((p.grad - ge).abs().max()) / scale < 2e-5, trial
    # gradients of the captured backward live in the graph's pool: read them through the tensors captured at record time
    assert all(torch.isfinite(v).all() for v in out.values() if isinstance(v, torch.Tensor) and v.is_floating_point())


def test_render_with_fused_activations_equals_property_getters(oracle):
    """render() reads the model through ONE activation kernel (HumanGaussianModel.frame_activations) and the two SH tensors in
    place when the model offers them; a model that only has the reference's property getters (util.GetterOnlyModel) goes
    through the torch ops: same images, same gradients."""
    from mygauhuman_amd.gaussian_renderer import render
    outs, grads = {}, {}
    for prop in (False, True):
        s = _human_scene(oracle, seed=5)
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
        o = render(1, s.cam, util.GetterOnlyModel(s.model) if prop else s.model, pipe, util.to_dev(np.array([0.1, 0.2, 0.3], np.float32)))
        keys = ("render", "normal", "albedo", "occlusion", "roughness", "world_normal", "render_axis", "render_alpha")
        sum(o[k].mean() * (i + 1) for i, k in enumerate(keys)).backward()
        outs[prop] = {k: o[k].detach().cpu().numpy() for k in keys + ("render_depth",)}
        grads[prop] = [p.grad.cpu().numpy() for p in s.model.parameters()]
    for k in outs[False]:
        np.testing.assert_allclose(outs[False][k], outs[True][k], atol=3e-5, err_msg=k)
    for ga, gb in zip(grads[False], grads[True]):
        util.assert_close("render grads", ga, gb, tol=1e-4, max_bad_frac=2e-4)


def test_gradients_joined_in_kernel_equal_autograd_accumulation(oracle, monkeypatch):
    """mygauhuman_amd.gradlink: the rasterizer parks its position gradient for the attribute kernel's backward, that one parks the
    raw-quaternion gradient for the activations' backward, and albedo == roughness gets one summed gradient -- three add kernels of
    autograd less per frame.  a + b in a kernel or in autograd is the same fp32 sum: every parameter gradient equals that of the
    frame rendered with GSR_GRAD_LINK off to the run-to-run noise of the blend backward's float atomics (1e-5 of the tensor's
    scale; measured 7e-7), and nothing stays parked."""
    import mygauhuman_amd.gaussian_renderer as gr
    from mygauhuman_amd import gradlink
    grads, links = {}, []
    real = gradlink.FrameLink

    class Spy(real):
        __slots__ = ()

        def __init__(self):
            super().__init__()
            links.append(self)
    monkeypatch.setattr(gradlink, "FrameLink", Spy)
    for on in (False, True):
        monkeypatch.setattr(gr, "GRAD_LINK", on)
        s = _human_scene(oracle, seed=7)
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
        o = gr.render(1, s.cam, s.model, pipe, util.to_dev(np.array([0.1, 0.2, 0.3], np.float32)))
        keys = ("render", "normal", "albedo", "occlusion", "roughness", "world_normal", "render_axis", "render_alpha")
        sum(o[k].mean() * (i + 1) for i, k in enumerate(keys)).backward()
        grads[on] = [None if p.grad is None else p.grad.cpu().numpy() for p in s.model.parameters()]
    assert len(links) == 1 and links[0].attr_means_ptr is not None and links[0].act_rot_in_ptr is not None   # the linked frame
    assert links[0].means_grad is None and links[0].rot_grad is None                                          # nothing left parked
    for ga, gb in zip(grads[False], grads[True]):
        assert (ga is None) == (gb is None)
        if ga is not None:
            assert np.isfinite(gb).all() and np.abs(gb).max() > 0
            util.assert_close("gradient joined in kernel", gb, ga, tol=1e-5, max_bad_frac=0.0)


def _chain64(s, leaf, ids, dec=None):
    """The reference's per-frame chain from the model's leaf parameters to the rasterizer's inputs, in float64 torch on the CPU
    (scene/gaussian_model.py:157-199 activations, :768-872 LBS with the nearest-vertex ids given, :35-42 covariance,
    gaussian_renderer/__init__.py:128-198 colours and feature colours) -- the differentiable half of the oracle composition.
    leaf: dict of float64 leaf tensors (requires_grad); dec: optional dict(delta [23,3,3], w [24]) of the motion decoders."""
    from tests.test_gpu_lbs import _torch_deform
    from tests.torch_reference import frame_attributes_torch, smpl_pose_transforms_torch
    t64 = lambda a: torch.as_tensor(np.asarray(a, np.float64))  # noqa: E731
    m = s.m
    smpl = dict(v_template=t64(m["v_template"]), shapedirs=t64(m["shapedirs"]), posedirs=t64(m["posedirs"]),
                J_regressor=t64(m["J_regressor"]), weights=t64(m["weights"]),
                kintree_table=torch.from_numpy(np.stack([PARENTS, np.arange(24)]).astype(np.int64)))
    V = m["v_template"].shape[0]
    big = dict(poses=t64(BIG_POSE[None]), shapes=t64(np.zeros((1, 10))), R=t64(np.eye(3)), Th=t64(np.zeros((1, 3))))
    tgt = dict(poses=t64(s.pose[None]), shapes=t64(s.betas[None]), R=t64(s.R), Th=t64(s.Th[None]))
    correct_Rs = None if dec is None else (torch.eye(3, dtype=torch.float64)[None] + dec["delta"])[None]
    A_big, rot_big, _ = smpl_pose_transforms_torch(smpl, big)
    A_pose, rot_pose, _ = smpl_pose_transforms_torch(smpl, tgt, correct_Rs)
    ident = torch.eye(3, dtype=torch.float64)
    pd = smpl["posedirs"].reshape(V * 3, -1)
    off_big = (pd @ (rot_big[0, 1:] - ident).reshape(-1)).view(V, 3)
    off_pose = (pd @ (rot_pose[0, 1:] - ident).reshape(-1)).view(V, 3)
    off_shape = (smpl["shapedirs"] @ tgt["shapes"].reshape(-1, 1)).squeeze(-1)
    opacity, albedo = torch.sigmoid(leaf["opacity"]), torch.sigmoid(leaf["albedo"])
    scaling = torch.exp(leaf["scaling"])
    rot_n = leaf["rotation"] / leaf["rotation"].norm(dim=1, keepdim=True)
    normal = leaf["normal"] / leaf["normal"].norm(dim=1, keepdim=True)
    P = leaf["xyz"].shape[0]
    loff = None if dec is None else dec["w"][None].expand(P, 24)
    world, transforms, world_normal = _torch_deform(leaf["xyz"], normal, loff, A_big[0], A_pose[0], off_big, off_shape, off_pose,
                                                    tgt["R"], tgt["Th"].reshape(3), torch.from_numpy(ids.astype(np.int64)),
                                                    smpl["weights"])
    shs = torch.cat((leaf["f_dc"], leaf["f_rest"]), dim=1)
    c = s.cam_np
    cov6, colors, features = frame_attributes_torch(world, transforms, world_normal, scaling, 1.0, leaf["rotation"], rot_n, albedo,
                                                    albedo, opacity.repeat(1, 3), shs, 3, t64(c["campos"]), t64(c["viewmatrix"]))
    return dict(means3D=world, cov6=cov6, colors=colors, features=features, opacity=opacity, transforms=transforms)


@pytest.mark.parametrize("motion", [False, True], ids=["static_weights", "motion_decoders"])
def test_render_features_and_parameter_gradients_match_oracle_composition(oracle, motion):
    """render() end to end against the oracle composition, all seven images and EVERY parameter gradient (VERDICT r2 #2d;
    gaussian_renderer/__init__.py:203-272):
      (A) the pre-raster chain -- activations, LBS deform, covariance, SH colour, six feature colour sets -- HIP fp32 vs the
          float64 torch restatement of the reference's op chain (nearest-vertex ids from the oracle);
      (B) the seven images + alpha + depth vs SEVEN oracle rasterizer passes (precomputed-colour mode, like the reference's
          feature passes) over the bit-identical per-Gaussian inputs, 1e-4 off the fragile pixels;
      (C) d(loss)/d(leaf) for _xyz, _features_dc, _features_rest, _scaling, _rotation, _opacity, _normal, _albedo (and the two
          decoder parameters) and the screen-space gradient vs  oracle.rasterize_backward (summed over the seven passes)
          chained through float64 autograd of (A), at north_star's 1e-4 (rounds 2-3 held these to 3e-4 / 2e-4; measured in round 4:
          they pass at 1e-4)."""
    from mygauhuman_amd import lbs
    from mygauhuman_amd.attributes import frame_attributes
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle, seed=21, motion=motion)
    model, c = s.model, s.cam_np
    H, W, P = c["H"], c["W"], s.g["means3D"].shape[0]
    dec_mods = None
    if motion:
        class PoseDec(torch.nn.Module):
            def __init__(self):
                super().__init__()
                g = torch.Generator().manual_seed(3)
                self.delta = torch.nn.Parameter(0.02 * torch.randn((23, 3, 3), generator=g).cuda())

            def forward(self, posevec):
                return {"Rs": (torch.eye(3, device="cuda")[None] + self.delta)[None]}

        class WDec(torch.nn.Module):
            def __init__(self):
                super().__init__()
                g = torch.Generator().manual_seed(4)
                self.w = torch.nn.Parameter(0.3 * torch.randn((1, 24, 1), generator=g).cuda())

            def forward(self, pts):
                return self.w.expand(1, 24, pts.shape[1])

        dec_mods = (PoseDec(), WDec())
        model.pose_decoder, model.lweight_offset_decoder = dec_mods
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    out = render(1, s.cam, model, pipe, util.to_dev(bg))

    # ---- the HIP path's own per-Gaussian rasterizer inputs (the same deterministic kernels render() just ran)
    with torch.no_grad():
        act = model.frame_activations()
        cr = lw = None
        if motion:
            cr = dec_mods[0](s.cam.smpl_param["poses"][:, 3:])["Rs"]
            lw = dec_mods[1](model.get_xyz[None]).permute(0, 2, 1)
        _, world, _, tf, _, wn = lbs.coarse_deform_c2source(model.SMPL_NEUTRAL, model.get_xyz[None], s.cam.smpl_param,
                                                            s.cam.big_pose_smpl_param, s.cam.big_pose_world_vertex[None],
                                                            lbs_weights=lw, correct_Rs=cr, normals=act.normal[None], lean=True)
        cov_h, col_h, feat_h = frame_attributes(world.reshape(-1, 3), tf.reshape(-1, 3, 3), wn.reshape(-1, 3), act.scaling, 1.0,
                                                model._rotation, act.rotation, act.albedo, act.roughness, act.occlusion,
                                                (model._features_dc, model._features_rest), 3, s.cam.camera_center,
                                                s.cam.world_view_transform)
    hip = dict(means3D=world.reshape(-1, 3).cpu().numpy(), cov6=cov_h.cpu().numpy(), colors=col_h.cpu().numpy(),
               features=feat_h.cpu().numpy(), opacity=act.opacity.cpu().numpy())
    assert torch.equal(out["transforms"].detach(), tf)

    # ---- (A) float64 chain
    ids = oracle.nearest_vertex(s.g["means3D"], s.big_verts)
    names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "normal", "albedo")
    leaf = {n: p.detach().cpu().double().requires_grad_(True) for n, p in zip(names, model.parameters())}
    dec64 = None
    if motion:
        dec64 = dict(delta=dec_mods[0].delta.detach().cpu().double().requires_grad_(True),
                     w=dec_mods[1].w.detach().cpu().double().reshape(24).requires_grad_(True))
    ch = _chain64(s, leaf, ids, dec64)
    for k in ("means3D", "cov6", "colors", "features", "opacity"):
        util.assert_close(f"chain {k}", hip[k], ch[k].detach().numpy().reshape(hip[k].shape), tol=2e-5)

    # ---- (B) seven oracle passes over the HIP path's own inputs
    sets = [hip["colors"]] + [np.ascontiguousarray(hip["features"][:, 3 * k:3 * k + 3]) for k in range(6)]
    keys = ("render", "normal", "world_normal", "albedo", "occlusion", "roughness", "render_axis")
    refs = [oracle.rasterize_forward(hip["means3D"], hip["opacity"], c["viewmatrix"], c["projmatrix"], c["campos"], W, H,
                                     c["tanfovx"], c["tanfovy"], bg, cov3D_precomp=hip["cov6"], colors_precomp=cs) for cs in sets]
    solid = refs[0]["img"]["fragile"] == 0
    assert solid.mean() > 0.99 and float((refs[0]["pre"]["radii"] > 0).mean()) > 0.9
    np.testing.assert_array_equal(out["radii"].cpu().numpy(), refs[0]["pre"]["radii"])
    m3 = np.broadcast_to(solid, (3, H, W))
    for k, r in zip(keys, refs):
        util.assert_close(k, out[k].detach().cpu().numpy(), r["img"]["color"], mask=m3)
    util.assert_close("render_alpha", out["render_alpha"].detach().cpu().numpy(), refs[0]["img"]["alpha"], mask=solid[None])
    util.assert_close("render_depth", out["render_depth"].detach().cpu().numpy(), refs[0]["img"]["depth"], mask=solid[None])

    # ---- (C) gradients: random image weights (zero on fragile pixels), all seven images + alpha + depth live
    rng = np.random.default_rng(5)
    Wk = [(rng.normal(0, 1, (3, H, W)) * solid).astype(np.float32) for _ in keys]
    Wa, Wd = ((rng.normal(0, 1, (1, H, W)) * solid).astype(np.float32) for _ in range(2))
    loss = sum((out[k] * util.to_dev(w)).sum() for k, w in zip(keys, Wk))
    loss = loss + (out["render_alpha"] * util.to_dev(Wa)).sum() + (out["render_depth"] * util.to_dev(Wd)).sum()
    loss.backward()
    zero1 = np.zeros((1, H, W), np.float32)
    g_mean, g_cov, g_op, g_m2d, g_sets = 0.0, 0.0, 0.0, 0.0, []
    for k, (r, w) in enumerate(zip(refs, Wk)):
        b = oracle.rasterize_backward(r, w, Wd if k == 0 else zero1, Wa if k == 0 else zero1)
        g_mean, g_cov = g_mean + b["dL_dmeans3D"].astype(np.float64), g_cov + b["dL_dcov3D"].astype(np.float64)
        g_op, g_m2d = g_op + b["dL_dopacity"].astype(np.float64), g_m2d + b["dL_dmean2D"].astype(np.float64)
        g_sets.append(b["dL_dcolors"].astype(np.float64))
    util.assert_close("viewspace_points.grad", out["viewspace_points"].grad.cpu().numpy(), g_m2d.reshape(P, 3),
                      tol=float(os.environ.get("GSR_RENDER_GRAD_TOL", 1e-4)), max_bad_frac=2e-4, outer_tol=2e-3)
    t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))  # noqa: E731
    outs = [ch["means3D"], ch["cov6"], ch["colors"], ch["features"], ch["opacity"]]
    gouts = [t64(g_mean), t64(g_cov), t64(g_sets[0]), t64(np.concatenate(g_sets[1:], axis=1)), t64(g_op.reshape(P, 1))]
    inputs = list(leaf.values()) + ([dec64["delta"], dec64["w"]] if motion else [])
    want = torch.autograd.grad(outs, inputs, gouts)
    got = [p.grad for p in model.parameters()] + ([dec_mods[0].delta.grad, dec_mods[1].w.grad.reshape(24)] if motion else [])
    for n, g, wnt in zip(names + (("pose_decoder.delta", "lweight_offset_decoder.w") if motion else ()), got, want):
        assert g is not None and float(wnt.abs().max()) > 0, n
        util.assert_close(f"d{n}", g.cpu().numpy().reshape(wnt.shape), wnt.numpy(), tol=float(os.environ.get("GSR_RENDER_GRAD_TOL", 1e-4)),
                          max_bad_frac=2e-4, outer_tol=3e-3)


def test_forward_only_render_raises_on_the_overflowing_call(oracle, monkeypatch):
    """under no_grad (render.py-style evaluation) nobody would examine a deferred overflow flag -- a frame whose
    binning capacity is too small must raise on the SAME render() call, not hand back a background-only image; the capacity
    has been raised by then, so the retry renders."""
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle, seed=13)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    bg = util.to_dev(np.array([0.2, 0.3, 0.1], np.float32))
    with torch.no_grad():
        good = render(1, s.cam, s.model, pipe, bg)["render"].clone()
    _C.AsyncCapacity.check_all()
    real = _C.AsyncCapacity.capacity
    calls = []

    def tiny(P, device=None):
        calls.append(P)
        return 512 if len(calls) == 1 else real(P, device)
    monkeypatch.setattr(_C.AsyncCapacity, "capacity", tiny)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="exceeded the binning capacity"):
            render(1, s.cam, s.model, pipe, bg)
        again = render(1, s.cam, s.model, pipe, bg)["render"]
    assert torch.equal(again, good)
    # with a backward to come the check stays deferred (the host does not wait inside forward): the raise comes from backward()
    calls.clear()
    out = render(1, s.cam, s.model, pipe, bg)
    with pytest.raises(RuntimeError, match="exceeded the binning capacity"):
        out["render"].mean().backward()
    _C.AsyncCapacity.check_all()


@pytest.mark.parametrize("extra_term", [False, True], ids=["phase1_only", "plus_other_terms"])
def test_fused_phase1_loss_equals_the_torch_loss_and_its_autograd(oracle, extra_term):
    """render(fused_loss=Phase1Loss) against the SAME loss written with torch ops the way train.py:261-265 writes it (boolean-mask
    indexing, utils/loss_utils.py:20-24 means): the value, and the gradient of every leaf and of the screen-space points -- with
    and without further loss terms on the images (whose gradients arrive through autograd and are added in the kernel's prologue).
    The torch path itself is pinned against the oracle composition by the test above."""
    from mygauhuman_amd.diff_gaussian_rasterization._C import Phase1Loss
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle, seed=31)
    model, c = s.model, s.cam_np
    H, W = c["H"], c["W"]
    rng = np.random.default_rng(8)
    d = util.to_dev
    gt_image, gt_normal = d(rng.uniform(0, 1, (3, H, W)).astype(np.float32)), d(rng.uniform(0, 1, (3, H, W)).astype(np.float32))
    bkgd = d((rng.uniform(0, 1, (1, H, W)) > 0.4).astype(np.float32))
    bound_np = np.zeros((1, H, W), np.float32)
    bound_np[:, H // 6: H - H // 8, W // 5: W - W // 7] = 1.0
    bound_np *= rng.uniform(0, 1, (1, H, W)) > 0.1   # a ragged mask, not a rectangle
    bound = d(bound_np)
    bg = d(np.array([0.1, 0.2, 0.3], np.float32))
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    Wx = d(rng.normal(0, 1, (3, H, W)).astype(np.float32))

    def others(out):   # stand-ins for SSIM / LPIPS / TV: terms that put their own gradients on image, normal, alpha and depth
        if not extra_term:
            return 0.0
        return (0.3 * (out["render"] * Wx).mean() + 0.2 * (out["normal"] ** 2).mean() + 0.1 * out["render_alpha"].mean()
                + 0.05 * out["render_depth"].mean() + 0.07 * (out["albedo"] * Wx).mean())

    def l1(a, b):
        return torch.abs(a - b).mean()

    params = list(model.parameters())
    # ---- torch ops, as the reference writes the loss
    out = render(1, s.cam, model, pipe, bg)
    bm = bound[0] == 1
    Ll1 = l1(out["render"].permute(1, 2, 0)[bm], gt_image.permute(1, 2, 0)[bm])
    mask_loss = ((out["render_alpha"][bound == 1] - bkgd[bound == 1]) ** 2).mean()
    normal_loss = l1(out["normal"].permute(1, 2, 0)[bm], gt_normal.permute(1, 2, 0)[bm])
    axis_loss = l1(out["render_axis"].permute(1, 2, 0)[bm], gt_normal.permute(1, 2, 0)[bm])
    want_loss = 1.0 * Ll1 + 0.1 * mask_loss + normal_loss + 1.0 * axis_loss
    (want_loss + others(out)).backward()
    want = [p.grad.detach().clone() for p in params] + [out["viewspace_points"].grad.detach().clone()]
    for p in params:
        p.grad = None
    # ---- fused
    spec = Phase1Loss(gt_image, gt_normal, bkgd, bound)
    out2 = render(1, s.cam, model, pipe, bg, fused_loss=spec)
    assert out2["loss"].dim() == 0
    np.testing.assert_allclose(float(out2["loss"].detach()), float(want_loss.detach()), rtol=2e-6)
    (out2["loss"] + others(out2)).backward()
    got = [p.grad for p in params] + [out2["viewspace_points"].grad]
    names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "normal", "albedo", "viewspace")
    for n, g, w in zip(names, got, want):
        if float(w.abs().max()) == 0.0:   # (albedo does not enter the phase-1 loss by itself)
            assert n == "albedo" and not extra_term and (g is None or float(g.abs().max()) == 0.0), n
            continue
        assert g is not None, n
        util.assert_close(f"fused loss d{n}", g.cpu().numpy(), w.cpu().numpy(), tol=2e-5, max_bad_frac=1e-5, outer_tol=2e-4)
    # an upstream factor on the fused loss (loss scaling) scales its gradients, not the other terms'
    for p in params:
        p.grad = None
    out3 = render(1, s.cam, model, pipe, bg, fused_loss=spec)
    (2.5 * out3["loss"]).backward()
    for p in params:
        p.grad = None if p.grad is None else p.grad / 2.5
    if not extra_term:
        for n, p, w in zip(names, params, want):
            if float(w.abs().max()) == 0.0:
                continue
            util.assert_close(f"scaled fused loss d{n}", p.grad.cpu().numpy(), w.cpu().numpy(), tol=2e-5, max_bad_frac=1e-5, outer_tol=2e-4)


def test_fused_phase1_loss_with_an_empty_bound_mask_is_zero_and_finite(oracle):
    from mygauhuman_amd.diff_gaussian_rasterization._C import Phase1Loss
    from mygauhuman_amd.gaussian_renderer import render
    s = _human_scene(oracle, seed=32)
    c = s.cam_np
    H, W = c["H"], c["W"]
    z3, z1 = torch.zeros((3, H, W), device="cuda"), torch.zeros((1, H, W), device="cuda")
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)
    out = render(1, s.cam, s.model, pipe, util.to_dev(np.zeros(3, np.float32)), fused_loss=Phase1Loss(z3, z3, z1, z1))
    assert float(out["loss"]) == 0.0
    out["loss"].backward()
    for p in s.model.parameters():
        assert p.grad is None or (bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) == 0.0)
