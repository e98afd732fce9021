"""The view-parallel step over render() (reference: train.py:212-224,401-417, scene/gaussian_model.py:764-766): two
FRESH rank processes (gloo, both on device 0) each run parallel.ViewParallelRender on their own ring camera and pose of the shared
articulated model.  Every leaf gradient (the nine model tensors + the parameters of the two motion decoders) must equal the
in-process mean over the two views of plain render() + autograd -- which knows nothing of buckets, sinks or exchanges --, the
replicas must be bit-identical, the densification statistics must be the sums / max over the views, and a rank that cannot bin
its view makes EVERY rank skip the step and raise."""
import sys

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
import os  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(GSR_DIST_BACKEND="gloo", GSR_SINGLE_DEVICE="1")
P, V, W, H = 6000, 1500, 208, 176


def _in_process(motion, world=2):
    """Plain render() + autograd per view, no parallel layer: mean gradients, summed statistics."""
    from mygauhuman_amd import human_synth
    from mygauhuman_amd.gaussian_renderer import render
    from tests.parallel_render_worker import image_weights, loss_of, pipe
    model, body = human_synth.build(P, V, "cuda", seed=0, motion=motion)
    leaves = {n: getattr(model, n) for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_normal",
                                             "_albedo", "_roughness")}
    if motion:
        for mn in ("pose_decoder", "lweight_offset_decoder"):
            leaves.update({f"{mn}.{n}": p for n, p in getattr(model, mn).named_parameters()})
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    acc = {n: torch.zeros_like(t, dtype=torch.float64) for n, t in leaves.items()}
    gn, vis_n, rad = torch.zeros((P, 1), device="cuda"), torch.zeros((P, 1), device="cuda"), torch.zeros((P,), dtype=torch.int32, device="cuda")
    for r in range(world):
        for t in leaves.values():
            t.grad = None
        cam = human_synth.view_camera(body, W, H, r, n_views=8, device="cuda")
        o = render(1, cam, model, pipe(), bg)
        loss_of(o, image_weights(W, H, r, "cuda")).backward()
        for n, t in leaves.items():
            if t.grad is not None:
                acc[n] += t.grad.double()
        vis = o["visibility_filter"]
        gn += torch.norm(o["viewspace_points"].grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
        vis_n += vis[:, None].float()
        rad = torch.maximum(rad, o["radii"])
        assert 0.3 < float(vis.float().mean()) <= 1.0
    return ({n: (a / world).cpu().numpy() for n, a in acc.items()}, gn.cpu().numpy(), vis_n.cpu().numpy(), rad.cpu().numpy())


def _run_ranks(tmp_path, compact, motion, overflow_rank=-1):
    from mygauhuman_amd.launch import spawn_ranks
    prefix = str(tmp_path / "vpr")
    argv = [sys.executable, "-m", "tests.parallel_render_worker", prefix, str(P), str(V), str(W), str(H), "1" if compact else "0",
            "1" if motion else "0"]
    if overflow_rank >= 0:
        argv.append(str(overflow_rank))
    codes = spawn_ranks(argv, 2, env=dict(ENV, PYTHONPATH=ROOT), timeout=600)
    assert codes == [0, 0], codes
    return [dict(np.load(f"{prefix}_rank{r}.npz")) for r in range(2)]


@pytest.mark.parametrize("compact,motion", [(True, True), (False, True), (True, False)], ids=["compact_motion", "plain_motion", "compact_static"])
def test_two_rank_render_step_equals_mean_of_view_gradients(tmp_path, compact, motion):
    want, gn, vis_n, rad = _in_process(motion)
    r0, r1 = _run_ranks(tmp_path, compact, motion)
    assert int(r0["overflow_seen"][0]) == 0 and len(r0["exchange_ms"]) == 1
    for n, w in want.items():
        assert n in r0, n
        if not np.any(w):   # _roughness always (get_roughness reads _albedo, scene/gaussian_model.py:197-199); _albedo: its image is not in this loss
            assert n in ("_roughness", "_albedo") and not np.any(r0[n]), n
        else:
            util.assert_close(f"{n} rank0", r0[n].reshape(w.shape), w, tol=5e-5, max_bad_frac=1e-4, outer_tol=1e-3)
        np.testing.assert_array_equal(r0[n], r1[n], err_msg=f"{n}: replicas differ")
    util.assert_close("stat_grad_norm", r0["stat_grad_norm"], gn, tol=2e-5, max_bad_frac=1e-4, outer_tol=1e-3)
    np.testing.assert_array_equal(r0["stat_visible"], vis_n)
    np.testing.assert_array_equal(r0["max_radii"], rad)
    for k in ("stat_grad_norm", "stat_visible", "max_radii"):
        np.testing.assert_array_equal(r0[k], r1[k])
    if compact:   # 17 + 2 floats per Gaussian in the all-reduce and 7 in the all-gather instead of 65 + 2 (+ the MLPs)
        assert int(r0["payload_bytes"][0]) < 0.5 * (67 * 4 * P)


def test_render_overflow_on_one_rank_skips_the_step_on_every_rank(tmp_path):
    want, _, _, _ = _in_process(True)
    r0, r1 = _run_ranks(tmp_path, True, True, overflow_rank=1)
    for r in (r0, r1):
        assert int(r["overflow_seen"][0]) == 1
        for n in want:
            assert not np.any(r[n]), n   # the skipped step: exact zeros on every replica
    for n, w in want.items():
        if not np.any(w):
            continue
        util.assert_close(f"retry {n}", r0["retry_" + n].reshape(w.shape), w, tol=5e-5, max_bad_frac=1e-4, outer_tol=1e-3)
        np.testing.assert_array_equal(r0["retry_" + n], r1["retry_" + n])


def test_single_process_view_parallel_render_equals_plain_autograd():
    """world = 1: the bucket / sink / compact machinery alone (forced compact mode) against plain render() + autograd."""
    from mygauhuman_amd import human_synth, parallel
    from tests.parallel_render_worker import image_weights, loss_of, pipe
    want, gn, vis_n, rad = _in_process(True, world=1)
    model, body = human_synth.build(P, V, "cuda", seed=0, motion=True)
    cam = human_synth.view_camera(body, W, H, 0, n_views=8, device="cuda")
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    for compact in (True, False):
        step = parallel.ViewParallelRender(model, pipe(), bg, compact_sh=compact)
        assert (step.compact is not None) == compact
        weights = image_weights(W, H, 0, "cuda")
        step(1, cam, lambda o: loss_of(o, weights))
        step.check()
        for n, w in want.items():
            if not np.any(w):
                continue
            util.assert_close(f"{n} compact={compact}", step.leaves[n].grad.cpu().numpy().reshape(w.shape), w, tol=5e-5, max_bad_frac=1e-4,
                              outer_tol=1e-3)
        np.testing.assert_array_equal(step.stat_visible.cpu().numpy(), vis_n)
        np.testing.assert_array_equal(step.max_radii.cpu().numpy(), rad)


def test_a_replaced_leaf_is_followed_and_a_resized_model_is_refused():
    """densify.reset_opacity (train.py:412) installs a NEW nn.Parameter of the same shape for `_opacity`: the step object must reduce
    and hand back the gradient of the live tensor, not of the one it saw at construction; a model whose Gaussian count
    changed must be refused with a message that says what to do."""
    from mygauhuman_amd import human_synth, parallel
    from tests.parallel_render_worker import image_weights, loss_of, pipe
    model, body = human_synth.build(P, V, "cuda", seed=0, motion=True)
    cam = human_synth.view_camera(body, W, H, 0, n_views=8, device="cuda")
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    weights = image_weights(W, H, 0, "cuda")
    for compact in (True, False):
        step = parallel.ViewParallelRender(model, pipe(), bg, compact_sh=compact)
        step(1, cam, lambda o: loss_of(o, weights))
        old = model._opacity
        model._opacity = torch.nn.Parameter((old.detach() * 0.5 - 1.0).clone().requires_grad_(True))   # what reset_opacity does
        step(2, cam, lambda o: loss_of(o, weights))
        step.check()
        assert step.leaves["_opacity"] is model._opacity
        assert model._opacity.grad is not None and model._opacity.grad.data_ptr() == step.bucket["_opacity"].data_ptr()
        # the same frame through plain render() + autograd
        from mygauhuman_amd.gaussian_renderer import render
        for p_ in step.leaves.values():
            p_.grad = None
        got = {n: None for n in step.leaves}
        step(3, cam, lambda o: loss_of(o, weights))
        got = {n: t.grad.detach().clone() for n, t in step.leaves.items()}
        for p_ in step.leaves.values():
            p_.grad = None
        loss_of(render(3, cam, model, pipe(), bg), weights).backward()
        for n, t in step.leaves.items():
            if t.grad is None:
                assert float(got[n].abs().max()) == 0.0, n
                continue
            util.assert_close(f"{n} after the leaf was replaced (compact={compact})", got[n].cpu().numpy(), t.grad.cpu().numpy(), tol=5e-5,
                              max_bad_frac=1e-4, outer_tol=1e-3)
        model._opacity = old
    # a different Gaussian count: refused
    step = parallel.ViewParallelRender(model, pipe(), bg, compact_sh=False)
    keep = model._xyz
    model._xyz = torch.nn.Parameter(keep.detach()[:-1].clone().requires_grad_(True))
    with pytest.raises(RuntimeError, match="build a new ViewParallelRender"):
        step(1, cam, lambda o: loss_of(o, weights))
    model._xyz = keep


def test_bench_render_workload_starts_its_own_ranks():
    """`python bench.py --workload render --gpus 2` (here: two ranks sharing the one GPU over gloo, a reduced Gaussian count):
    one JSON line with the exchange / compute split."""
    import json
    import subprocess
    env = dict(os.environ, GSR_BENCH_P="20000", **ENV)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "render", "--gpus", "2", "--steps", "3", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["exchange_ms"] > 0 and d["step_compute_ms"] > 0 and "REHEARSAL" in d["config"]["workload"]
    assert "_xyz" in d["config"]["leaves"] and any(k.startswith("pose_decoder.") for k in d["config"]["leaves"])


def test_view_parallel_training_loop_keeps_replicas_identical(tmp_path):
    """Eight iterations of the reference's training loop on two views per step (train.py:212-417): ViewParallelRender step ->
    densification statistics -> Adam on the nine parameter groups and the two decoders, with one densify-and-prune in the middle.
    Every rank applies the same reduced gradients and statistics and draws the same split samples, so the two replicas must hold
    the SAME model -- bit for bit, Gaussian count included -- at the end, and the views differ per rank and per step."""
    from mygauhuman_amd.launch import spawn_ranks
    prefix = str(tmp_path / "vpt")
    argv = [sys.executable, "-m", "tests.parallel_train_worker", prefix, str(P), str(V), str(W), str(H), "8", "5"]
    codes = spawn_ranks(argv, 2, env=dict(ENV, PYTHONPATH=ROOT), timeout=900)
    assert codes == [0, 0], codes
    r0, r1 = (dict(np.load(f"{prefix}_rank{r}.npz")) for r in range(2))
    assert r0["counts"][0] == P and r0["counts"][-1] != P, r0["counts"]      # the densify-and-prune really changed the model
    np.testing.assert_array_equal(r0["counts"], r1["counts"])
    assert not np.array_equal(r0["losses"], r1["losses"])                      # the ranks render different views
    for k in r0:
        if k in ("losses",):
            continue
        assert np.all(np.isfinite(r0[k])), k
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=f"{k}: replicas diverged")
    assert float(r0["denom"].max()) >= 2.0   # statistics of BOTH views of a step were counted
