"""The N > 1 path end to end on the one-GPU box: two FRESH rank processes (gloo, both on device 0) each run
ViewParallelStep(reduce=True) on their own orbit view; every reduced gradient (flat bucket + compact-SH reconstruction) must
equal the in-process mean of the two views' gradients, the replicas must be bit-identical, a rank that cannot bin its view
makes EVERY rank skip the step (zeros) and raise, and `python bench.py --gpus 2` starts its own ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(GSR_DIST_BACKEND="gloo", GSR_SINGLE_DEVICE="1")


def _in_process_mean(P, W, H, deg, world=2):
    from mygauhuman_amd import parallel
    from tests.parallel_worker import camera_of_rank, scene
    g, gt, mask = scene(P, W, H, deg)
    to = util.to_dev
    params = dict(means3D=to(g["means3D"]), shs=to(g["shs"]), opacities=to(g["opacities"]), scales=to(g["scales"]),
                  rotations=to(g["rotations"]))
    bg = to(np.array([0.1, 0.2, 0.3], np.float32))
    acc = None
    for r in range(world):
        cam = camera_of_rank(W, H, r, world)
        camd = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
        step = parallel.ViewParallelStep(params, deg, camd, bg, compact_sh=False)
        step(camd, bg, to(gt), to(mask), reduce=False)
        step.check()
        cur = {k: v.double().cpu().numpy() for k, v in step.grads.items()}
        acc = cur if acc is None else {k: acc[k] + cur[k] for k in acc}
    return {k: v / world for k, v in acc.items()}


def _run_ranks(tmp_path, P, W, H, deg, compact, overflow_rank=-1):
    from mygauhuman_amd.launch import spawn_ranks
    prefix = str(tmp_path / "vp")
    argv = [sys.executable, "-m", "tests.parallel_worker", prefix, str(P), str(W), str(H), str(deg), "1" if compact else "0"]
    if overflow_rank >= 0:
        argv.append(str(overflow_rank))
    codes = spawn_ranks(argv, 2, env=dict(ENV, PYTHONPATH=ROOT), timeout=600)
    assert codes == [0, 0], codes
    return [dict(np.load(f"{prefix}_rank{r}.npz")) for r in range(2)]


@pytest.mark.parametrize("compact", [True, False])
def test_two_rank_step_equals_mean_of_view_gradients(tmp_path, compact):
    P, W, H, deg = 6000, 176, 112, 3
    want = _in_process_mean(P, W, H, deg)
    r0, r1 = _run_ranks(tmp_path, P, W, H, deg, compact)
    assert str(r0["backend"][0]) == "gloo" and int(r0["overflow_seen"][0]) == 0
    assert int(r0["R"][0]) != int(r1["R"][0])  # the ranks really rendered different views
    for k in ("means3D", "sh", "opacity", "scales", "rotations"):
        assert float(np.abs(want[k]).max()) > 0
        util.assert_close(f"{k} rank0", r0[k], want[k], tol=2e-5, max_bad_frac=1e-4)
        np.testing.assert_array_equal(r0[k], r1[k])  # replicas stay bit-identical


def test_overflow_on_one_rank_skips_the_step_on_every_rank(tmp_path):
    P, W, H, deg = 6000, 176, 112, 3
    want = _in_process_mean(P, W, H, deg)
    r0, r1 = _run_ranks(tmp_path, P, W, H, deg, True, overflow_rank=1)
    for r in (r0, r1):
        assert int(r["overflow_seen"][0]) == 1
        for k in ("means3D", "sh", "opacity", "scales", "rotations"):
            assert not np.any(r[k]), k   # the skipped step: exact zeros on every replica, not a biased half-mean
    for k in ("means3D", "sh", "opacity", "scales", "rotations"):
        util.assert_close(f"retry {k}", r0["retry_" + k], want[k], tol=2e-5, max_bad_frac=1e-4)
        np.testing.assert_array_equal(r0["retry_" + k], r1["retry_" + k])


def test_single_process_overflow_is_never_silent():
    """Calibrate on a sparse view, then step on a dense one: the step must raise, regrow and then work."""
    from mygauhuman_amd import cameras, parallel
    P, W, H, deg = 8000, 160, 96, 3
    cam, g = util.make_scene(P, W, H, 3, deg, scale=0.05)
    to = util.to_dev
    params = dict(means3D=to(g["means3D"]), shs=to(g["shs"]), opacities=to(g["opacities"]), scales=to(g["scales"]),
                  rotations=to(g["rotations"]))
    bg = to(np.zeros(3, np.float32))
    dense = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
    far = cameras.make_camera(W, H, 50.0, T=np.array([0.0, 0.0, 40.0]))  # the scene shrinks to a few pixels: few instances
    sparse = dict(far, viewmatrix=to(far["viewmatrix"]), projmatrix=to(far["projmatrix"]), campos=to(far["campos"]))
    rng = np.random.default_rng(0)
    gt, mask = to(rng.uniform(0, 1, (3, H, W)).astype(np.float32)), to((rng.uniform(0, 1, (1, H, W)) > 0.5).astype(np.float32))
    step = parallel.ViewParallelStep(params, deg, sparse, bg, slack=1.0)
    cap0 = step.session.capacity
    step(sparse, bg, gt, mask, reduce=False)
    step.check()                                  # fits
    step(dense, bg, gt, mask, reduce=False)
    assert step.session.num_rendered() > cap0
    assert not np.any(step.grads["means3D"].cpu().numpy())  # nothing was rendered ...
    with pytest.raises(parallel.BinningOverflow):            # ... and the step says so
        step.check()
    assert step.session.capacity > cap0
    step(dense, bg, gt, mask, reduce=False)
    step.check()
    assert float(step.grads["means3D"].abs().max()) > 0
    # the deferred check also fires by itself once the step is max_in_flight calls old
    step2 = parallel.ViewParallelStep(params, deg, sparse, bg, slack=1.0, max_in_flight=1)
    step2(dense, bg, gt, mask, reduce=False)
    with pytest.raises(parallel.BinningOverflow):  # at the next call: the overflowed step is max_in_flight = 1 calls old
        step2(sparse, bg, gt, mask, reduce=False)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: exits 0, one JSON line, n_gpus 2."""
    env = dict(os.environ, **ENV)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and "REHEARSAL" in d["config"]["workload"]
    assert d["roofline"]["kernel"] in ("blend_bwd", "blend_fwd", "binning", "preprocess_fwd", "preprocess_bwd")  # two ranks share one GPU here


def test_bench_under_the_torchrun_launch_line():
    """The launch line of the round-end scaling run (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`), here with N = 2 ranks sharing the one GPU over gloo."""
    from mygauhuman_amd.launch import free_port
    env = dict(os.environ, **ENV)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
