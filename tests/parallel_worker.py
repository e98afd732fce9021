"""Rank process of the two-process view-parallel test (tests/test_gpu_parallel_procs.py): started by
mygauhuman_amd.launch.spawn_ranks with the torchrun environment; renders ITS view of the shared scene through
ViewParallelStep(reduce=True) and stores the reduced gradients.

  python -m tests.parallel_worker <out_prefix> <P> <W> <H> <deg> <compact 0|1> [overflow_rank]
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def scene(P, W, H, deg):
    from tests import util
    cam0, g = util.make_scene(P, W, H, 17, deg)
    rng = np.random.default_rng(3)
    gt = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    mask = (rng.uniform(0, 1, (1, H, W)) > 0.5).astype(np.float32)
    return g, gt, mask


def camera_of_rank(W, H, rank, world):
    from mygauhuman_amd import cameras
    return cameras.orbit_camera(W, H, 4.0 * (rank - (world - 1) / 2))


def main():
    out, P, W, H, deg, compact = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6] == "1"
    overflow_rank = int(sys.argv[7]) if len(sys.argv) > 7 else -1
    from mygauhuman_amd import parallel
    from tests import util
    rank, world, local = parallel.init_distributed("cuda")
    assert world > 1 and dist.is_initialized()
    torch.cuda.set_device(local)
    g, gt, mask = scene(P, W, H, deg)
    to = util.to_dev
    params = dict(means3D=to(g["means3D"]), shs=to(g["shs"]), opacities=to(g["opacities"]), scales=to(g["scales"]),
                  rotations=to(g["rotations"]))
    cam = camera_of_rank(W, H, rank, world)
    camd = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
    bg = to(np.array([0.1, 0.2, 0.3], np.float32))
    kw = dict(compact_sh=compact)
    if rank == overflow_rank:
        kw["capacity"] = 64  # far too small: this rank's view cannot be binned
    step = parallel.ViewParallelStep(params, deg, camd, bg, **kw)
    step(camd, bg, to(gt), to(mask), reduce=True)
    res = {k: v.detach().cpu().numpy() for k, v in step.grads.items()}
    res["R"] = np.array([step.session.num_rendered()])
    overflow_seen = 0
    try:
        step.check()
    except parallel.BinningOverflow:
        overflow_seen = 1
        if rank == overflow_rank:
            assert step.session.capacity >= 2 * int(res["R"][0])
        step(camd, bg, to(gt), to(mask), reduce=True)   # the repeated step: every rank renders now
        step.check()
        res.update({"retry_" + k: v.detach().cpu().numpy() for k, v in step.grads.items()})
    res["overflow_seen"] = np.array([overflow_seen])
    res["backend"] = np.array([dist.get_backend()])
    np.savez(f"{out}_rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
