"""Rank process of the view-parallel render() test (tests/test_gpu_parallel_render.py): started by
mygauhuman_amd.launch.spawn_ranks with the torchrun environment; renders ITS camera / pose of the shared articulated model
through parallel.ViewParallelRender and stores every leaf's reduced gradient and the densification statistics.

  python -m tests.parallel_render_worker <out_prefix> <P> <V> <W> <H> <compact 0|1> <motion 0|1> [overflow_rank]
"""
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KEYS = ("render", "render_alpha", "normal", "render_axis", "albedo")   # train.py:256-286's phase-1 images + one PBR image (_albedo gets a gradient)


def loss_of(out, weights):
    """A fixed linear functional of the four images (seeded weights): every view has a non-trivial, reproducible gradient."""
    return sum((out[k] * w).sum() for k, w in zip(KEYS, weights)) / out["render"].numel()


def image_weights(W, H, view, device):
    g = torch.Generator().manual_seed(500 + view)
    return [torch.randn((1 if k == "render_alpha" else 3, H, W), generator=g).to(device) for k in KEYS]


def pipe():
    return types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)


def main():
    out, P, V, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    compact, motion = sys.argv[6] == "1", sys.argv[7] == "1"
    overflow_rank = int(sys.argv[8]) if len(sys.argv) > 8 else -1
    from mygauhuman_amd import human_synth, parallel
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    rank, world, local = parallel.init_distributed("cuda")
    assert world > 1 and dist.is_initialized()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    model, body = human_synth.build(P, V, dev, seed=0, motion=motion)
    cam = human_synth.view_camera(body, W, H, rank, n_views=8, device=dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    step = parallel.ViewParallelRender(model, pipe(), bg, compact_sh=compact)
    weights = image_weights(W, H, rank, dev)
    if rank == overflow_rank:   # this rank's first frame cannot be binned
        real, calls = _C.AsyncCapacity.capacity, []

        def tiny(Pn, device=None):
            calls.append(Pn)
            return 256 if len(calls) == 1 else real(Pn, device)
        _C.AsyncCapacity.capacity = tiny
    step(1, cam, lambda o: loss_of(o, weights))
    res = {n: t.grad.detach().cpu().numpy().copy() for n, t in step.leaves.items()}
    res["stat_grad_norm"] = step.stat_grad_norm.cpu().numpy().copy()
    res["stat_visible"] = step.stat_visible.cpu().numpy().copy()
    res["max_radii"] = step.max_radii.cpu().numpy().copy()
    overflow_seen = 0
    try:
        step.check()
    except parallel.BinningOverflow:
        overflow_seen = 1
        step(1, cam, lambda o: loss_of(o, weights))   # the repeated step: every rank renders now
        step.check()
        res.update({"retry_" + n: t.grad.detach().cpu().numpy().copy() for n, t in step.leaves.items()})
    res["overflow_seen"] = np.array([overflow_seen])
    res["exchange_ms"] = np.array(step.timer.read_ms())
    res["payload_bytes"] = np.array([step.payload_bytes])
    np.savez(f"{out}_rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
