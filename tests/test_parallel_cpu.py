"""World-size-2 gloo tests (CPU) of the view-parallel layer: flat gradient bucket all-reduce, densification
statistics, view assignment."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mygauhuman_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_distributed("cpu")
    assert (r, w) == (rank, world)
    P, M = 257, 16
    b = parallel.GradientBucket(parallel.gaussian_gradient_shapes(P, M), "cpu")
    gen = torch.Generator().manual_seed(100 + rank)
    for v in b.views.values():
        v.copy_(torch.randn(v.shape, generator=gen))
    mine = {k: v.clone() for k, v in b.views.items()}
    b.all_reduce_mean()
    # densification statistics
    acc = torch.full((P, 1), float(rank + 1))
    den = torch.full((P, 1), float(rank))
    rad = torch.arange(P, dtype=torch.int32) * (rank + 1)
    parallel.all_reduce_densify_stats(acc, den, rad)
    views = [parallel.view_for_step(s, rank, world) for s in range(3)]
    # compact SH exchange: every rank ends up with every rank's (masked dL_dRGB | campos) block, in rank order
    ex = parallel.CompactShExchange(P, M, "cpu")
    ex.mine[:P * 3].copy_(torch.arange(P * 3, dtype=torch.float32) + 1000.0 * rank)
    ex.mine[P * 3:P * 3 + 3].copy_(torch.tensor([rank + 0.25, rank + 0.5, rank + 0.75]))
    ex.exchange()
    assert ex.world == world and ex.gathered.shape == (world, ex.stride) and ex.stride % 64 == 0
    for r2 in range(world):
        assert torch.equal(ex.gathered[r2, :P * 3], torch.arange(P * 3, dtype=torch.float32) + 1000.0 * r2)
        assert torch.equal(ex.gathered[r2, P * 3:P * 3 + 3], torch.tensor([r2 + 0.25, r2 + 0.5, r2 + 0.75]))
    assert parallel.gaussian_gradient_shapes(P, M, "sh_compact").keys() == {"means3D", "opacity", "scales", "rotations"}
    # the construction-time self-test of the two collectives (checks their RESULTS) and the posed block layout of render()
    assert parallel.collective_selftest(torch.device("cpu")) in ("tensor", "list")
    exp = parallel.CompactShExchange(P, 16, "cpu", posed=True)
    assert (exp.means_off, exp.cam_off, exp.radii_off) == (3 * P, 6 * P, 6 * P + 4) and exp.stride % 64 == 0 and exp.stride >= 7 * P + 4
    exp.mine.copy_(torch.arange(exp.stride, dtype=torch.float32) + 10000.0 * rank)
    exp.mine[exp.radii_off:exp.radii_off + P].copy_(torch.arange(P, dtype=torch.float32) * (rank + 1))
    exp.exchange()
    for r2 in range(world):
        assert torch.equal(exp.gathered[r2, :exp.radii_off], torch.arange(exp.radii_off, dtype=torch.float32) + 10000.0 * r2)
    assert torch.equal(exp.max_radii(), (torch.arange(P) * world).to(torch.int32))
    assert exp.mine.data_ptr() != exp.gathered.data_ptr()   # send and receive memory never alias
    q.put((rank, {k: v.numpy() for k, v in mine.items()}, {k: v.clone().numpy() for k, v in b.views.items()},
           acc.numpy(), den.numpy(), rad.numpy(), views))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_allreduce_and_densify_stats_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, mine0, red0, acc0, den0, rad0, v0), (_, mine1, red1, acc1, den1, rad1, v1) = res
    for k in mine0:
        want = (mine0[k] + mine1[k]) / 2
        np.testing.assert_allclose(red0[k], want, rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(red0[k], red1[k])  # replicas stay bit-identical
    assert np.all(acc0 == 3.0) and np.all(den0 == 1.0) and np.array_equal(acc0, acc1)
    np.testing.assert_array_equal(rad0, np.arange(257, dtype=np.int32) * 2)
    assert v0 == [0, 2, 4] and v1 == [1, 3, 5]


def test_bucket_layout_single_process():
    b = parallel.GradientBucket(parallel.gaussian_gradient_shapes(100, 16), "cpu")
    assert b["sh"].shape == (100, 16, 3) and b["rotations"].shape == (100, 4)
    for name, (off, n, shape) in b.slices.items():
        assert off % 64 == 0 and b[name].is_contiguous() and b[name].data_ptr() == b.flat.data_ptr() + 4 * off
    b["means3D"].fill_(2.0)
    assert float(b.flat[:300].sum()) == 600.0
    b.all_reduce_mean()  # no process group: no-op
    assert float(b.flat[:300].sum()) == 600.0
    shapes = parallel.gaussian_gradient_shapes(10, 0, mode="precomp")
    assert list(shapes) == ["means3D", "colors", "opacity", "cov3D"]
