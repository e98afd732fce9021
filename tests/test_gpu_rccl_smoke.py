"""RCCL itself, as far as a one-GPU box can take it: a fresh process initialises the `nccl` backend (= RCCL on ROCm) with a world
of ONE rank on the device and runs the exact collective calls of the view-parallel exchange -- the construction-time self-test
(all_gather_into_tensor + all_reduce with checked results), the flat-bucket SUM all-reduce and the out-of-place all-gather of a
view block -- on device tensors.  Two ranks cannot share a device under RCCL, so the multi-rank tests of this suite run over gloo;
this one pins that the RCCL code path (library load, communicator creation, the tensor-form all-gather, stream ordering against
the compute stream) works on this image, before the round-end scaling run is its first multi-GPU use."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import os, sys
sys.path.insert(0, os.environ["GSR_ROOT"])
import torch, torch.distributed as dist
from mygauhuman_amd import parallel
from mygauhuman_amd.launch import free_port
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
form = parallel.collective_selftest(dev)
b = parallel.GradientBucket(parallel.gaussian_gradient_shapes(1000, 16, "sh_compact"), dev)
b.flat.copy_(torch.arange(b.flat.numel(), dtype=torch.float32, device=dev))
want = b.flat.clone()
parallel.all_reduce_(b.flat)
ex = parallel.CompactShExchange(1000, 16, dev, posed=True)
ex.mine.copy_(torch.arange(ex.stride, dtype=torch.float32, device=dev))
dist.all_gather_into_tensor(ex.gathered.view(-1), ex.mine)       # what exchange() issues for world > 1
y = (b.flat * 2).sum()                                             # compute-stream work ordered after the collectives
torch.cuda.synchronize()
assert torch.equal(b.flat, want) and torch.equal(ex.gathered[0], ex.mine) and float(y) == float((want * 2).sum())
# the exchange AS ISSUED since round 4: all-gather on a side stream through a communicator of its own, the all-reduce on the compute
# stream at the same time, wait() before the blocks are read -- ordering against compute-stream work on BOTH sides of it, 20 rounds
ex2 = parallel.CompactShExchange(1000, 16, dev, posed=True, side_stream=True)
assert ex2.side is not None and ex2.ag_group is not None
for k in range(20):
    ex2.mine.copy_(torch.arange(ex2.stride, dtype=torch.float32, device=dev) + float(k))   # "the pack": compute-stream work
    ex2.exchange_async(timing=True)
    b.flat.copy_(want + float(k))
    parallel.all_reduce_(b.flat)                                   # travels while the all-gather is in flight
    ex2.wait()
    got = ex2.gathered[0].clone()                                  # compute-stream read after wait()
    torch.cuda.synchronize()
    assert torch.equal(got, torch.arange(ex2.stride, dtype=torch.float32, device=dev) + float(k)), k
    assert torch.equal(b.flat, want + float(k)), k
assert len(ex2.allgather_ms()) == 20
print("RCCL_SMOKE_OK", form)
dist.destroy_process_group()
"""


def test_rccl_backend_single_rank_runs_the_exchange_collectives():
    env = dict(os.environ, GSR_ROOT=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_SMOKE_OK tensor" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
