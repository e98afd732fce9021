"""Rank launcher of the view-parallel layer: starts one fresh process per rank with the torch.distributed.run environment.

Imports nothing that loads the HIP library, so a parent process that only launches ranks never touches the GPU (the
parent of `python bench.py --gpus N` must be able to start its ranks and exit with their status)."""
import os
import socket
import subprocess
import time

import torch


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(argv, world, env=None, timeout=None, rehearse_on_one_device=None):
    """Start `world` fresh child processes of `argv` (one rank each: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way
    torch.distributed.run sets them) and wait for them.  The caller must not have touched the GPU if it is going to exit with
    the children's status; children are new processes, never an exec of the caller.  Returns the list of exit codes.

    rehearse_on_one_device: None = decide from the device count (fewer devices than ranks -> every rank on device 0 over
    gloo: GSR_SINGLE_DEVICE=1, GSR_DIST_BACKEND=gloo -- a functional rehearsal, not a scaling measurement)."""
    base = dict(os.environ)
    base.update(env or {})
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(free_port())
    # dmabuf IPC: the host driver of the GPU pool supports no other kind, and RCCL / device-tensor sharing between processes
    # fails with `hipIpcGetMemHandle: invalid argument` in the legacy mode (the pool exports this itself; a rank started from
    # an environment that lost it -- `env -i`, a scheduler -- still gets it)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # read by the HIP runtime of every rank when it starts (graph.py): set here so that no rank depends on its import order
    base.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    if rehearse_on_one_device is None:
        # (device_count() may start the HIP runtime in THIS process -- harmless: the parent launches ranks and exits with their
        # status, it neither captures graphs nor execs)
        rehearse_on_one_device = torch.cuda.device_count() < world
    if rehearse_on_one_device:
        base.setdefault("GSR_SINGLE_DEVICE", "1")
        base.setdefault("GSR_DIST_BACKEND", "gloo")
    procs = []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world))
        procs.append(subprocess.Popen(list(argv), env=e))
    # wait for all; a rank that dies takes the job down (its peers would wait in a collective forever)
    deadline = None if timeout is None else time.monotonic() + timeout
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            return codes
        failed = any(c not in (None, 0) for c in codes)
        if failed or (deadline is not None and time.monotonic() > deadline):
            time.sleep(2.0 if failed else 0.0)  # let the peers print their own error first
            for p in procs:
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() for p in procs]
            if not failed:
                raise subprocess.TimeoutExpired(list(argv), timeout)
            return codes
        time.sleep(0.05)
