"""Interleaved A/B of tuning knobs inside ONE process (same device, alternating rounds; reports medians).
usage: python tools/sweep.py "bucket_cstride=1" "bucket_cstride=2" ...   (each arg: comma-separated key=value list)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd import _lib, cameras, parallel, synthetic  # noqa: E402


def main():
    configs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
    P, W, H, deg = 200_000, 1024, 1024, 3
    dev = torch.device("cuda", 0)
    g = synthetic.uniform_gaussians(P, seed=0, sh_degree=deg)
    gt, mask = synthetic.loss_targets(W, H, seed=0)
    cam = cameras.make_camera(W, H, 50.0)
    to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    params = dict(means3D=to(g["means3D"]), shs=to(g["shs"]), opacities=to(g["opacities"]), scales=to(g["scales"]),
                  rotations=to(g["rotations"]))
    camd = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
    bg, gt_d, mask_d = to(np.zeros(3, np.float32)), to(gt), to(mask)
    step = parallel.ViewParallelStep(params, deg, camd, bg)
    res = {i: [] for i in range(len(configs))}
    stages = {i: [] for i in range(len(configs))}
    for rnd in range(7):
        for i, c in enumerate(configs):
            for k, v in c.items():
                _lib.set_tuning(k, int(v))
            for _ in range(5):
                step(camd, bg, gt_d, mask_d, reduce=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(60):
                step(camd, bg, gt_d, mask_d, reduce=False)
            torch.cuda.synchronize()
            res[i].append((time.perf_counter() - t0) / 60 * 1e3)
            _lib.profile_enable(_lib.PROF_STAGES)
            for _ in range(5):
                step(camd, bg, gt_d, mask_d, reduce=False)
            torch.cuda.synchronize()
            pr = _lib.profile_read()
            stages[i].append({k: ms / n for k, (ms, n) in pr.items() if n})
            _lib.profile_enable([])
    for i, c in enumerate(configs):
        st = {k: round(float(np.median([s[k] for s in stages[i]])), 4) for k in stages[i][0]}
        print(f"{c}: median {np.median(res[i]):.4f} ms/step (min {min(res[i]):.4f})  stages {st}", flush=True)


if __name__ == "__main__":
    main()
