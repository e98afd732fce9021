"""Quick stage timing probe (not part of the product): C2/C3-like scenes, forward / backward per tuning knob."""
import argparse
import json
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mygauhuman_amd import _lib, synthetic  # noqa: E402
from mygauhuman_amd.diff_gaussian_rasterization import _C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=200000)
    ap.add_argument("--W", type=int, default=1024)
    ap.add_argument("--H", type=int, default=1024)
    ap.add_argument("--deg", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--waves", type=str, default="1,2,4")
    ap.add_argument("--binning", type=int, default=-1)
    a = ap.parse_args()
    dev = "cuda"
    if a.binning >= 0:
        _lib.check(_lib.lib.gsr_set_binning_mode(a.binning), "binning")
    cam, g = synthetic.uniform_scene(a.P, a.W, a.H, seed=0, sh_degree=a.deg)
    gt, mask = synthetic.loss_targets(a.W, a.H)
    t = {k: torch.from_numpy(v).to(dev) for k, v in g.items() if isinstance(v, np.ndarray)}
    view, proj, campos = (torch.from_numpy(cam[k]).to(dev) for k in ("viewmatrix", "projmatrix", "campos"))
    bg = torch.zeros(3, device=dev)
    gt, mask = torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev)
    e = torch.empty(0)

    def fwd():
        return _C.rasterize_gaussians(bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e, view, proj,
                                      cam["tanfovx"], cam["tanfovy"], a.H, a.W, t["shs"], a.deg, campos, False, False)

    def bwd(o, dc, dd, da):
        R, color, depth, alpha, radii, gb, bb, ib = o
        return _C.rasterize_gaussians_backward(bg, t["means3D"], radii, e, t["scales"], t["rotations"], 1.0, e, view, proj,
                                               cam["tanfovx"], cam["tanfovy"], dc, dd, da, t["shs"], a.deg, campos, gb, R,
                                               bb, ib, alpha, False)

    def timeit(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    res = {}
    for w in [int(x) for x in a.waves.split(",")]:
        _lib.set_tuning("blend_fwd_waves", w)
        _lib.set_tuning("blend_bwd_waves", w)
        o = fwd()
        R = o[0]
        color, alpha = o[1], o[3]
        dc = torch.sign(color - gt) / color.numel()
        da = 0.2 * (alpha - mask) / alpha.numel()
        dd = torch.zeros_like(alpha)
        bwd(o, dc, dd, da)
        f_ms = timeit(fwd, a.iters)
        b_ms = timeit(lambda: bwd(o, dc, dd, da), a.iters)
        res[f"waves{w}"] = dict(R=R, fwd_ms=round(f_ms, 3), bwd_ms=round(b_ms, 3))
        print(json.dumps({f"waves{w}": res[f"waves{w}"]}), flush=True)
    ncon = _C.query_state("N_CONTRIB", a.P, o[0], a.W, a.H, o[5], o[6], o[7]).float()
    rng = _C.query_state("RANGES", a.P, o[0], a.W, a.H, o[5], o[6], o[7]).float()
    print(json.dumps(dict(mean_list=float((rng[:, 1] - rng[:, 0]).mean()), max_list=float((rng[:, 1] - rng[:, 0]).max()),
                          mean_ncontrib=float(ncon.mean()), visible=int((o[4] > 0).sum()))), flush=True)


if __name__ == "__main__":
    main()
