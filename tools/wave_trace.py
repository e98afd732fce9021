"""Where does the time of the render() frame's blend kernels go?  (GPU box.)  Registers a trace buffer (gsr_debug_wave_trace), runs ONE
eager forward + backward of the tools/render_bench.py body scene and prints, for the forward and the backward blend kernel: the kernel's
span, how many waves were resident over time, and how a wave's duration relates to the length of its list."""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mygauhuman_amd import _lib  # noqa: E402
from mygauhuman_amd.gaussian_renderer import render  # noqa: E402

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_tuning(k, int(v))


def report(name, rec, length_of):
    rec = rec[rec[:, 0] > 0]
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    n = length_of(rec)
    start = t0.min()
    span = (t1.max() - start) / 100.0  # us (100 MHz ticks)
    dur = (t1 - t0) / 100.0
    print(f"== {name}: {len(rec)} waves, span {span:.1f} us, sum of wave durations {dur.sum() / 1e3:.1f} ms = {dur.sum() / span:.0f} waves resident on average")
    edges = np.linspace(0, span, 11)
    act = [int(((t0 - start) / 100.0 <= x) & ((t1 - start) / 100.0 > x)).sum() if False else int((((t0 - start) / 100.0 <= x) & ((t1 - start) / 100.0 > x)).sum())
           for x in (edges[:-1] + edges[1:]) / 2]
    print("   resident waves at 5 %, 15 %, ... 95 % of the span:", act)
    late = (t0 - start) / 100.0
    print(f"   wave starts: 50 % by {np.percentile(late, 50):.1f} us, 90 % by {np.percentile(late, 90):.1f}, last {late.max():.1f} us")
    order = np.argsort(-dur)[:5]
    print("   longest waves (us, entries, start us):", [(round(float(dur[i]), 1), int(n[i]), round(float(late[i]), 1)) for i in order])
    for lo, hi in ((1, 128), (128, 256), (256, 512), (512, 768), (768, 1024), (1024, 4096)):
        sel = (n >= lo) & (n < hi)
        if sel.sum():
            print(f"   lists {lo:4d}..{hi:4d}: {int(sel.sum()):5d} waves, median {np.median(dur[sel]):6.1f} us = {np.median(dur[sel] / n[sel]) * 1e3:6.0f} ns per entry; "
                  f"started at median {np.median(late[sel]):5.1f} us")


def c3_main():
    """SCENE=c3: the headline workload (200k Gaussians, SH3, 1024^2, alpha-mask loss) through the raw bindings."""
    from mygauhuman_amd import synthetic
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    P, W, H = 200000, 1024, 1024
    cam, g = synthetic.uniform_scene(P, W, H, seed=0, sh_degree=3)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items() if isinstance(v, np.ndarray)}
    view, proj, campos = (torch.from_numpy(cam[k]).cuda() for k in ("viewmatrix", "projmatrix", "campos"))
    bg, e = torch.zeros(3, device="cuda"), torch.empty(0)

    def step():
        o = _C.rasterize_gaussians(bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e, view, proj, cam["tanfovx"],
                                   cam["tanfovy"], H, W, t["shs"], 3, campos, False, False)
        R, color, depth, alpha, radii, gb, bb, ib = o
        dc = torch.sign(color - 0.5) / color.numel()
        _C.rasterize_gaussians_backward(bg, t["means3D"], radii, e, t["scales"], t["rotations"], 1.0, e, view, proj, cam["tanfovx"],
                                        cam["tanfovy"], dc, torch.zeros_like(alpha), 0.2 * (alpha - 0.5) / alpha.numel(), t["shs"], 3, campos,
                                        gb, R, bb, ib, alpha, False)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    words = 2 * 16 * 5120
    buf = torch.zeros(words, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib.gsr_debug_wave_trace(buf.data_ptr(), words), "trace")
    step()
    torch.cuda.synchronize()
    _lib.check(_lib.lib.gsr_debug_wave_trace(None, 0), "trace")
    rec = buf.cpu().numpy().view(np.uint64).reshape(2, -1, 4)
    report("blend forward (C3)", rec[0], lambda r: np.maximum(r[:, 2].astype(np.int64), 1))
    report("blend backward (C3)", rec[1], lambda r: np.maximum((r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64), 1))


def main():
    if os.environ.get("SCENE") == "c3":
        return c3_main()
    import render_bench
    if os.environ.get("SCENE", "render_bench") == "bench":   # the frame of bench.py's render_200k lines
        import bench
        from mygauhuman_amd import human_synth
        wl = bench.RENDER_WL
        model, body = human_synth.build(wl["P"], wl["V"], torch.device("cuda", 0), seed=0)
        cam = human_synth.view_camera(body, wl["W"], wl["H"], 0, n_views=8, device=torch.device("cuda", 0))
        bg = torch.zeros(3, device="cuda")
    else:
        model, cam, bg = render_bench.scene()
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True, separate_feature_passes=False, sync_free_raster=True)

    def step():
        for p in model.parameters():
            p.grad = None
        o = render(1, cam, model, pipe, bg)
        sum(o[k].mean() for k in (render_bench.ALL_KEYS if os.environ.get("KEYS") == "all" else render_bench.PHASE1_KEYS)).backward()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    slots = 4096 + 1024
    words = 2 * 16 * slots
    buf = torch.zeros(words, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib.gsr_debug_wave_trace(buf.data_ptr(), words), "trace")
    step()
    torch.cuda.synchronize()
    _lib.check(_lib.lib.gsr_debug_wave_trace(None, 0), "trace")
    rec = buf.cpu().numpy().view(np.uint64).reshape(2, -1, 4)
    report("blend forward (18 channels)", rec[0], lambda r: np.maximum(r[:, 2].astype(np.int64), 1))
    report("blend backward (features)", rec[1], lambda r: np.maximum((r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64), 1))


if __name__ == "__main__":
    main()
