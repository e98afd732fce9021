#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Gaussian-splat hot path.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path over one synthetic camera view: rasterizer forward (preprocess -> binning ->
blend), the alpha-mask loss gradient (L1(color, gt) + 0.1 MSE(alpha, mask), train.py:261-262), rasterizer backward
(blend backward -> backward preprocess) and, for N > 1, the RCCL all-reduce of the flat Gaussian-gradient bucket.
Workload at N = 1: BASELINE.json configs[2] ("C3"): 200k Gaussians, SH degree 3, 1024x1024, fp32, seeded synthetic
scene of SURVEY.md §8(d) (S-uniform).  For N > 1 every rank renders its own view of the same replica (weak scaling).

Rank 0 prints ONE JSON line: frames/s (whole job), the roofline of the dominant kernel (HIP events recorded on the
launch stream inside the timed region) and a CPU baseline (the C oracle of the same workload on the host cores, plus
the reference's CPU LBS+project path restated in C).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md); measured copy ceiling is ~6300

WORKLOADS = {
    "C2": dict(P=50_000, W=512, H=512, deg=0, backward=False, desc="50k Gaussians, SH0, 512x512, forward only"),
    "C3": dict(P=200_000, W=1024, H=1024, deg=3, backward=True, desc="200k Gaussians, SH3, 1024x1024, fwd+bwd, alpha-mask loss"),
    "C5": dict(P=500_000, W=1024, H=1024, deg=3, backward=True, log_scale=math.log(0.005), sh_half=True,
               desc="500k Gaussians, SH3 stored as fp16, 1024x1024, fwd+bwd (scale 0.005)"),
}


def stage_bytes(P, R, HW, M, backward=True):
    """Algorithmic (compulsory) bytes per stage, SURVEY.md §8(d): every array counted once per stage that must
    produce or consume it, the sort as one read + one write of its 12-byte pairs, atomics as one write of the
    final per-Gaussian gradient."""
    b = dict(preprocess_fwd=P * (119 + 12 * M), scan=P * 8, binning=P * 20 + R * 44, blend_fwd=R * 44 + HW * 28)
    if backward:
        b.update(blend_bwd=R * 44 + HW * 28 + P * 44, preprocess_bwd=P * (175 + 24 * M))
    return b


def synthetic_smpl(V=6890, seed=0):
    rng = np.random.default_rng(seed)
    vt = rng.uniform(-1, 1, (V, 3)).astype(np.float32) * np.array([0.9, 0.9, 0.15], np.float32)
    sd = rng.normal(0, 0.01, (V, 3, 10)).astype(np.float32)
    pd = rng.normal(0, 0.001, (207, V * 3)).astype(np.float32)
    J = rng.uniform(0, 1, (24, V)).astype(np.float32)
    J /= J.sum(1, keepdims=True)
    w = rng.uniform(0, 1, (V, 24)).astype(np.float32) ** 4
    w /= w.sum(1, keepdims=True)
    parents = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int32)
    return dict(v_template=vt, shapedirs=sd, posedirs=pd, J_regressor=J, weights=w.astype(np.float32), parents=parents)


def cpu_baseline(wl, cam, g, gt, mask, bg):
    """Time the CPU oracle (a port of the reference algorithm, oracle/) on the host: one full frame of the same
    workload (bounded sample), plus the reference's CPU LBS + project path (smplx/lbs.py + geom_transform_points)."""
    from oracle import oracle as orc
    orc.build()
    threads = max(1, min(16, os.cpu_count() or 1))
    orc.set_threads(threads)
    kw = dict(scales=g["scales"], rotations=g["rotations"], shs=g["shs"], degree=g["sh_degree"], scale_modifier=1.0)
    t0 = time.perf_counter()
    fwd = orc.rasterize_forward(g["means3D"], g["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                                cam["W"], cam["H"], cam["tanfovx"], cam["tanfovy"], bg, **kw)
    t_f = time.perf_counter() - t0
    t_b = 0.0
    if wl["backward"]:
        color, alpha = fwd["img"]["color"], fwd["img"]["alpha"]
        dc = (np.sign(color - gt) / color.size).astype(np.float32)
        da = (0.2 * (alpha - mask) / alpha.size).astype(np.float32)
        t0 = time.perf_counter()
        orc.rasterize_backward(fwd, dc, np.zeros_like(alpha), da)
        t_b = time.perf_counter() - t0
    # LBS + project: the reference's CPU path
    m = synthetic_smpl()
    rng = np.random.default_rng(1)
    betas, pose = rng.normal(0, 1, 10).astype(np.float32), rng.normal(0, 0.2, 72).astype(np.float32)
    orc.set_threads(1)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        verts, _, _, _ = orc.smpl_lbs(betas, pose, m["v_template"], m["shapedirs"], m["posedirs"], m["J_regressor"],
                                      m["parents"], m["weights"])
        orc.project(verts, cam["projmatrix"])
        ts.append(time.perf_counter() - t0)
    lbs_ms = float(np.median(ts[2:]) * 1e3)
    return dict(value=round(1.0 / (t_f + t_b), 4), unit="frames/s", cores=threads, kind="port",
                sample=f"1 frame of the same workload ({wl['desc']}) through oracle/gsr_oracle.c with {threads} OpenMP threads: "
                       f"forward {t_f:.2f} s, backward {t_b:.2f} s",
                lbs_project_ms=round(lbs_ms, 3), lbs_project_cores=1,
                lbs_project_sample="SMPL lbs() of 6,890 verts + projection (oracle/lbs_oracle.c, restates smplx/lbs.py:156-252 "
                                   "+ utils/graphics_utils.py:22-29), median of 10 calls, 1 thread",
                host_cpus=os.cpu_count())


PMC_KERNEL = {"blend_bwd": "blend_backward_kernel<1, 0, 0>", "blend_fwd": "blend_forward_kernel<1, 0>"}


def list_stats(session, P, W, H):
    """Per-tile list statistics of the last frame: instances kept after the exact tile culling, mean list
    length over non-empty tiles, fraction of pixels whose blend stopped early (transmittance cut-off before the end of the list)."""
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    q = lambda what: _C.query_state(what, P, session.capacity, W, H, session.geom, session.bin, session.img)  # noqa: E731
    ranges = q("RANGES").long()
    lens = ranges[:, 1] - ranges[:, 0]
    gx = (W + 15) // 16
    ncon, T = q("N_CONTRIB").long(), q("FINAL_T")
    ys, xs = torch.meshgrid(torch.arange(H, device=T.device), torch.arange(W, device=T.device), indexing="ij")
    tile_len = lens[(ys // 16) * gx + xs // 16]
    early = (ncon < tile_len) & (T < 1e-3)
    ne = lens > 0
    return {"instances_after_tile_cull": int(lens.sum()), "mean_list_length": round(float(lens[ne].float().mean()), 1) if bool(ne.any()) else 0.0,
            "max_list_length": int(lens.max()), "pixels_stopped_early_frac": round(float(early.float().mean()), 4)}


def measured_copy_gbs(dev):
    """Stream-copy ceiling of this GPU (read + write bytes per second of a 1 GiB device-to-device copy), SURVEY.md §8(d)."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return round(5 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def pmc_traffic(stage, workload):
    """Memory-side bytes per launch of the dominant kernel.  PMC passes cannot run inside the bench, so the figure comes from
    the newest committed summary profiles/*_pmc.csv (tools/profile_round.sh: separate rocprofv3 --pmc passes of THIS command
    for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled, the gfx950 correction of MI355X_MICROARCH.md) -- C3 only; else null."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.csv")))
    if workload != "C3" or not files or stage not in PMC_KERNEL:
        return {"traffic": None}
    for r in csv.DictReader(open(files[-1])):
        if r["kernel"] == PMC_KERNEL[stage]:
            extra = {}
            try:  # vector-ALU occupancy of the kernel: SQ_ACTIVE_INST_VALU counts quad-cycles over all SIMDs, SQ_BUSY_CYCLES
                # cycles summed over the 32 shader engines -> busy SIMD-cycles / (kernel cycles x 1024 SIMDs)
                busy = 4.0 * float(r["SQ_ACTIVE_INST_VALU_per_launch"]) / (float(r["SQ_BUSY_CYCLES_per_launch"]) / 32.0 * 1024.0)
                extra = {"valu_busy_frac_pmc": round(min(busy, 1.0), 3)}
            except (KeyError, ZeroDivisionError, ValueError):
                pass
            return {**extra, "traffic": int(float(r["fetch_bytes_x2"]) + float(r["write_bytes"])),
                    "traffic_source": "profiles/" + os.path.basename(files[-1]) + " (2 x FETCH_SIZE + WRITE_SIZE per launch; "
                                      "WRITE_SIZE counts every float-atomic lane as 4 B)"}
    return {"traffic": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--binning", type=int, default=-1, help="0 global radix, 1 tile bucket (default: library default)")
    ap.add_argument("--tune", action="append", default=[], help="key=value tuning knob (gsr_set_tuning), repeatable")
    a = ap.parse_args()

    from mygauhuman_amd import _lib, cameras, parallel, synthetic
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    rank, world, local = parallel.init_distributed("cuda")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.binning >= 0:
        _lib.check(_lib.lib.gsr_set_binning_mode(a.binning), "gsr_set_binning_mode")
    for kv in a.tune:
        k, v = kv.split("=")
        _lib.set_tuning(k, int(v))

    wl = WORKLOADS[a.workload]
    P, W, H, deg = wl["P"], wl["W"], wl["H"], wl["deg"]
    M = (deg + 1) ** 2
    g = synthetic.uniform_gaussians(P, seed=0, sh_degree=deg, log_scale_mean=wl.get("log_scale", math.log(0.01)))
    gt, mask = synthetic.loss_targets(W, H, seed=0)
    # one view per rank: yaw the S-uniform camera about the scene centre (rank 0 of a 1-GPU run = identity view)
    yaw = (rank - (world - 1) / 2.0) * 3.0
    cam = cameras.orbit_camera(W, H, yaw) if world > 1 else cameras.make_camera(W, H, 50.0)
    bg_np = np.zeros(3, np.float32)

    to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    params = dict(means3D=to(g["means3D"]), shs=to(g["shs"]), opacities=to(g["opacities"]), scales=to(g["scales"]),
                  rotations=to(g["rotations"]))
    if wl.get("sh_half"):
        params["shs"] = params["shs"].half()  # fp16 SH storage (BASELINE configs[4]); gradients stay fp32
    camd = dict(cam, viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), campos=to(cam["campos"]))
    bg, gt_d, mask_d = to(bg_np), to(gt), to(mask)
    step = parallel.ViewParallelStep(params, deg, camd, bg) if wl["backward"] else None
    from mygauhuman_amd.diff_gaussian_rasterization import _C
    e = torch.empty(0)

    def fwd_only():
        return _C.rasterize_gaussians(bg, params["means3D"], e, params["opacities"], params["scales"], params["rotations"],
                                      1.0, e, camd["viewmatrix"], camd["projmatrix"], cam["tanfovx"], cam["tanfovy"], H, W,
                                      params["shs"], deg, camd["campos"], False, False)

    from mygauhuman_amd.fastpath import RasterSession
    fsession = None if wl["backward"] else RasterSession.calibrated(params, camd, bg, deg, with_backward=False)

    def one_step():
        if wl["backward"]:
            step(camd, bg, gt_d, mask_d, reduce=world > 1)
        else:
            fsession.forward(params, camd, bg, deg)  # sync-free session, like the fwd+bwd workloads

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed: warm-up, then a pass with every stage bracketed by events to find the dominant kernel
    for _ in range(max(1, a.warmup)):
        one_step()
    sync()
    R = step.session.num_rendered() if step is not None else fwd_only()[0]
    _lib.profile_enable(_lib.PROF_STAGES)
    for _ in range(5):
        one_step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    stage_ms = {k: (ms / n if n else 0.0) for k, (ms, n) in prof.items()}
    dominant = max(stage_ms, key=stage_ms.get)
    _lib.profile_enable([dominant])  # only the dominant kernel keeps its two event records in the timed region

    # ---- timed region
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    Rt = torch.tensor([float(R)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(Rt, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    if (step is not None and step.session.overflowed()) or (fsession is not None and fsession.overflowed()):
        raise SystemExit("binning capacity overflow during the timed region: results invalid")
    dom_ms, dom_n = _lib.profile_read()[dominant]
    _lib.profile_enable([])

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        fps = world * a.steps / elapsed
        sb = stage_bytes(P, R, W * H, M, wl["backward"])
        dom_avg_ms = dom_ms / max(dom_n, 1)
        achieved = sb[dominant] / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        frame_bytes = sum(sb.values())
        out = {
            "metric": "frames/sec fwd+bwd @1024^2, 200k Gaussians" if a.workload == "C3" else f"frames/sec {wl['desc']}",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {wl['desc']}; S-uniform seed 0, FoV 50deg, "
                                   + ("identity camera" if world == 1 else f"{world} views/step, 1 view/GPU (cameras orbiting the scene centre in 3deg steps), RCCL all-reduce of "
                                      f"{(step.bucket.nbytes if step else 0) / 1e6:.1f} MB gradients"
                                      + (f" + all-gather of {step.compact.stride * 4 / 1e6:.1f} MB/rank (compact SH gradient)"
                                         if (step is not None and step.compact is not None) else "")),
                       "P": P, "sh_degree": deg, "width": W, "height": H, "num_rendered_rank0": int(R),
                       "binning": "tile_bucket" if _lib.lib.gsr_get_binning_mode() == 1 else "global_radix",
                       "host_sync_per_step": 0},
            "splatted_gaussians_per_s": round(fps * P, 1),
            "instances_per_s": round(float(Rt.item()) * a.steps / elapsed, 1),
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), **pmc_traffic(dominant, a.workload),
                         "algorithmic_bytes_per_launch": int(sb[dominant]), "avg_launch_ms": round(dom_avg_ms, 5),
                         "launches_timed": int(dom_n)},
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items() if k in sb},
            "frame_algorithmic_bytes": int(frame_bytes),
            "frame_hbm_frac": round(frame_bytes * fps / world / 1e9 / HBM_PEAK_GBS, 5),
            "hbm_copy_measured_gbs": measured_copy_gbs(dev),
            "list_stats": list_stats(step.session if step is not None else fsession, P, W, H),
        }
        if not a.no_cpu_baseline and world == 1:  # the CPU baseline is reported by the 1-GPU run only
            out["cpu_baseline"] = cpu_baseline(wl, cameras.make_camera(W, H, 50.0), g, gt, mask, bg_np)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
