#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Gaussian-splat hot path.

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one rank per GPU,
RANK / LOCAL_RANK / WORLD_SIZE from the environment), or as plain `python bench.py --gpus N`, which starts its own N
rank processes BEFORE anything touches the GPU and exits with their status.  With fewer devices than ranks (one-GPU box)
the ranks share device 0 over gloo: a functional rehearsal of the view-parallel path, labelled as such, not a scaling number.

A step = one pass of the hot path over one synthetic camera view: rasterizer forward (preprocess -> binning ->
blend), the alpha-mask loss gradient (L1(color, gt) + 0.1 MSE(alpha, mask), train.py:261-262; formed per pixel inside the
blend-backward kernel, gsr_rasterize_backward_alpha_mask_loss), rasterizer backward (blend backward -> backward preprocess) and, for N > 1, the exchange of the Gaussian gradients (RCCL all-reduce of the
flat bucket + compact SH all-gather).  Workload at N = 1: BASELINE.json configs[2] ("C3"): 200k Gaussians, SH degree 3,
1024x1024, fp32, seeded synthetic scene of SURVEY.md §8(d) (S-uniform).  For N > 1 every rank renders its own view of the
same replica (weak scaling).

Rank 0 prints ONE JSON line: frames/s (whole job), median / p10 / p90 of the per-step times, the roofline of the dominant
kernel (HIP events recorded on the launch stream inside the timed region) against the HBM peak AND against the VALU issue
rate that actually bounds it, the forward-only figure, the other single-GPU configs (C2, C5) as `extra`, and a CPU baseline
(the C oracle of the same workload on the host cores, plus the reference's CPU LBS+project path restated in C).
"""
import argparse
import json
import math
import os
import sys
import time

# read by the HIP runtime when it starts (hipGraph replays, mygauhuman_amd/graph.py): before anything imports torch
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E vendor peak (MI355X_MICROARCH.md); measured copy ceiling is ~6300
N_SIMD = 1024           # 256 CUs x 4 SIMDs
# cycles per wave64 VALU instruction per SIMD: a SIMD-32 issues the 64 lanes of a v_fma_f32 in two cycles
# (MI355X_MICROARCH.md "Wave scheduling"); the issue floor of a kernel is instructions x 2 cycles / 1024 SIMDs at the shader
# clock MEASURED on this box in this run (gsr_debug_clock_probe) -- rounds 1-3 priced it with a constant 1.08 ns taken on a cold GPU
VALU_CYCLES = 2.0

WORKLOADS = {
    "C2": dict(P=50_000, W=512, H=512, deg=0, backward=False, desc="50k Gaussians, SH0, 512x512, forward only"),
    "C3": dict(P=200_000, W=1024, H=1024, deg=3, backward=True, desc="200k Gaussians, SH3, 1024x1024, fwd+bwd, alpha-mask loss"),
    "C5": dict(P=500_000, W=1024, H=1024, deg=3, backward=True, log_scale=math.log(0.005), sh_half=True,
               desc="500k Gaussians, SH3 stored as fp16, 1024x1024, fwd+bwd (scale 0.005)"),
}


def stage_bytes(P, R_ref, R_walked, HW, M, backward=True):
    """Algorithmic (compulsory) bytes per stage, SURVEY.md §8(d): every array counted once per stage that must
    produce or consume it, the sort as one read + one write of its 12-byte pairs, atomics as one write of the
    final per-Gaussian gradient.  R_ref = the reference's instance count (sum of tiles_touched); R_walked = the
    instances the binning back-end keeps and the blend kernels walk (after the exact tile culling): the binning and
    blend terms use R_walked."""
    del R_ref
    b = dict(preprocess_fwd=P * (119 + 12 * M), scan=P * 8, binning=P * 20 + R_walked * 44, blend_fwd=R_walked * 44 + HW * 28)
    if backward:
        b.update(blend_bwd=R_walked * 44 + HW * 28 + P * 44, preprocess_bwd=P * (175 + 24 * M))
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


def pmc_summary(stage, workload):
    """Counter figures per launch of the dominant kernel.  PMC passes cannot run inside the bench, so they come from the newest
    committed summary profiles/*_pmc.csv (tools/profile_round.sh: separate rocprofv3 --pmc passes of THIS command for
    FETCH_SIZE, WRITE_SIZE and the SQ counters; FETCH_SIZE doubled, the gfx950 correction of MI355X_MICROARCH.md) -- C3 only."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.csv")))  # r1d < r2a < ...: the newest round last
    prefix = {"blend_bwd": ("blend_backward_lds_kernel<", "blend_backward_kernel<"), "blend_fwd": ("blend_forward_kernel<",)}.get(stage)
    if workload != "C3" or not files or prefix is None:
        return None
    for r in csv.DictReader(open(files[-1])):
        if r["kernel"].startswith(prefix):
            out = {"source": "profiles/" + os.path.basename(files[-1]), "kernel": r["kernel"],
                   "traffic": int(float(r["fetch_bytes_x2"]) + float(r["write_bytes"]))}
            for k_csv, k in (("SQ_INSTS_VALU_per_launch", "valu_insts"), ("SQ_ACTIVE_INST_VALU_per_launch", "active_inst_valu_quads"),
                             ("SQ_BUSY_CYCLES_per_launch", "sq_busy_cycles"), ("avg_us_under_pmc", "avg_us_under_pmc"),
                             ("SQ_LDS_IDX_ACTIVE_per_launch", "lds_idx_active"), ("SQ_LDS_BANK_CONFLICT_per_launch", "lds_bank_conflict")):
                try:
                    out[k] = float(r[k_csv])
                except (KeyError, ValueError):
                    pass
            return out
    return None


class Scene:
    """Device-resident inputs of one workload + the sync-free session(s) that render it."""

    def __init__(self, name, rank, world, dev):
        import torch

        from mygauhuman_amd import cameras, parallel, synthetic
        from mygauhuman_amd.fastpath import RasterSession
        self.name, self.wl = name, WORKLOADS[name]
        wl = self.wl
        self.P, self.W, self.H, self.deg = int(os.environ.get("GSR_BENCH_P", wl["P"])), wl["W"], wl["H"], wl["deg"]
        self.M = (self.deg + 1) ** 2
        self.g = synthetic.uniform_gaussians(self.P, seed=0, sh_degree=self.deg, log_scale_mean=wl.get("log_scale", math.log(0.01)))
        self.gt, self.mask = synthetic.loss_targets(self.W, self.H, seed=0)
        # one view per rank: yaw the S-uniform camera about the scene centre (a 1-GPU run = the identity view)
        yaw = (rank - (world - 1) / 2.0) * 3.0
        self.cam = cameras.orbit_camera(self.W, self.H, yaw) if world > 1 else cameras.make_camera(self.W, self.H, 50.0)
        self.bg_np = np.zeros(3, np.float32)
        to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
        g = self.g
        self.params = dict(means3D=to(g["means3D"]), shs=to(g["shs"]), opacities=to(g["opacities"]), scales=to(g["scales"]),
                           rotations=to(g["rotations"]))
        if wl.get("sh_half"):
            self.params["shs"] = self.params["shs"].half()  # fp16 SH storage (BASELINE configs[4]); gradients stay fp32
        self.camd = dict(self.cam, viewmatrix=to(self.cam["viewmatrix"]), projmatrix=to(self.cam["projmatrix"]),
                         campos=to(self.cam["campos"]))
        self.bg, self.gt_d, self.mask_d = to(self.bg_np), to(self.gt), to(self.mask)
        self.world = world
        self.step = parallel.ViewParallelStep(self.params, self.deg, self.camd, self.bg) if wl["backward"] else None
        self.fsession = (RasterSession.calibrated(self.params, self.camd, self.bg, self.deg, with_backward=False)
                         if self.step is None else None)

    @property
    def session(self):
        return self.step.session if self.step is not None else self.fsession

    def one_step(self):
        if self.step is not None:
            self.step(self.camd, self.bg, self.gt_d, self.mask_d, reduce=self.world > 1)
        else:
            self.fsession.forward(self.params, self.camd, self.bg, self.deg)

    def forward_only(self):
        self.session.forward(self.params, self.camd, self.bg, self.deg)

    def check(self):
        if self.step is not None:
            self.step.check()  # raises BinningOverflow for any overflowed step
        if self.session.overflowed():
            raise SystemExit("binning capacity overflow during the timed region: results invalid")

    def list_stats(self):
        """Per-tile list statistics of the last frame: instances kept after the exact tile culling, mean list
        length over non-empty tiles, fraction of pixels whose blend stopped early (transmittance cut-off before the list end)."""
        import torch

        from mygauhuman_amd.diff_gaussian_rasterization import _C
        s, P, W, H = self.session, self.P, self.W, self.H
        q = lambda what: _C.query_state(what, P, s.capacity, W, H, s.geom, s.bin, s.img)  # noqa: E731
        ranges = q("RANGES").long()
        lens = ranges[:, 1] - ranges[:, 0]
        gx = (W + 15) // 16
        ncon, T = q("N_CONTRIB").long(), q("FINAL_T")
        ys, xs = torch.meshgrid(torch.arange(H, device=T.device), torch.arange(W, device=T.device), indexing="ij")
        tile_len = lens[(ys // 16) * gx + xs // 16]
        early = (ncon < tile_len) & (T < 1e-3)
        ne = lens > 0
        return {"instances_reference": int(s.num_rendered()), "instances_after_tile_cull": int(lens.sum()),
                "mean_list_length": round(float(lens[ne].float().mean()), 1) if bool(ne.any()) else 0.0,
                "max_list_length": int(lens.max()), "pixels_stopped_early_frac": round(float(early.float().mean()), 4)}


def timed(fn, steps, sync, per_step_events=True):
    """EXACTLY `steps` calls of fn bracketed by sync() on both sides; returns (elapsed seconds, per-step ms from events)."""
    import torch
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step_events else []
    sync()
    t0 = time.perf_counter()
    if per_step_events:
        evs[0].record()
    for i in range(steps):
        fn()
        if per_step_events:
            evs[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)] if per_step_events else []
    return elapsed, per


def pct(xs):
    if not xs:
        return None
    a = np.asarray(xs, np.float64)
    return {"median_ms": round(float(np.median(a)), 4), "p10_ms": round(float(np.percentile(a, 10)), 4),
            "p90_ms": round(float(np.percentile(a, 90)), 4)}


def measured_copy_gbs(dev):
    """Stream-copy ceiling of this GPU (read + write bytes per second of a 1 GiB device-to-device copy), SURVEY.md §8(d)."""
    import torch
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


def extra_workload(name, dev, steps=60, warmup=5):
    """A secondary single-GPU config (C2 / C5) for the `extra` block of the line: frames/s from the same timed-loop protocol."""
    import torch
    sc = Scene(name, 0, 1, dev)
    for _ in range(warmup):
        sc.one_step()
    elapsed, per = timed(sc.one_step, steps, torch.cuda.synchronize)
    sc.check()
    ls = sc.list_stats()
    return {"workload": f"{name}: {sc.wl['desc']}", "value": round(steps / elapsed, 2), "unit": "frames/s",
            "ms_per_step": round(elapsed / steps * 1e3, 4), "steps": steps, "dtype": "f32" if not sc.wl.get("sh_half") else "f32 (SH stored f16)",
            "step_ms": pct(per), "instances_reference": ls["instances_reference"],
            "instances_after_tile_cull": ls["instances_after_tile_cull"]}


RENDER_WL = dict(P=200_000, V=6890, W=1024, H=1024, desc="render(): 200k articulated Gaussians (LBS -> attributes -> fused 21-channel "
                 "raster), 1024x1024, fwd+bwd, phase-1 training loss of train.py:261-265 (bound-masked L1 image / normal / axis + 0.1 L2 "
                 "alpha); motion decoders off in the `extra` figures `eager` / `one_graph`, on in `with_reference_sized_decoder*` and in --workload render")
PHASE1_KEYS = ("render", "render_alpha", "normal", "render_axis")   # the images train.py:256-286 puts in the loss before the PBR phase


def _phase1_loss(out):
    """(tools/: the round-3 stand-in, four plain means)"""
    return sum(out[k].mean() for k in PHASE1_KEYS)


def _render_pipe():
    import types
    return types.SimpleNamespace(debug=False, compute_cov3D_python=True, convert_SHs_python=True)


def _phase1_targets(W, H, dev, seed=0):
    """Synthetic targets of the phase-1 training loss (train.py:246-262): gt image, gt normal, background mask, bound mask (a box
    around the subject with ragged edges, ~45 % of the pixels: what cv2.boundingRect-sized ZJU masks look like)."""
    import torch
    rng = np.random.default_rng(seed)
    gt_image = torch.from_numpy(rng.uniform(0, 1, (3, H, W)).astype(np.float32)).to(dev)
    gt_normal = torch.from_numpy(rng.uniform(0, 1, (3, H, W)).astype(np.float32)).to(dev)
    bkgd = torch.from_numpy((rng.uniform(0, 1, (1, H, W)) > 0.5).astype(np.float32)).to(dev)
    bound = np.zeros((1, H, W), np.float32)
    bound[:, H // 8: H - H // 8, W // 4: W - W // 4] = 1.0
    return gt_image, gt_normal, bkgd, torch.from_numpy(bound).to(dev)


def _phase1_loss_torch(out, gt_image, gt_normal, bkgd, bound):
    """The loss of train.py:261-265 in torch ops, in the form a graph capture accepts (the reference's boolean-mask indexing has a
    data-dependent shape: a host synchronisation per term; masked sums divided by the bound count are the same numbers)."""
    nb = bound.sum()
    l1 = lambda a, b: ((a - b).abs() * bound).sum() / (3.0 * nb)  # noqa: E731
    mask_loss = (((out["render_alpha"] - bkgd) ** 2) * bound).sum() / nb
    return l1(out["render"], gt_image) + 0.1 * mask_loss + l1(out["normal"], gt_normal) + l1(out["render_axis"], gt_normal)


def render_extra(dev, steps=40, warmup=30):
    """render() forward + backward at 200k articulated Gaussians / 1024^2 through the drop-in signature
    (gaussian_renderer/__init__.py:53) with the phase-1 TRAINING loss of train.py:261-265 (bound-masked L1 on image / normal / axis,
    0.1 L2 on alpha): fused with the rasterizer (render(..., fused_loss=Phase1Loss): the default figures `eager` / `one_graph`) and
    written in torch ops (`torch_loss`), each eagerly and as ONE hipGraph replay (mygauhuman_amd.graph.GraphedFrame)."""
    import torch

    from mygauhuman_amd import human_synth
    from mygauhuman_amd.diff_gaussian_rasterization._C import Phase1Loss
    from mygauhuman_amd.gaussian_renderer import render
    from mygauhuman_amd.graph import GraphedFrame
    wl = RENDER_WL
    model, body = human_synth.build(wl["P"], wl["V"], dev, seed=0)
    cam = human_synth.view_camera(body, wl["W"], wl["H"], 0, n_views=8, device=dev)
    bg, pipe, params = torch.zeros(3, device=dev), _render_pipe(), list(model.parameters())
    targets = _phase1_targets(wl["W"], wl["H"], dev)
    spec = Phase1Loss(*targets)

    def step_fused():
        o = render(1, cam, model, pipe, bg, fused_loss=spec)
        o["loss"].backward()
        return o["render"]

    def step_torch():
        o = render(1, cam, model, pipe, bg)
        _phase1_loss_torch(o, *targets).backward()
        return o["render"]

    def measure(step, params):
        def eager():
            for p in params:
                p.grad = None
            step()
        for _ in range(warmup):
            eager()
        el, per = timed(eager, steps, torch.cuda.synchronize)
        res = {"eager": {"value": round(steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 4), "step_ms": pct(per)}}
        try:
            frame = GraphedFrame(step, warmup=3, zero_grads=params)   # verifies itself (replay / eager work / replay vs eager)
            for _ in range(5):
                frame.replay()
            el, per = timed(frame.replay, steps, torch.cuda.synchronize)
            frame.check()
            res["one_graph"] = {"value": round(steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 4),
                                "step_ms": pct(per), "self_check": "passed"}
        except RuntimeError as ex:   # never silent: the line says why there is no graph number
            res["one_graph"] = {"error": str(ex)[:300]}
        return res
    out = {"workload": wl["desc"]}
    out.update(measure(step_fused, params))
    out["loss"] = ("phase-1 training loss of train.py:261-265 fused with the rasterizer (gsr_phase1_loss_forward + the blend backward's "
                   "prologue): `eager` / `one_graph`; the same loss in torch ops: `torch_loss`")
    out["torch_loss"] = measure(step_torch, params)
    # ---- two more figures: the same fused-loss step with motion_offset_flag on and a per-Gaussian skinning-offset network
    # of the REFERENCE'S layers in the frame (nets.FusedLBSOffsetDecoder: 63-d embedding, 63-128-128-128-(191)-128-24, random init;
    # gaussian_renderer/__init__.py:100-106 runs the reference's own every frame) + the pose MLP stand-in: the network on the fused
    # f32-MFMA kernels of csrc/mlp.hip, and the same module in torch ops
    for key, dec in (("with_reference_sized_decoder", "reference_size"), ("with_reference_sized_decoder_in_torch_ops", "reference_size_torch")):
        model_m, _ = human_synth.build(wl["P"], wl["V"], dev, seed=0, motion=True, decoder=dec)
        params_m = (list(model_m.parameters()) + list(model_m.pose_decoder.parameters())
                    + list(model_m.lweight_offset_decoder.parameters()))

        def step_motion(model_m=model_m):
            o = render(1, cam, model_m, pipe, bg, fused_loss=spec)
            o["loss"].backward()
            return o["render"]
        out[key] = measure(step_motion, params_m)
        del model_m, params_m
    out["with_reference_sized_decoder"]["what"] = (
        "motion_offset_flag on: pose MLP stand-in + a skinning-offset network of the reference's layers (random init), fused phase-1 "
        "loss; the network forward + backward on the f32-MFMA kernels of csrc/mlp.hip (`with_reference_sized_decoder`) or in torch ops "
        "(`..._in_torch_ops`); `eager` / `one_graph` above have the decoders off")
    return out


def dropin_extra(dev, steps=60, warmup=10):
    """C3 forward + backward through the DROP-IN operator surface, the way train.py would call it: GaussianRasterizer(...)(...)
    under autograd (diff_gaussian_rasterization/__init__.py:190-223), torch ops for the alpha-mask loss.  Two figures: the module's
    default since round 4 (no host read of num_rendered -- the module never returns it; deferred, never-silent overflow check) and
    the reference's blocking read of num_rendered every forward (CR/rasterizer_impl.cu:283; diff_gaussian_rasterization.SYNC_FREE =
    False), which is what rounds 2-3 reported."""
    import torch

    import mygauhuman_amd.diff_gaussian_rasterization as dgr
    from mygauhuman_amd import cameras, synthetic
    from mygauhuman_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    wl = WORKLOADS["C3"]
    P, W, H, deg = wl["P"], wl["W"], wl["H"], wl["deg"]
    g = synthetic.uniform_gaussians(P, seed=0, sh_degree=deg)
    gt, mask = synthetic.loss_targets(W, H, seed=0)
    cam = cameras.make_camera(W, H, 50.0)
    to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    t = {k: to(g[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    rast = GaussianRasterizer(GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev), scale_modifier=1.0,
        viewmatrix=to(cam["viewmatrix"]), projmatrix=to(cam["projmatrix"]), sh_degree=deg, campos=to(cam["campos"]), prefiltered=False,
        debug=False))
    gt_d, mask_d = to(gt), to(mask)

    def step():
        for v in t.values():
            v.grad = None
        means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
        color, radii, depth, alpha = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"],
                                          scales=t["scales"], rotations=t["rotations"])
        loss = (color - gt_d).abs().mean() + 0.1 * ((alpha - mask_d) ** 2).mean()
        loss.backward()
    res = {"workload": "C3 through GaussianRasterizer + autograd, torch-op loss: the module default (no host read of num_rendered) and "
                       "`blocking_num_rendered_read` (the reference's read per forward)"}
    saved = dgr.SYNC_FREE
    try:
        for key, flag in ((None, True), ("blocking_num_rendered_read", False)):
            dgr.SYNC_FREE = flag
            for _ in range(warmup):
                step()
            el, per = timed(step, steps, torch.cuda.synchronize)
            r = {"value": round(steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 4), "step_ms": pct(per)}
            if key is None:
                res.update(r)
            else:
                res[key] = r
    finally:
        dgr.SYNC_FREE = saved
    return res


def c5_parts_extra(dev, reps=20):
    """The two C5 items of BASELINE.md §4 that are not the rasterizer: per-frame LBS deform and distCUDA2, at 500k points."""
    import torch

    from mygauhuman_amd import human_synth, lbs
    from mygauhuman_amd.simple_knn._C import distCUDA2
    P = WORKLOADS["C5"]["P"]
    model, body = human_synth.build(P, RENDER_WL["V"], dev, seed=0)
    cam = human_synth.view_camera(body, 1024, 1024, 0, n_views=8, device=dev)
    xyz, nrm = model.get_xyz.detach(), torch.nn.functional.normalize(model._normal.detach())

    def deform():
        return lbs.coarse_deform_c2source(model.SMPL_NEUTRAL, xyz[None], cam.smpl_param, cam.big_pose_smpl_param,
                                          cam.big_pose_world_vertex[None], normals=nrm[None], lean=True)

    def deform_fb():
        q = xyz.clone().requires_grad_(True)
        o = lbs.coarse_deform_c2source(model.SMPL_NEUTRAL, q[None], cam.smpl_param, cam.big_pose_smpl_param,
                                       cam.big_pose_world_vertex[None], normals=nrm[None], lean=True)
        (o[1].sum() + o[3].sum()).backward()
    res = {}
    with torch.no_grad():
        for _ in range(3):
            deform()
        el, _ = timed(deform, reps, torch.cuda.synchronize, per_step_events=False)
        res["lbs_forward_ms"] = round(el / reps * 1e3, 4)
    for _ in range(3):
        deform_fb()
    el, _ = timed(deform_fb, reps, torch.cuda.synchronize, per_step_events=False)
    res["lbs_forward_backward_ms"] = round(el / reps * 1e3, 4)
    for _ in range(3):
        distCUDA2(xyz)
    el, _ = timed(lambda: distCUDA2(xyz), reps, torch.cuda.synchronize, per_step_events=False)
    res["dist2_ms"] = round(el / reps * 1e3, 4)
    res["points"] = P
    res["what"] = ("coarse_deform_c2source (pose kernels + blend-shape GEMV + per-point LBS kernel, frame constants cached) and distCUDA2 "
                   "(blocking, as the reference calls it) at 500k points, wall ms per call")
    return res


def main_render(a, rank, world, local, dev, rehearsal):
    """--workload render: the view-parallel TRAINING step of the articulated model (parallel.ViewParallelRender): every rank renders
    its own ring camera / pose through render(), phase-1 loss, autograd backward, then the exchange of every leaf gradient and of
    the densification statistics.  N = 1: the same step without an exchange."""
    import torch
    import torch.distributed as dist

    from mygauhuman_amd import _lib, human_synth, parallel
    wl = RENDER_WL
    P = int(os.environ.get("GSR_BENCH_P", wl["P"]))
    # the skinning-offset network: the reference's layers on the fused MFMA kernels (nets.FusedLBSOffsetDecoder, random init);
    # GSR_BENCH_DECODER=affine: rounds 3-4's 96-parameter stand-in, =reference_size_torch: the same network in torch ops
    decoder = os.environ.get("GSR_BENCH_DECODER", "reference_size")
    model, body = human_synth.build(P, wl["V"], dev, seed=0, motion=True, decoder=decoder)
    cam = human_synth.view_camera(body, wl["W"], wl["H"], rank, n_views=8, device=dev)
    bg = torch.zeros(3, device=dev)
    step = parallel.ViewParallelRender(model, _render_pipe(), bg)
    from mygauhuman_amd.diff_gaussian_rasterization._C import Phase1Loss
    spec = Phase1Loss(*_phase1_targets(wl["W"], wl["H"], dev, seed=rank))   # every view has its own targets

    def one():
        step(1, cam, lambda o: o["loss"], fused_loss=spec)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    _lib.settle_clock(dev)   # (as in the C3 workload: a freshly leased GPU starts below the clock it sustains)
    for _ in range(max(3, a.warmup)):
        one()
    sync()
    step.check()
    step.timer.reset()
    step.ar_timer.reset()
    if step.compact is not None:
        step.compact.allgather_ms()
    elapsed, per_step = timed(one, a.steps, sync)
    step.check()
    ex, ar = step.timer.read_ms(), step.ar_timer.read_ms()
    ag = step.compact.allgather_ms() if step.compact is not None else []
    t = torch.tensor([elapsed, float(np.mean(ex)) if ex else 0.0, float(np.mean(ar)) if ar else 0.0, float(np.mean(ag)) if ag else 0.0],
                     dtype=torch.float64, device=dev)
    if world > 1:
        parallel.all_reduce_(t, dist.ReduceOp.MAX)
    elapsed, exchange_ms, allreduce_ms, allgather_ms = float(t[0]), float(t[1]), float(t[2]), float(t[3])
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        out = {"metric": "frames/sec render() fwd+bwd @1024^2, 200k articulated Gaussians (view-parallel training step)",
               "value": round(world * a.steps / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": wl["desc"] + f"; S-human seed 0, {world} ring view(s)/step, 1 view/rank, own pose per view; "
                          "motion_offset_flag on: a 69-128-69 pose MLP (stand-in for nets/mlp_delta_body_pose.py) and the per-Gaussian "
                          "skinning-offset network with the layers of nets/mlp_delta_weight_lbs.py (63-d embedding, 63-128-128-128-(191)-"
                          "128-24, random init, run every frame: gaussian_renderer/__init__.py:100-106) "
                          + {"reference_size": "on the fused MFMA kernels of csrc/mlp.hip", "reference_size_torch": "in torch ops",
                             "affine": "REPLACED by an affine 96-parameter stand-in (its cost is NOT in this figure)"}.get(decoder, decoder) + "; "
                          + ("REHEARSAL: all ranks on ONE device over gloo with host-staged collectives -- not a scaling number; "
                             if rehearsal else "") + f"exchange payload {step.payload_bytes / 1e6:.1f} MB/rank"
                          + (" (compact SH: all-gather + all-reduce)" if step.compact is not None else " (one all-reduce)"),
                          "P": P, "width": wl["W"], "height": wl["H"], "leaves": list(step.leaves),
                          "backend": (dist.get_backend() if world > 1 else None)},
               "step_ms": pct(per_step), "exchange_ms": round(exchange_ms, 4), "allreduce_ms": round(allreduce_ms, 4),
               "allgather_ms": round(allgather_ms, 4), "step_compute_ms": round(ms - exchange_ms, 4),
               "splatted_gaussians_per_s": round(world * a.steps / elapsed * P, 1)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()



def self_launch(a, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script and return their status.
    Runs before anything has touched the GPU (no HIP call, no libgsr.so): the parent only waits."""
    from mygauhuman_amd.launch import spawn_ranks
    codes = spawn_ranks([sys.executable, os.path.abspath(__file__)] + argv, a.gpus)
    bad = [c for c in codes if c != 0]
    return bad[0] if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS) + ["render"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the forward-only / C2 / C5 figures")
    ap.add_argument("--binning", type=int, default=-1, help="0 global radix, 1 tile bucket (default: library default)")
    ap.add_argument("--tune", action="append", default=[], help="key=value tuning knob (gsr_set_tuning), repeatable")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    from mygauhuman_amd import _lib, cameras, parallel
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    rank, world, local = parallel.init_distributed("cuda")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rehearsal = world > 1 and os.environ.get("GSR_SINGLE_DEVICE") == "1"
    if a.binning >= 0:
        _lib.check(_lib.lib.gsr_set_binning_mode(a.binning), "gsr_set_binning_mode")
    for kv in a.tune:
        k, v = kv.split("=")
        _lib.set_tuning(k, int(v))
    if a.workload == "render":
        return main_render(a, rank, world, local, dev, rehearsal)

    sc = Scene(a.workload, rank, world, dev)
    wl, P, W, H, M = sc.wl, sc.P, sc.W, sc.H, sc.M

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed: bring the GPU to the shader clock it sustains.  A freshly leased MI355X starts at 2.26 GHz and needs ~20 ms of
    # load to reach 2.39-2.43 (it falls back after 50 ms of idle: tools/clock_ramp.py, profiles/r4_clock_ramp.txt); the driver's
    # `--steps 20 --warmup 5` is 13 ms of work in all and read 10 % below a 300-step run of the same build on the same box (round 4).
    # The probe (a dependent FMA chain per SIMD bracketed by s_memtime and the 100 MHz counter) runs until five readings agree.
    clock_hist = _lib.settle_clock(dev)
    clock_before = clock_hist[-1][1]
    # ---- untimed: two passes with every stage bracketed by events to find the dominant kernel (the smaller average per stage: one
    # stray preemption must not pick the kernel).  They run BETWEEN the clock settle and the warm-up, back to back with it: a pause
    # of tens of milliseconds between the last work of this kind and the timed region costs 3-4 % (profiles/r4_settle_experiments.txt:
    # neither a longer clock settle nor memory traffic during it closes that gap, more steps of the workload itself do).
    sc.one_step()
    stage_ms_cold = None
    for _ in range(2):
        _lib.profile_enable(_lib.PROF_STAGES)
        for _ in range(8):
            sc.one_step()
        torch.cuda.synchronize()
        cur = {k: (ms / n if n else 0.0) for k, (ms, n) in _lib.profile_read().items()}
        stage_ms_cold = cur if stage_ms_cold is None else {k: min(v, stage_ms_cold[k]) for k, v in cur.items()}
    dominant = max(stage_ms_cold, key=stage_ms_cold.get)
    _lib.profile_enable([dominant])  # only the dominant kernel keeps its two event records in the timed region
    # ---- untimed: 64 more steps of the workload itself, back to back (23 ms).  The probe's FMA chain brings the clock to ~2.39 GHz,
    # sustained load of the real kernels to ~2.43: without these steps the driver's 20 timed steps read 2 % below a 300-step run,
    # with them 0.6 % (profiles/r4_settle_experiments.txt, block 4)
    for _ in range(int(os.environ.get("GSR_BENCH_PRESTEPS", "64"))):
        sc.one_step()
    # ---- then the W warm-up steps and the K timed steps; the clock is read again right after the timed region
    for _ in range(max(1, a.warmup)):
        sc.one_step()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides, max over ranks
    if sc.step is not None:
        sc.step.timer.reset()
        sc.step.ar_timer.reset()
        if sc.step.compact is not None:
            sc.step.compact.allgather_ms()
    elapsed, per_step = timed(sc.one_step, a.steps, sync)
    clock_after = _lib.clock_probe(1024, 1 << 19, dev)[0]
    ex_ms = sc.step.timer.read_ms() if sc.step is not None else []
    ar_ms = sc.step.ar_timer.read_ms() if sc.step is not None else []
    ag_ms = sc.step.compact.allgather_ms() if (sc.step is not None and sc.step.compact is not None) else []
    # stage times at the settled clock (the pass above ran on a cold GPU and only chose the kernel to time)
    dom_ms, dom_n = _lib.profile_read()[dominant]
    _lib.profile_enable(_lib.PROF_STAGES)
    for _ in range(5):
        sc.one_step()
    torch.cuda.synchronize()
    stage_ms = {k: (ms / n if n else 0.0) for k, (ms, n) in _lib.profile_read().items()}
    _lib.profile_enable([])
    ls = sc.list_stats()  # (synchronises; after the timed region)
    t = torch.tensor([elapsed, float(np.mean(ex_ms)) if ex_ms else 0.0, float(np.mean(ar_ms)) if ar_ms else 0.0,
                      float(np.mean(ag_ms)) if ag_ms else 0.0], dtype=torch.float64, device=dev)
    Rt = torch.tensor([float(ls["instances_after_tile_cull"])], dtype=torch.float64, device=dev)
    if world > 1:
        parallel.all_reduce_(t, dist.ReduceOp.MAX)
        parallel.all_reduce_(Rt, dist.ReduceOp.SUM)
    elapsed, exchange_ms, allreduce_ms, allgather_ms = float(t[0]), float(t[1]), float(t[2]), float(t[3])
    sc.check()

    out = None
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        fps = world * a.steps / elapsed
        R_ref, R_walked = ls["instances_reference"], ls["instances_after_tile_cull"]
        sb = stage_bytes(P, R_ref, R_walked, W * H, M, wl["backward"])
        sb_ref = stage_bytes(P, R_ref, R_ref, W * H, M, wl["backward"])
        dom_avg_ms = dom_ms / max(dom_n, 1)
        achieved = sb[dominant] / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        frame_bytes = sum(sb.values())
        pmc = pmc_summary(dominant, a.workload)
        roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc["traffic"] if pmc else None,
                "algorithmic_bytes_per_launch": int(sb[dominant]), "algorithmic_bytes_per_launch_reference_R": int(sb_ref[dominant]),
                "avg_launch_ms": round(dom_avg_ms, 5), "launches_timed": int(dom_n)}
        if pmc:
            roof["traffic_source"] = pmc["source"] + (" (counters collected on the builder's box, not in this run: 2 x FETCH_SIZE + WRITE_SIZE "
                                                      "per launch; WRITE_SIZE counts every float-atomic lane as 4 B)")
        if pmc and "valu_insts" in pmc and dom_avg_ms > 0:
            # Issue roofline of the kernel as it is: a SIMD issues a wave64 VALU instruction in VALU_CYCLES cycles at best, so a kernel
            # made of the same number of instructions cannot run faster than  insts x VALU_CYCLES / 1024 SIMDs / clock  -- priced at
            # the shader clock MEASURED in this run.  frac = that floor / the measured launch time.  (Rounds 1-3 read frac ~ 0.6
            # plus a counter-derived "busy" of ~1.0 as "issue-bound"; round 4's census says the LDS array is the busier unit.)
            n_i = pmc["valu_insts"]
            floor_ms = n_i * VALU_CYCLES / N_SIMD / (clock_before * 1e9) * 1e3
            roof["valu_issue"] = {
                "bound": "valu_issue", "valu_insts_per_launch": int(n_i), "cycles_per_inst_per_simd_floor": VALU_CYCLES,
                "clock_ghz": clock_before, "floor_ms": round(floor_ms, 5), "frac": round(floor_ms / dom_avg_ms, 4),
                "cycles_per_inst_per_simd": round(dom_avg_ms * 1e-3 * clock_before * 1e9 * N_SIMD / n_i, 3),
                "insts_per_walked_instance": round(n_i / max(R_walked, 1), 1),
                "source": pmc["source"] + " (instruction count: the builder's box; launch time and clock: this run)"}
            if "lds_idx_active" in pmc:
                # the unit round 4's census found busiest: LDS-array cycles (SQ_LDS_IDX_ACTIVE, summed over the CUs) / 256 CUs over the
                # launch's cycles at the measured clock; conflict cycles are part of them
                roof["lds_array"] = {"busy_cycles_per_launch": int(pmc["lds_idx_active"]), "bank_conflict_cycles": int(pmc.get("lds_bank_conflict", 0)),
                                     "frac": round(pmc["lds_idx_active"] / 256.0 / (dom_avg_ms * 1e-3 * clock_before * 1e9), 4),
                                     "source": pmc["source"]}
        out = {
            "metric": "frames/sec fwd+bwd @1024^2, 200k Gaussians" if a.workload == "C3" else f"frames/sec {wl['desc']}",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {wl['desc']}; "
                                   + (f"P OVERRIDDEN to {P} by GSR_BENCH_P (a profiling experiment, NOT the headline workload); "
                                      if P != wl["P"] else "") + "S-uniform seed 0, FoV 50deg, "
                                   + ("identity camera" if world == 1 else
                                      f"{world} views/step, 1 view/rank (cameras orbiting the scene centre in 3deg steps), "
                                      + ("REHEARSAL: all ranks on ONE device over gloo with host-staged collectives -- functional "
                                         "check of the view-parallel path, not a scaling number; " if rehearsal else "RCCL ")
                                      + f"all-reduce of {sc.step.bucket.nbytes / 1e6:.1f} MB gradients"
                                      + (f" + all-gather of {sc.step.compact.stride * 4 / 1e6:.1f} MB/rank (compact SH gradient)"
                                         if (sc.step is not None and sc.step.compact is not None) else "")),
                       "P": P, "sh_degree": sc.deg, "width": W, "height": H,
                       "instances_reference_rank0": int(R_ref), "instances_after_tile_cull_rank0": int(R_walked),
                       "binning": "tile_bucket" if _lib.lib.gsr_get_binning_mode() == 1 else "global_radix",
                       "host_sync_per_step": 0, "backend": (dist.get_backend() if world > 1 else None)},
            "step_ms": pct(per_step),
            # shader clock (s_memtime over the 100 MHz counter, gsr_debug_clock_probe) when the settle loop ended, i.e. right before
            # the warm-up steps, and right after the timed region; first = the clock the process found the GPU at
            "device_clock_ghz_measured": round(clock_before, 4), "device_clock_ghz_after": round(clock_after, 4),
            "device_clock_ghz_first": clock_hist[0][1], "clock_settle_ms": clock_hist[-1][0],
            # N > 1: HIP events around the collectives of every step (max over ranks of the per-rank mean) and what is left of the
            # step -- so that a scaling run says where its time went
            "exchange_ms": round(exchange_ms, 4) if world > 1 else None,
            # the two collectives travel at the same time (all-gather: side stream, its own communicator): exchange_ms is what the
            # compute stream spends on both; allreduce_ms / allgather_ms are each one's own span (max over ranks of the per-rank mean)
            "allreduce_ms": round(allreduce_ms, 4) if world > 1 else None,
            "allgather_ms": round(allgather_ms, 4) if world > 1 else None,
            "step_compute_ms": round(ms_per_step - exchange_ms, 4) if world > 1 else None,
            "splatted_gaussians_per_s": round(fps * P, 1),
            "instances_per_s": round(float(Rt.item()) * a.steps / elapsed, 1),
            "roofline": roof,
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items() if k in sb},
            "stage_algorithmic_gbs": {k: round(sb[k] / (stage_ms[k] * 1e-3) / 1e9, 1) for k in sb if stage_ms.get(k, 0) > 0},
            "frame_algorithmic_bytes": int(frame_bytes), "frame_algorithmic_bytes_reference_R": int(sum(sb_ref.values())),
            "frame_hbm_frac": round(frame_bytes * fps / world / 1e9 / HBM_PEAK_GBS, 5),
            "list_stats": ls,
        }
    # ---- untimed extras (1-GPU run only): forward-only figure of the same workload, the other single-GPU configs
    if rank == 0 and world == 1 and not a.no_extra:
        out["hbm_copy_measured_gbs"] = measured_copy_gbs(dev)
        extra = {}
        if wl["backward"]:
            for _ in range(3):
                sc.forward_only()
            el, per = timed(sc.forward_only, a.steps, torch.cuda.synchronize)
            extra["forward_only"] = {"workload": f"{a.workload} forward only", "value": round(a.steps / el, 2), "unit": "frames/s",
                                     "ms_per_step": round(el / a.steps * 1e3, 4), "step_ms": pct(per)}
        del sc
        torch.cuda.empty_cache()
        for name in ("C2", "C5"):
            if name != a.workload:
                extra[name] = extra_workload(name, dev)
        torch.cuda.empty_cache()
        # the drop-in surface and the non-rasterizer C5 items
        extra["dropin_rasterizer_c3"] = dropin_extra(dev)
        torch.cuda.empty_cache()
        extra["render_200k"] = render_extra(dev)
        torch.cuda.empty_cache()
        extra["c5_lbs_dist2"] = c5_parts_extra(dev)
        out["extra"] = extra
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:  # the CPU baseline is reported by the 1-GPU run only
            from mygauhuman_amd import synthetic
            g = synthetic.uniform_gaussians(P, seed=0, sh_degree=wl["deg"], log_scale_mean=wl.get("log_scale", math.log(0.01)))
            gt, mask = synthetic.loss_targets(W, H, seed=0)
            out["cpu_baseline"] = cpu_baseline(wl, cameras.make_camera(W, H, 50.0), g, gt, mask, np.zeros(3, np.float32))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
