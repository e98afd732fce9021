"""ctypes binding of libgsr.so (the C ABI declared in include/gsr.h).

There is NO fallback: if the HIP library is missing or does not load, importing this module raises."""
import ctypes as C
import os

# torch ships its own libamdhip64 / libhsa-runtime64 under torch/lib.  Import it BEFORE libgsr.so is dlopen'ed so that both
# bind to ONE HIP runtime (the loader reuses the already loaded soname); the other order puts two runtimes in the process
# and the second one finds "no ROCm-capable device".
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsr.so")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class Phase1LossStruct(C.Structure):
    """gsr_phase1_loss (include/gsr.h): the fused phase-1 training loss of render()."""
    _fields_ = [("gt_image", C.c_void_p), ("gt_normal", C.c_void_p), ("alpha_target", C.c_void_p), ("bound", C.c_void_p),
                ("w_image", C.c_float), ("w_alpha", C.c_float), ("w_normal", C.c_float), ("w_axis", C.c_float),
                ("normal_triple", C.c_int), ("axis_triple", C.c_int),
                ("color", C.c_void_p), ("alpha", C.c_void_p), ("extra_images", C.c_void_p), ("stats", C.c_void_p),
                ("upstream", C.c_void_p)]

# every symbol include/gsr.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "gsr_version", "gsr_has_experiments", "gsr_target_arch", "gsr_last_error", "gsr_set_binning_mode", "gsr_get_binning_mode", "gsr_set_tuning", "gsr_set_stream_tuning", "gsr_clear_stream_tuning", "gsr_profile_enable", "gsr_profile_reset", "gsr_profile_read", "gsr_debug_wave_trace", "gsr_debug_clock_probe",
    "gsr_mark_visible", "gsr_rasterize_forward", "gsr_rasterize_backward", "gsr_query_state",
    "gsr_geometry_bytes", "gsr_image_bytes", "gsr_binning_bytes", "gsr_rasterize_forward_async",
    "gsr_alpha_mask_loss_backward", "gsr_phase1_loss_partials", "gsr_phase1_loss_forward", "gsr_rasterize_backward_phase1_loss", "gsr_rasterize_backward_alpha_mask_loss", "gsr_rasterize_forward_ex", "gsr_rasterize_forward_async_ex", "gsr_rasterize_backward_ex",
    "gsr_dist2_workspace_bytes", "gsr_dist2", "gsr_sort_workspace_bytes", "gsr_sort_pairs_u64",
    "gsr_sort_pairs_u32", "gsr_lbs_forward", "gsr_lbs_backward", "gsr_lbs_backward_workgroups", "gsr_lbs_workspace_bytes", "gsr_lbs_grid_build", "gsr_lbs_forward_grid", "gsr_lbs_forward_cached", "gsr_lbs_nn_cache_bytes", "gsr_smpl_pose_forward", "gsr_smpl_pose_backward", "gsr_sh_view_pack", "gsr_sh_grad_from_views", "gsr_sh_view_pack_posed", "gsr_sh_grad_from_views_posed", "gsr_step_status", "gsr_step_finish", "gsr_knn_self", "gsr_knn_nearest", "gsr_gather_rows", "gsr_ssim_forward", "gsr_ssim_backward", "gsr_gemv_rows", "gsr_gemv_rows_t", "gsr_frame_attributes_forward", "gsr_frame_attributes_backward", "gsr_model_activations_forward", "gsr_model_activations_backward", "gsr_frame_attributes_forward_split", "gsr_frame_attributes_backward_split", "gsr_frame_attributes_backward_acc", "gsr_model_activations_backward_acc", "gsr_lbs_offset_mlp_packed_floats", "gsr_lbs_offset_mlp_pack", "gsr_lbs_offset_mlp_forward", "gsr_lbs_offset_mlp_backward_workspace_floats", "gsr_lbs_offset_mlp_backward", "gsr_debug_lbs_offset_mlp_forward_bf16x3", "gsr_lbs_offset_mlp_set_precision",
]

GSR_OK = 0
Q = dict(DEPTHS=0, MEANS2D=1, CONIC_OPACITY=2, RGB=3, COV3D=4, TILES_TOUCHED=5, POINT_OFFSETS=6, CLAMPED=7,
         POINT_LIST=8, KEYS_SORTED=9, RANGES=10, FINAL_T=11, N_CONTRIB=12, ORDER=13)
BINNING_GLOBAL_RADIX, BINNING_TILE_BUCKET = 0, 1
SH_F32, SH_F16 = 0, 1  # sh_dtype of the _ex entry points
N_EXTRA = 18  # extra feature channels of the fused multi-feature blend (six RGB triples)
DEFAULT_BINNING = BINNING_TILE_BUCKET
DEFAULT_TILE_CULL = 1  # tuning knob "tile_cull": exact ellipse-vs-tile culling in the tile-bucket back-end
DEFAULT_BWD_REDUCE = 3
DEFAULT_TILE_ORDER = 1  # tuning knob "tile_order" (csrc/gsr_common.h: Options)
DEFAULT_BLEND_LAYOUT = 0  # tuning knob "blend_layout"
DEFAULT_BLEND_SEGMENTS = 8  # tuning knob "blend_segments": lists >= 8 / 4 x the frame's mean are walked in segments


class GsrError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first (python -m mygauhuman_amd.build, or "
            "__graft_entry__.build()).  There is no CPU/PyTorch fallback for the rasterizer.")
    lib = C.CDLL(LIB_PATH)
    vp, fp, ip, sz = C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t  # device pointers travel as integers
    lib.gsr_version.restype = C.c_int
    lib.gsr_has_experiments.restype = C.c_int
    lib.gsr_target_arch.restype = C.c_char_p
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_set_binning_mode.argtypes = [C.c_int]
    lib.gsr_get_binning_mode.restype = C.c_int
    lib.gsr_set_tuning.argtypes = [C.c_char_p, C.c_int]
    lib.gsr_set_stream_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.gsr_clear_stream_tuning.argtypes = [C.c_void_p]
    lib.gsr_set_stream_tuning.restype = lib.gsr_clear_stream_tuning.restype = C.c_int
    lib.gsr_profile_enable.argtypes = [C.c_uint]
    lib.gsr_debug_wave_trace.argtypes = [C.c_void_p, C.c_size_t]
    lib.gsr_debug_clock_probe.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.gsr_debug_clock_probe.restype = C.c_int
    lib.gsr_profile_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    lib.gsr_mark_visible.argtypes = [C.c_int, fp, fp, fp, vp, vp]
    lib.gsr_rasterize_forward.argtypes = [
        ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp,
        C.c_float, fp, fp, fp, fp, fp, C.c_float, C.c_float, C.c_int, fp, fp, fp, ip, C.c_int, C.POINTER(C.c_int), vp]
    lib.gsr_rasterize_backward.argtypes = [
        C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_float, fp, fp, fp, fp, fp,
        C.c_float, C.c_float, ip, vp, vp, vp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_int, vp]
    lib.gsr_geometry_bytes.argtypes = [C.c_int]
    lib.gsr_geometry_bytes.restype = sz
    lib.gsr_image_bytes.argtypes = [C.c_int, C.c_int]
    lib.gsr_image_bytes.restype = sz
    lib.gsr_binning_bytes.argtypes = [sz, C.c_int, C.c_int]
    lib.gsr_binning_bytes.restype = sz
    lib.gsr_rasterize_forward_async.argtypes = [
        vp, vp, sz, vp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_float, fp, fp, fp, fp, fp,
        C.c_float, C.c_float, C.c_int, fp, fp, fp, ip, C.c_int, vp, vp]
    lib.gsr_rasterize_forward_async.restype = C.c_int
    lib.gsr_alpha_mask_loss_backward.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, C.c_float, fp, fp, vp]
    lib.gsr_alpha_mask_loss_backward.restype = C.c_int
    lib.gsr_rasterize_forward_ex.argtypes = lib.gsr_rasterize_forward.argtypes[:-1] + [fp, C.c_int, fp, C.c_int, vp]
    lib.gsr_rasterize_forward_async_ex.argtypes = lib.gsr_rasterize_forward_async.argtypes[:-1] + [fp, C.c_int, fp, C.c_int, vp]
    lib.gsr_rasterize_backward_ex.argtypes = lib.gsr_rasterize_backward.argtypes[:-1] + [fp, C.c_int, C.POINTER(C.c_void_p), fp, C.c_int, vp]
    lib.gsr_rasterize_backward_alpha_mask_loss.argtypes = (lib.gsr_rasterize_backward.argtypes[:24] + [fp, fp, fp, C.c_float] + [fp] * 9
                                                           + [C.c_int, C.c_int, vp])
    lib.gsr_phase1_loss_partials.restype = sz
    lib.gsr_phase1_loss_forward.argtypes = [C.c_int, C.c_int, C.POINTER(Phase1LossStruct), fp, vp]
    lib.gsr_phase1_loss_forward.restype = C.c_int
    lib.gsr_rasterize_backward_phase1_loss.argtypes = lib.gsr_rasterize_backward_ex.argtypes[:-1] + [C.POINTER(Phase1LossStruct), vp]
    for _n in ("gsr_rasterize_forward_ex", "gsr_rasterize_forward_async_ex", "gsr_rasterize_backward_ex",
               "gsr_rasterize_backward_alpha_mask_loss", "gsr_rasterize_backward_phase1_loss"):
        getattr(lib, _n).restype = C.c_int
    lib.gsr_query_state.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.gsr_dist2_workspace_bytes.argtypes = [C.c_int]
    lib.gsr_dist2_workspace_bytes.restype = sz
    lib.gsr_dist2.argtypes = [C.c_int, fp, fp, vp, sz, vp]
    lib.gsr_sort_workspace_bytes.argtypes = [sz]
    lib.gsr_sort_workspace_bytes.restype = sz
    lib.gsr_sort_pairs_u64.argtypes = [sz, vp, vp, vp, vp, C.c_int, vp, sz, vp]
    lib.gsr_sort_pairs_u32.argtypes = [sz, vp, vp, vp, vp, C.c_int, vp, sz, vp]
    lib.gsr_lbs_forward.argtypes = [C.c_int, C.c_int] + [fp] * 12 + [ip] + [fp] * 6 + [vp]
    lib.gsr_lbs_forward_grid.argtypes = [C.c_int, C.c_int] + [fp] * 12 + [ip] + [fp] * 6 + [vp, sz, C.c_int, vp]
    lib.gsr_lbs_forward_cached.argtypes = [C.c_int, C.c_int] + [fp] * 12 + [ip] + [fp] * 6 + [vp, sz, vp, sz, C.c_int, vp]
    lib.gsr_lbs_forward_cached.restype = C.c_int
    lib.gsr_lbs_nn_cache_bytes.argtypes = [C.c_int]
    lib.gsr_lbs_nn_cache_bytes.restype = sz
    lib.gsr_lbs_grid_build.argtypes = [C.c_int, fp, vp, sz, vp]
    lib.gsr_lbs_grid_build.restype = C.c_int
    lib.gsr_lbs_forward_grid.restype = C.c_int
    lib.gsr_lbs_workspace_bytes.argtypes = [C.c_int]
    lib.gsr_lbs_workspace_bytes.restype = sz
    lib.gsr_smpl_pose_forward.argtypes = [fp, fp, fp, C.POINTER(C.c_int), fp, fp, vp]
    lib.gsr_smpl_pose_backward.argtypes = [fp, fp, fp, C.POINTER(C.c_int)] + [fp] * 5 + [vp]
    lib.gsr_smpl_pose_forward.restype = lib.gsr_smpl_pose_backward.restype = C.c_int
    lib.gsr_sh_view_pack.argtypes = [C.c_int, vp, fp, fp, vp]
    lib.gsr_sh_grad_from_views.argtypes = [C.c_int] * 4 + [fp, fp, sz, C.c_float, fp, fp, vp]
    lib.gsr_sh_view_pack.restype = lib.gsr_sh_grad_from_views.restype = C.c_int
    lib.gsr_step_status.argtypes = [C.c_int, vp, fp, C.c_float, fp, vp, vp]
    lib.gsr_step_status.restype = C.c_int
    lib.gsr_sh_view_pack_posed.argtypes = [C.c_int, fp, fp, fp, fp, fp, sz, sz, vp]
    lib.gsr_sh_grad_from_views_posed.argtypes = [C.c_int, C.c_int, C.c_int, fp, sz, sz, sz, C.c_float, fp, fp, fp, vp]
    lib.gsr_step_finish.argtypes = [vp, fp, sz, sz, C.c_float, fp, vp, vp]
    lib.gsr_sh_view_pack_posed.restype = lib.gsr_sh_grad_from_views_posed.restype = lib.gsr_step_finish.restype = C.c_int
    lib.gsr_knn_self.argtypes = [C.c_int, fp, C.c_int, ip, fp, vp, sz, vp]
    lib.gsr_knn_nearest.argtypes = [C.c_int, fp, C.c_int, fp, ip, fp, vp, sz, vp]
    lib.gsr_knn_self.restype = lib.gsr_knn_nearest.restype = C.c_int
    lib.gsr_gather_rows.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.c_int, ip, vp]
    lib.gsr_gather_rows.restype = C.c_int
    lib.gsr_ssim_forward.argtypes = [C.c_int] * 3 + [fp] * 6 + [vp]
    lib.gsr_ssim_backward.argtypes = [C.c_int] * 3 + [fp] * 3 + [C.c_float] + [fp] * 4 + [vp]
    lib.gsr_ssim_forward.restype = lib.gsr_ssim_backward.restype = C.c_int
    lib.gsr_gemv_rows.argtypes = [C.c_int, C.c_int, fp, fp, fp, vp]
    lib.gsr_gemv_rows_t.argtypes = [C.c_int, C.c_int, fp, fp, fp, vp]
    lib.gsr_gemv_rows.restype = lib.gsr_gemv_rows_t.restype = C.c_int
    lib.gsr_lbs_backward.argtypes = [C.c_int, C.c_int, fp, fp, ip] + [fp] * 8 + [fp] * 3 + [fp] * 6 + [vp]
    lib.gsr_lbs_backward_workgroups.argtypes = [C.c_int]
    lib.gsr_lbs_backward_workgroups.restype = C.c_int
    lib.gsr_frame_attributes_forward.argtypes = [C.c_int] * 3 + [fp] * 4 + [C.c_float] + [fp] * 8 + [fp] * 3 + [vp]
    lib.gsr_frame_attributes_backward.argtypes = [C.c_int] * 3 + [fp] * 4 + [C.c_float] + [fp] * 8 + [fp] * 3 + [fp] * 10 + [vp]
    lib.gsr_frame_attributes_forward_split.argtypes = [C.c_int] * 3 + [fp] * 4 + [C.c_float] + [fp] * 9 + [fp] * 3 + [vp]
    lib.gsr_frame_attributes_backward_split.argtypes = [C.c_int] * 3 + [fp] * 4 + [C.c_float] + [fp] * 9 + [fp] * 3 + [fp] * 11 + [vp]
    lib.gsr_model_activations_forward.argtypes = [C.c_int] + [fp] * 11 + [vp]
    lib.gsr_model_activations_backward.argtypes = [C.c_int] + [fp] * 16 + [vp]
    lib.gsr_model_activations_backward_acc.argtypes = [C.c_int] + [fp] * 17 + [vp]
    lib.gsr_lbs_offset_mlp_packed_floats.argtypes = []
    lib.gsr_lbs_offset_mlp_packed_floats.restype = C.c_size_t
    lib.gsr_lbs_offset_mlp_pack.argtypes = [C.POINTER(fp), C.POINTER(fp), fp, vp]
    lib.gsr_lbs_offset_mlp_pack.restype = C.c_int
    lib.gsr_lbs_offset_mlp_forward.argtypes = [C.c_int, fp, fp, fp, vp]
    lib.gsr_lbs_offset_mlp_forward.restype = C.c_int
    lib.gsr_debug_lbs_offset_mlp_forward_bf16x3.argtypes = [C.c_int, fp, fp, fp, vp]
    lib.gsr_debug_lbs_offset_mlp_forward_bf16x3.restype = C.c_int
    lib.gsr_lbs_offset_mlp_set_precision.argtypes = [C.c_int]
    lib.gsr_lbs_offset_mlp_set_precision.restype = C.c_int
    lib.gsr_lbs_offset_mlp_backward_workspace_floats.argtypes = [C.c_int]
    lib.gsr_lbs_offset_mlp_backward_workspace_floats.restype = C.c_size_t
    lib.gsr_lbs_offset_mlp_backward.argtypes = [C.c_int, fp, fp, fp, fp, C.POINTER(fp), C.POINTER(fp), vp]
    lib.gsr_lbs_offset_mlp_backward.restype = C.c_int
    lib.gsr_frame_attributes_backward_acc.argtypes = [C.c_int] * 3 + [fp] * 4 + [C.c_float] + [fp] * 9 + [fp] * 3 + [fp] * 11 + [fp] + [vp]
    for name in ("gsr_frame_attributes_forward", "gsr_frame_attributes_backward", "gsr_model_activations_forward", "gsr_model_activations_backward", "gsr_frame_attributes_forward_split", "gsr_frame_attributes_backward_split", "gsr_frame_attributes_backward_acc", "gsr_model_activations_backward_acc", "gsr_set_binning_mode", "gsr_set_tuning", "gsr_mark_visible", "gsr_rasterize_forward",
                 "gsr_rasterize_backward", "gsr_query_state", "gsr_dist2", "gsr_sort_pairs_u64", "gsr_sort_pairs_u32",
                 "gsr_lbs_forward", "gsr_lbs_backward"):
        getattr(lib, name).restype = C.c_int
    return lib


lib = _load()


def check(rc, what):
    if rc != GSR_OK:
        msg = lib.gsr_last_error()
        raise GsrError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a tensor, or None for an empty tensor (selects the other input mode, like the reference)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


PROF_STAGES = ["preprocess_fwd", "scan", "binning", "blend_fwd", "blend_bwd", "preprocess_bwd"]


def profile_enable(stages):
    """stages: iterable of names from PROF_STAGES (empty = off)."""
    mask = 0
    for s in stages:
        mask |= 1 << PROF_STAGES.index(s)
    check(lib.gsr_profile_enable(mask), "gsr_profile_enable")
    check(lib.gsr_profile_reset(), "gsr_profile_reset")


def profile_read():
    """{stage: (total_ms, launches)} accumulated since profile_enable()."""
    out = {}
    for i, s in enumerate(PROF_STAGES):
        ms, n = C.c_double(0), C.c_long(0)
        check(lib.gsr_profile_read(i, C.byref(ms), C.byref(n)), "gsr_profile_read")
        out[s] = (ms.value, n.value)
    return out


def clock_probe(workgroups=1024, fmas=1 << 19, device=None):
    """Shader clock of the device right now, in GHz (gsr_debug_clock_probe; blocks until the probe kernel has run: ~2 ms at the
    default chain length).  Returns (GHz from s_memtime ticks over the constant 100 MHz counter, median over the workgroups;
    shader cycles per dependent v_fma_f32 of the chain: 8.8 on MI355X, a sanity value that does not move with the clock)."""
    import numpy as np
    out = torch.zeros(2 * workgroups, dtype=torch.int64, device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    check(lib.gsr_debug_clock_probe(workgroups, fmas, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "gsr_debug_clock_probe")
    v = out.cpu().numpy().reshape(workgroups, 2).astype(np.float64)
    ticks, cyc = float(np.median(v[:, 0])), float(np.median(v[:, 1]))
    if ticks <= 0:
        raise GsrError("gsr_debug_clock_probe: the 100 MHz counter did not advance")
    return cyc / (ticks * 10.0), cyc / fmas


def settle_clock(device=None, max_ms=1500.0, tol=0.004, probe_fmas=1 << 19, min_ms=0.0):
    """Run the clock probe back to back until five consecutive readings agree within `tol` (or max_ms have passed): brings a GPU
    that has just been handed to the process to the clock it sustains (2.26 -> 2.39 GHz over ~20 ms of load on MI355X; it falls
    back after ~50 ms of idle).  Returns the list of (ms since start, GHz) readings."""
    import time
    t0, hist = time.perf_counter(), []
    while True:
        ghz, _ = clock_probe(1024, probe_fmas, device)
        ms = (time.perf_counter() - t0) * 1e3
        hist.append((round(ms, 1), round(ghz, 4)))
        last = [h[1] for h in hist[-5:]]
        if (ms >= min_ms and len(last) == 5 and max(last) - min(last) <= tol * max(last)) or ms > max_ms:
            return hist


def set_tuning(key, value, stream=None):
    """Process default of a knob, or (stream = a torch.cuda.Stream / raw handle) that stream's own value."""
    if stream is None:
        check(lib.gsr_set_tuning(key.encode(), int(value)), "gsr_set_tuning")
    else:
        check(lib.gsr_set_stream_tuning(getattr(stream, "cuda_stream", stream), key.encode(), int(value)), "gsr_set_stream_tuning")


def clear_stream_tuning(stream):
    check(lib.gsr_clear_stream_tuning(getattr(stream, "cuda_stream", stream)), "gsr_clear_stream_tuning")
