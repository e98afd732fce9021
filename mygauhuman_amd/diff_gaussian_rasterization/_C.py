"""`diff_gaussian_rasterization._C`: the three raw bindings of the reference extension (DGR/ext.cpp:15-19),
same positional signatures and return tuples (DGR/rasterize_points.h:19-70), implemented over the C ABI of
libgsr.so.  Torch only owns memory and the stream here; all compute is in the HIP library."""
import ctypes as C

import torch

from .. import _lib
from .._lib import ALLOC_FN, check, lib, ptr


def _f32c(t, name):
    if t.numel() and t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t.contiguous()  # silent copy for non-contiguous inputs, DGR/rasterize_points.cu:97-116


def _sh(t):
    """SH coefficients: float32 like the reference, or float16 storage (extension) -> (contiguous tensor, sh_dtype flag)."""
    if t.numel() and t.dtype == torch.float16:
        return t.contiguous(), _lib.SH_F16
    return _f32c(t, "sh"), _lib.SH_F32


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _Scratch:
    """One growable byte tensor handed to the library through an allocation callback (the resize lambda of
    DGR/rasterize_points.cu:27-33)."""

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.error = None
        self.cb = ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        try:
            self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            return self.tensor.data_ptr()
        except Exception as ex:  # noqa: BLE001 -- exceptions cannot cross the C boundary
            self.error = ex
            return 0


FWD_ZERO_ROWS = 2   # GSR_FWD_ZERO_ROWS / GSR_BWD_ROWS_ZEROED (include/gsr.h): the forward zeroes the gradient rows, the backward
BWD_ROWS_ZEROED = 2  # of the same frame then skips its fill kernel -- `rows_zeroed=True` on the three bindings below


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, extra=None, rows_zeroed=False):
    """RasterizeGaussiansCUDA (DGR/rasterize_points.cu:36-120).

    `extra` (extension): [P, 18] float32 feature channels blended in the same pass (fused multi-feature render); the
    return tuple then carries a ninth element, out_extra [18, H, W]."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("rasterize_gaussians: tensors must live on a HIP device (no CPU path)")
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    # the kernels write every pixel and every radius, so no fill kernels are needed unless nothing is launched
    new = torch.zeros if P == 0 else torch.empty
    out_color = new((3, H, W), dtype=torch.float32, device=dev)
    out_depth = new((1, H, W), dtype=torch.float32, device=dev)
    out_alpha = new((1, H, W), dtype=torch.float32, device=dev)
    radii = new((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    rendered = C.c_int(0)
    out_extra = None
    if extra is not None:
        if tuple(extra.shape) != (P, _lib.N_EXTRA):
            raise RuntimeError(f"extra must have shape (num_points, {_lib.N_EXTRA})")
        extra = _f32c(extra, "extra")
        out_extra = new((_lib.N_EXTRA, H, W), dtype=torch.float32, device=dev)
        if P == 0:
            out_extra += background.to(dev).float().repeat(_lib.N_EXTRA // 3)[:, None, None]
    if P != 0:
        M = sh.size(1) if sh.numel() != 0 else 0
        means3D, colors, opacity = _f32c(means3D, "means3D"), _f32c(colors, "colors"), _f32c(opacity, "opacity")
        scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D")
        (sh, sh_dtype), background = _sh(sh), _f32c(background, "background")
        viewmatrix, projmatrix, campos = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos")
        with torch.cuda.device(dev):
            rc = lib.gsr_rasterize_forward_ex(
                geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), int(M), ptr(background), W, H,
                ptr(means3D), ptr(sh), ptr(colors), ptr(opacity), ptr(scales), float(scale_modifier), ptr(rotations),
                ptr(cov3D_precomp), ptr(viewmatrix), ptr(projmatrix), ptr(campos), float(tan_fovx), float(tan_fovy),
                int(bool(prefiltered)), out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr(),
                radii.data_ptr(), int(bool(debug)) | (FWD_ZERO_ROWS if rows_zeroed else 0), C.byref(rendered), ptr(extra),
                0 if extra is None else _lib.N_EXTRA,
                None if out_extra is None else out_extra.data_ptr(), sh_dtype, _stream(dev))
        for s in (geom, binning, img):
            if s.error is not None:
                raise s.error
        check(rc, "gsr_rasterize_forward")
    if extra is not None:
        return rendered.value, out_color, out_depth, out_alpha, radii, geom.tensor, binning.tensor, img.tensor, out_extra
    return rendered.value, out_color, out_depth, out_alpha, radii, geom.tensor, binning.tensor, img.tensor


PREFILTER_MSG = "Point is filtered although prefiltered is set. This shouldn't happen!"  # CR/auxiliary.h:158


class BinningCapacityExceeded(RuntimeError):
    """A sync-free forward needed more (Gaussian, tile) instances than its binning buffer holds: that frame rendered only the
    background.  The capacity has been raised by the time this is raised: render the frame again."""


class AsyncCapacity:
    """Capacity policy and deferred overflow checks of the sync-free forward (rasterize_gaussians_async).

    The blocking read of num_rendered (CR/rasterizer_impl.cu:283) exists to size the binning buffer.  On a 288 GB part the
    buffer is simply sized generously instead -- max(MIN, 64 instances per Gaussian, 2 x the largest R seen so far on that
    device), 24 B per instance -- and the device reports R and a flag word into `status`.  The words travel to pinned host memory
    with an asynchronous copy; they are examined when the frame's backward starts, at the next forward on that device, by
    check(watch) -- which render() calls itself for frames that will have no backward -- or by check_all(): an overflow raises
    RuntimeError there (the overflowing frame rendered only the background) and the capacity is raised for the retry.
    All state is kept per device."""
    MIN = 4 << 20
    _state = {}  # device index -> dict(largest_R, pending, pinned)
    # status tensors of forwards that were recorded into a HIP graph (torch.cuda.graph): a captured region cannot hold the pinned
    # copy / event of watch() (they would bake a host pointer and an unqueryable event into the graph), so those forwards are
    # checked by the caller after replays: AsyncCapacity.check_graph_status()
    graph_status = []

    @classmethod
    def _dev(cls, device):
        idx = torch.device(device).index
        idx = torch.cuda.current_device() if idx is None else idx
        st = cls._state.get(idx)
        if st is None:
            st = cls._state[idx] = dict(largest_R=0, pending=[], pinned=[])
        return st

    @classmethod
    def capacity(cls, P, device=None):
        return int(max(cls.MIN, 64 * P, 2 * cls._dev(device if device is not None else "cuda")["largest_R"]))

    @classmethod
    def status_words(cls, device):
        """Two pinned host words for a forward's (R, flags): pinned memory is mapped into the device's address space, so the
        binning kernel writes them straight to the host -- no device tensor, no copy kernel; the event of watch() tells the host
        when they are there."""
        st = cls._dev(device)
        return st["pinned"].pop() if st["pinned"] else torch.zeros(2, dtype=torch.int32).pin_memory()

    @classmethod
    def watch(cls, host, capacity, device):
        st = cls._dev(device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        w = [host, ev, int(capacity), False, st]
        st["pending"].append(w)
        return w

    @classmethod
    def _examine(cls, w, wait):
        if w is None or w[3]:
            return
        if wait:
            w[1].synchronize()
        elif not w[1].query():
            return
        w[3] = True
        st = w[4]
        if w in st["pending"]:
            st["pending"].remove(w)
        R, flags = int(w[0][0]) & 0xFFFFFFFF, int(w[0][1])
        st["pinned"].append(w[0])
        st["largest_R"] = max(st["largest_R"], R)
        if flags & 2:
            raise RuntimeError("rasterize_gaussians_async: " + PREFILTER_MSG)
        if flags & 1:
            raise BinningCapacityExceeded(
                f"rasterize_gaussians_async: {R} (Gaussian, tile) instances exceeded the binning capacity of {w[2]}; "
                "that frame rendered only the background.  The capacity has been raised: render it again.")

    KEEP_IN_FLIGHT = 2  # frames whose flag words may still be unread when the next forward is issued

    @classmethod
    def poll(cls, device=None):
        """Examine what has arrived without blocking, and BLOCK on watches more than KEEP_IN_FLIGHT frames old (their events
        fired long ago, so this costs nothing): at most the last KEEP_IN_FLIGHT frames of a loop can be unverified --
        check_all() / `with AsyncCapacity.frames():` closes that window."""
        pending = list(cls._dev(device if device is not None else "cuda")["pending"])
        for k, w in enumerate(pending):
            cls._examine(w, wait=k < len(pending) - cls.KEEP_IN_FLIGHT)

    @classmethod
    def frames(cls):
        """Context manager for forward-only loops over the sync-free rasterizer: every frame issued inside is verified when the
        block ends (raises for a frame that overflowed)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            try:
                yield cls
            finally:
                cls.check_all()
        return _cm()

    @classmethod
    def check(cls, w):
        cls._examine(w, wait=True)

    @classmethod
    def check_graph_status(cls):
        """Synchronously examine the status words of every graph-captured forward (after one or more replays)."""
        for st in cls.graph_status:
            R, flags = int(st[0].item()) & 0xFFFFFFFF, int(st[1].item())
            if flags & 2:
                raise RuntimeError("rasterize_gaussians_async (graph): " + PREFILTER_MSG)
            if flags & 1:
                raise BinningCapacityExceeded(
                    f"rasterize_gaussians_async (graph): {R} instances exceeded the binning capacity baked into the graph; "
                    "the last replay rendered only the background -- capture again with a larger capacity")

    @classmethod
    def check_all(cls):
        for st in list(cls._state.values()):
            for w in list(st["pending"]):
                cls._examine(w, wait=True)


def rasterize_gaussians_async(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                              projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
                              debug, extra=None, capacity=None, rows_zeroed=False):
    """Sync-free forward (extension): same inputs as rasterize_gaussians, no host read of num_rendered.
    Returns (capacity, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer, out_extra | None, watch) where
    `capacity` takes the place of num_rendered in rasterize_gaussians_backward and `watch` is the deferred overflow check
    (AsyncCapacity.check(watch))."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("rasterize_gaussians_async: tensors must live on a HIP device (no CPU path)")
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    if P == 0:
        out = rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                  projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
                                  debug, extra=extra, rows_zeroed=rows_zeroed)
        return (0,) + tuple(out[1:8]) + ((out[8] if extra is not None else None), None)
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:
        AsyncCapacity.poll(dev)
    cap = int(capacity) if capacity is not None else AsyncCapacity.capacity(P, dev)
    f32, u8 = torch.float32, torch.uint8
    out_color = torch.empty((3, H, W), dtype=f32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=f32, device=dev)
    out_alpha = torch.empty((1, H, W), dtype=f32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    geom = torch.empty((lib.gsr_geometry_bytes(P),), dtype=u8, device=dev)
    img = torch.empty((lib.gsr_image_bytes(W, H),), dtype=u8, device=dev)
    binning = torch.empty((lib.gsr_binning_bytes(cap, W, H),), dtype=u8, device=dev)
    # the (R, flags) words: a device tensor inside a graph capture (examined after replays), pinned host words otherwise
    status = torch.empty((2,), dtype=torch.int32, device=dev) if capturing else AsyncCapacity.status_words(dev)
    out_extra = None
    if extra is not None:
        if tuple(extra.shape) != (P, _lib.N_EXTRA):
            raise RuntimeError(f"extra must have shape (num_points, {_lib.N_EXTRA})")
        extra = _f32c(extra, "extra")
        out_extra = torch.empty((_lib.N_EXTRA, H, W), dtype=f32, device=dev)
    M = sh.size(1) if sh.numel() != 0 else 0
    means3D, colors, opacity = _f32c(means3D, "means3D"), _f32c(colors, "colors"), _f32c(opacity, "opacity")
    scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D")
    (sh, sh_dtype), background = _sh(sh), _f32c(background, "background")
    viewmatrix, projmatrix, campos = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos")
    with torch.cuda.device(dev):
        rc = lib.gsr_rasterize_forward_async_ex(
            geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(), P, int(degree), int(M), ptr(background), W, H, ptr(means3D),
            ptr(sh), ptr(colors), ptr(opacity), ptr(scales), float(scale_modifier), ptr(rotations), ptr(cov3D_precomp),
            ptr(viewmatrix), ptr(projmatrix), ptr(campos), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
            out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr(), radii.data_ptr(),
            int(bool(debug)) | (FWD_ZERO_ROWS if rows_zeroed else 0), status.data_ptr(), ptr(extra), 0 if extra is None else _lib.N_EXTRA,
            None if out_extra is None else out_extra.data_ptr(), sh_dtype, _stream(dev))
        check(rc, "gsr_rasterize_forward_async")
        if capturing:
            watch = None
            AsyncCapacity.graph_status.append(status)
        else:
            watch = AsyncCapacity.watch(status, cap, dev)
    return cap, out_color, out_depth, out_alpha, radii, geom, binning, img, out_extra, watch


class Phase1Loss:
    """The loss render() feeds before the PBR phase (train.py:261-265, utils/loss_utils.py:20-24), to be evaluated FUSED with the
    rasterizer (gsr_phase1_loss_forward / gsr_rasterize_backward_phase1_loss):
        w_image L1_b(image, gt_image) + w_alpha L2_b(alpha, alpha_target) + w_normal L1_b(normal, gt_normal) + w_axis L1_b(axis, gt_normal)
    with the means taken over the pixels inside bound_mask.  gt_image, gt_normal: [3,H,W]; alpha_target (bkgd_mask[0]) and
    bound_mask: [H,W] or [1,H,W] (any dtype; nonzero = inside).  Pass it to gaussian_renderer.render(..., fused_loss=...) (or to
    GaussianRasterizer.forward_multi): the result carries "loss", and its backward forms the image gradients inside the blend
    kernel -- no gradient images, no reduction kernels.  Other terms (SSIM, LPIPS, ...) are computed on the images as usual and
    simply added to it: their image gradients arrive through autograd and are added in the kernel's prologue."""

    def __init__(self, gt_image, gt_normal, alpha_target, bound_mask, w_image=1.0, w_alpha=0.1, w_normal=1.0, w_axis=1.0,
                 normal_triple=0, axis_triple=5):
        f = lambda t, planes: t.detach().to(torch.float32).reshape(planes, t.shape[-2], t.shape[-1]).contiguous()  # noqa: E731
        self.gt_image, self.gt_normal = f(gt_image, 3), f(gt_normal, 3)
        self.alpha_target, self.bound = f(alpha_target, 1), f(bound_mask, 1)
        self.weights = (float(w_image), float(w_alpha), float(w_normal), float(w_axis))
        self.normal_triple, self.axis_triple = int(normal_triple), int(axis_triple)
        self._partials = None

    def struct(self, color, alpha, out_extra, stats, upstream=None):
        s = _lib.Phase1LossStruct()
        s.gt_image, s.gt_normal = self.gt_image.data_ptr(), self.gt_normal.data_ptr()
        s.alpha_target, s.bound = self.alpha_target.data_ptr(), self.bound.data_ptr()
        s.w_image, s.w_alpha, s.w_normal, s.w_axis = self.weights
        s.normal_triple, s.axis_triple = self.normal_triple, self.axis_triple
        s.color, s.alpha, s.extra_images, s.stats = color.data_ptr(), alpha.data_ptr(), out_extra.data_ptr(), stats.data_ptr()
        s.upstream = None if upstream is None else upstream.data_ptr()
        return s


def phase1_loss_forward(spec, color, alpha, out_extra):
    """(loss [0-dim view], stats [8]) of a Phase1Loss over the forward's images; no host synchronisation."""
    dev = color.device
    H, W = color.shape[-2], color.shape[-1]
    for t, shape in ((spec.gt_image, (3, H, W)), (spec.gt_normal, (3, H, W)), (spec.alpha_target, (1, H, W)), (spec.bound, (1, H, W))):
        if tuple(t.shape) != shape or t.device != dev:
            raise RuntimeError(f"Phase1Loss: targets must be {H}x{W} images on {dev}")
    stats = torch.empty((8,), dtype=torch.float32, device=dev)
    if spec._partials is None or spec._partials.device != dev:
        spec._partials = torch.empty((int(lib.gsr_phase1_loss_partials()),), dtype=torch.float32, device=dev)
    st = spec.struct(_f32c(color, "color"), _f32c(alpha, "alpha"), _f32c(out_extra, "out_extra"), stats)
    with torch.cuda.device(dev):
        check(lib.gsr_phase1_loss_forward(W, H, C.byref(st), spec._partials.data_ptr(), _stream(dev)), "gsr_phase1_loss_forward")
    return stats[0], stats


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                 dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas,
                                 debug, out=None, extra=None, dL_dout_extra=None, rows_zeroed=False, phase1=None):
    """RasterizeGaussiansBackwardCUDA (DGR/rasterize_points.cu:122-207).

    `out` (extension, keyword only in practice): dict name -> preallocated contiguous float32 tensor for any of
    means3D / sh / opacity / scales / rotations / cov3D / colors / means2D; the view-parallel trainer passes views of
    one flat all-reduce bucket so gradients are produced in place.
    `extra` [P,18], `dL_dout_extra` (extension): the fused multi-feature blend; dL_dout_extra is a list of six [3,H,W]
    gradient images (None = that image received no gradient and costs nothing), or one [18,H,W] tensor."""
    dev = means3D.device
    P, H, W = means3D.size(0), alphas.size(-2), alphas.size(-1)
    M = sh.size(1) if sh.numel() != 0 else 0
    opts = dict(dtype=torch.float32, device=dev)
    # the backward-preprocess kernel writes every element of the tensors it owns (zeros for culled Gaussians)
    new = torch.zeros if P == 0 else torch.empty
    has_sr = scales.numel() != 0
    out = out or {}

    def _get(name, shape, alloc):
        t = out.get(name)
        if t is None:
            return alloc(shape, **opts)
        if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise RuntimeError(f"out['{name}'] must be a contiguous float32 tensor of shape {tuple(shape)} on {dev}")
        return t

    dL_dmeans3D = _get("means3D", (P, 3), new)
    dL_dmeans2D = _get("means2D", (P, 3), new)
    dL_dcolors = _get("colors", (P, 3), new)
    dL_dconic = new((P, 2, 2), **opts)
    dL_dopacity = _get("opacity", (P, 1), new)
    dL_dcov3D = _get("cov3D", (P, 6), new)
    dL_dsh = _get("sh", (P, M, 3), new)
    # without scales / rotations the reference still returns zero tensors for them (DGR/rasterize_points.cu:159-167); the
    # autograd wrappers of this package ask for None instead (out["lean"]): nobody reads those two fills
    lean = bool(out.get("lean")) and not has_sr
    dL_dscales = None if lean else _get("scales", (P, 3), new if has_sr else torch.zeros)
    dL_drotations = None if lean else _get("rotations", (P, 4), new if has_sr else torch.zeros)
    dL_dextra = None
    if extra is not None:
        extra = _f32c(extra, "extra")
        if isinstance(dL_dout_extra, torch.Tensor):
            dL_dout_extra = [dL_dout_extra[3 * t:3 * t + 3] for t in range(_lib.N_EXTRA // 3)]
        grads_extra = [None if g is None else _f32c(g, "dL_dout_extra") for g in dL_dout_extra]   # kept alive until the launch
        extra_ptrs = (C.c_void_p * (_lib.N_EXTRA // 3))(*[None if g is None else g.data_ptr() for g in grads_extra])
        dL_dextra = new((P, _lib.N_EXTRA), **opts)
    if P != 0:
        means3D, colors = _f32c(means3D, "means3D"), _f32c(colors, "colors")
        scales, rotations, cov3D_precomp = _f32c(scales, "scales"), _f32c(rotations, "rotations"), _f32c(cov3D_precomp, "cov3D")
        (sh, sh_dtype), background, alphas = _sh(sh), _f32c(background, "background"), _f32c(alphas, "alphas")
        viewmatrix, projmatrix, campos = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos")
        opt = lambda t, n: None if t is None else _f32c(t, n)  # noqa: E731  (fused loss: an image without a further gradient is None)
        dL_dout_color, dL_dout_depth = opt(dL_dout_color, "dL_dout_color"), opt(dL_dout_depth, "dL_dout_depth")
        dL_dout_alpha = opt(dL_dout_alpha, "dL_dout_alpha")
        radii = radii.contiguous()
        if phase1 is not None:
            # phase1 = (Phase1Loss, stats [8], upstream dL/dloss (0-dim tensor or None), color [3,H,W], out_extra [18,H,W])
            spec, stats, upstream, p1_color, p1_extra = phase1
            if extra is None:
                raise RuntimeError("the fused phase-1 loss needs the fused multi-feature pass (extra colours)")
            up = None if upstream is None else upstream.detach().to(torch.float32).reshape(1).contiguous()
            st = spec.struct(_f32c(p1_color, "color"), alphas, _f32c(p1_extra, "out_extra"), stats, up)
            with torch.cuda.device(dev):
                rc = lib.gsr_rasterize_backward_phase1_loss(
                    P, int(degree), int(M), int(R), ptr(background), W, H, ptr(means3D), ptr(sh), ptr(colors), ptr(alphas),
                    ptr(scales), float(scale_modifier), ptr(rotations), ptr(cov3D_precomp), ptr(viewmatrix), ptr(projmatrix),
                    ptr(campos), float(tan_fovx), float(tan_fovy), ptr(radii), geomBuffer.data_ptr(),
                    binningBuffer.data_ptr(), imageBuffer.data_ptr(), ptr(dL_dout_color), ptr(dL_dout_depth),
                    ptr(dL_dout_alpha), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(),
                    dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
                    dL_dsh.data_ptr() if M else None, None if dL_dscales is None else dL_dscales.data_ptr(),
                    None if dL_drotations is None else dL_drotations.data_ptr(),
                    int(bool(debug)) | (BWD_ROWS_ZEROED if rows_zeroed else 0), ptr(extra), _lib.N_EXTRA, extra_ptrs,
                    dL_dextra.data_ptr(), sh_dtype, C.byref(st), _stream(dev))
            check(rc, "gsr_rasterize_backward_phase1_loss")
            return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dextra
        with torch.cuda.device(dev):
            rc = lib.gsr_rasterize_backward_ex(
                P, int(degree), int(M), int(R), ptr(background), W, H, ptr(means3D), ptr(sh), ptr(colors), ptr(alphas),
                ptr(scales), float(scale_modifier), ptr(rotations), ptr(cov3D_precomp), ptr(viewmatrix), ptr(projmatrix),
                ptr(campos), float(tan_fovx), float(tan_fovy), ptr(radii), geomBuffer.data_ptr(),
                binningBuffer.data_ptr(), imageBuffer.data_ptr(), ptr(dL_dout_color), ptr(dL_dout_depth),
                ptr(dL_dout_alpha), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(),
                dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
                dL_dsh.data_ptr() if M else None, None if dL_dscales is None else dL_dscales.data_ptr(),
                None if dL_drotations is None else dL_drotations.data_ptr(), int(bool(debug)) | (BWD_ROWS_ZEROED if rows_zeroed else 0),
                ptr(extra), 0 if extra is None else _lib.N_EXTRA, None if extra is None else extra_ptrs,
                None if dL_dextra is None else dL_dextra.data_ptr(), sh_dtype, _stream(dev))
        check(rc, "gsr_rasterize_backward")
    if extra is not None:
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dextra
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (DGR/rasterize_points.cu:209-228)."""
    if not means3D.is_cuda:
        raise RuntimeError("mark_visible: tensors must live on a HIP device (no CPU path)")
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        means3D, viewmatrix, projmatrix = _f32c(means3D, "means3D"), _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix")
        with torch.cuda.device(means3D.device):
            check(lib.gsr_mark_visible(P, ptr(means3D), ptr(viewmatrix), ptr(projmatrix), present.data_ptr(),
                                       _stream(means3D.device)), "gsr_mark_visible")
    return present


def query_state(what, P, R, W, H, geomBuffer, binningBuffer, imageBuffer):
    """Parity-test helper: copy one private scratch array out of the opaque buffers (gsr_query_state)."""
    dev = geomBuffer.device
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    shapes = dict(DEPTHS=((P,), torch.float32), MEANS2D=((P, 2), torch.float32), CONIC_OPACITY=((P, 4), torch.float32),
                  RGB=((P, 3), torch.float32), COV3D=((P, 6), torch.float32), TILES_TOUCHED=((P,), torch.int32),
                  POINT_OFFSETS=((P,), torch.int32), CLAMPED=((P, 3), torch.uint8), POINT_LIST=((R,), torch.int32),
                  KEYS_SORTED=((R,), torch.int64), RANGES=((tiles, 2), torch.int32), FINAL_T=((H, W), torch.float32),
                  N_CONTRIB=((H, W), torch.int32), ORDER=((4 * tiles + 66,), torch.int32))
    shape, dt = shapes[what]
    out = torch.zeros(shape, dtype=dt, device=dev)
    if out.numel():
        with torch.cuda.device(dev):
            check(lib.gsr_query_state(_lib.Q[what], P, R, W, H, geomBuffer.data_ptr(),
                                      binningBuffer.data_ptr() if binningBuffer.numel() else None,
                                      imageBuffer.data_ptr(), out.data_ptr(), _stream(dev)), "gsr_query_state")
    return out
