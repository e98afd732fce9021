"""Sync-free rasterizer session for training loops (extension over the reference's operator API).

The reference operator blocks the host every forward to read num_rendered back (CR/rasterizer_impl.cu:283) and
re-allocates every output and scratch buffer per call (DGR/rasterize_points.cu:69-81,159-167).  A RasterSession owns
all of that memory for a fixed problem shape and drives gsr_rasterize_forward_async / gsr_rasterize_backward /
gsr_alpha_mask_loss_backward directly: no host<->device synchronisation, no allocations and no fill kernels inside
a step, so the host can run ahead of the GPU.  The binning buffer is sized for `capacity` instances; the device
reports R and an overflow flag in `status` (checked with .overflowed(), which does synchronise).
"""
import torch

from ._lib import SH_F16, SH_F32, check, lib

ROWS_ZEROED = 2  # GSR_BWD_ROWS_ZEROED (include/gsr.h)


class RasterSession:
    def __init__(self, P, W, H, M, device, capacity, with_backward=True):
        self.P, self.W, self.H, self.M, self.device = int(P), int(W), int(H), int(M), torch.device(device)
        self.capacity = int(capacity)
        dev, u8, f32 = self.device, torch.uint8, torch.float32
        # zero-filled once: the backward calls of this session carry GSR_BWD_ROWS_ZEROED (the gradient rows inside this buffer
        # are cleared by the backward that consumed them, no memset per frame)
        self.geom = torch.zeros(lib.gsr_geometry_bytes(self.P), dtype=u8, device=dev)
        self.img = torch.empty(lib.gsr_image_bytes(self.W, self.H), dtype=u8, device=dev)
        self.bin = torch.empty(lib.gsr_binning_bytes(self.capacity, self.W, self.H), dtype=u8, device=dev)
        self.color = torch.empty((3, self.H, self.W), dtype=f32, device=dev)
        self.depth = torch.empty((1, self.H, self.W), dtype=f32, device=dev)
        self.alpha = torch.empty((1, self.H, self.W), dtype=f32, device=dev)
        self.radii = torch.empty((self.P,), dtype=torch.int32, device=dev)
        self.status = torch.zeros(2, dtype=torch.int32, device=dev)
        if with_backward:
            self.dL_dcolor = torch.empty((3, self.H, self.W), dtype=f32, device=dev)
            self.dL_dalpha = torch.empty((1, self.H, self.W), dtype=f32, device=dev)
            self.dL_ddepth = torch.zeros((1, self.H, self.W), dtype=f32, device=dev)
            self.dL_dmean2D = torch.empty((self.P, 3), dtype=f32, device=dev)
            self.dL_dconic = torch.empty((self.P, 4), dtype=f32, device=dev)
            self.dL_dcolors = torch.empty((self.P, 3), dtype=f32, device=dev)
            self.dL_dcov3D = torch.empty((self.P, 6), dtype=f32, device=dev)

    @staticmethod
    def calibrated(params, cam, bg, sh_degree, slack=1.3, with_backward=True):
        """Size the session from one synchronous forward of the given view (R known on the host)."""
        from .diff_gaussian_rasterization import _C
        e = torch.empty(0)
        R = _C.rasterize_gaussians(bg, params["means3D"], e, params["opacities"], params["scales"], params["rotations"], 1.0,
                                   e, cam["viewmatrix"], cam["projmatrix"], cam["tanfovx"], cam["tanfovy"], cam["H"], cam["W"],
                                   params["shs"], sh_degree, cam["campos"], False, False)[0]
        P, M = params["means3D"].shape[0], params["shs"].shape[1]
        return RasterSession(P, cam["W"], cam["H"], M, params["means3D"].device, int(R * slack) + 4096, with_backward)

    def regrown(self, capacity):
        """A session of the same shape with a larger binning capacity (the other buffers are reused)."""
        if capacity <= self.capacity:
            return self
        n = RasterSession.__new__(RasterSession)
        n.__dict__.update(self.__dict__)
        n.capacity = int(capacity)
        n.bin = torch.empty(lib.gsr_binning_bytes(n.capacity, n.W, n.H), dtype=torch.uint8, device=n.device)
        return n

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self, params, cam, bg, sh_degree, scale_modifier=1.0):
        """SH + scales/rotations input mode.  Returns (color, depth, alpha, radii) views owned by the session."""
        p = params
        sh_dtype = SH_F16 if p["shs"].dtype == torch.float16 else SH_F32  # fp16 SH storage (extension): widened on load
        check(lib.gsr_rasterize_forward_async_ex(
            self.geom.data_ptr(), self.bin.data_ptr(), self.capacity, self.img.data_ptr(), self.P, int(sh_degree), self.M,
            bg.data_ptr(), self.W, self.H, p["means3D"].data_ptr(), p["shs"].data_ptr(), None, p["opacities"].data_ptr(),
            p["scales"].data_ptr(), float(scale_modifier), p["rotations"].data_ptr(), None, cam["viewmatrix"].data_ptr(),
            cam["projmatrix"].data_ptr(), cam["campos"].data_ptr(), float(cam["tanfovx"]), float(cam["tanfovy"]), 0,
            self.color.data_ptr(), self.depth.data_ptr(), self.alpha.data_ptr(), self.radii.data_ptr(), 0,
            self.status.data_ptr(), None, 0, None, sh_dtype, self._stream()), "gsr_rasterize_forward_async")
        return self.color, self.depth, self.alpha, self.radii

    def alpha_mask_loss_backward(self, gt, mask, lambda_alpha=0.1):
        check(lib.gsr_alpha_mask_loss_backward(self.W, self.H, self.color.data_ptr(), self.alpha.data_ptr(), gt.data_ptr(),
                                               mask.data_ptr(), float(lambda_alpha), self.dL_dcolor.data_ptr(),
                                               self.dL_dalpha.data_ptr(), self._stream()), "gsr_alpha_mask_loss_backward")
        return self.dL_dcolor, self.dL_dalpha

    def backward(self, params, cam, bg, sh_degree, dL_dcolor, dL_ddepth, dL_dalpha, out, scale_modifier=1.0):
        """out: dict with means3D / sh / opacity / scales / rotations gradient tensors (written in place)."""
        p = params
        sh_dtype = SH_F16 if p["shs"].dtype == torch.float16 else SH_F32
        check(lib.gsr_rasterize_backward_ex(
            self.P, int(sh_degree), self.M, self.capacity, bg.data_ptr(), self.W, self.H, p["means3D"].data_ptr(),
            p["shs"].data_ptr(), None, self.alpha.data_ptr(), p["scales"].data_ptr(), float(scale_modifier),
            p["rotations"].data_ptr(), None, cam["viewmatrix"].data_ptr(), cam["projmatrix"].data_ptr(),
            cam["campos"].data_ptr(), float(cam["tanfovx"]), float(cam["tanfovy"]), self.radii.data_ptr(),
            self.geom.data_ptr(), self.bin.data_ptr(), self.img.data_ptr(), dL_dcolor.data_ptr(), dL_ddepth.data_ptr(),
            dL_dalpha.data_ptr(), self.dL_dmean2D.data_ptr(), self.dL_dconic.data_ptr(), out["opacity"].data_ptr(),
            self.dL_dcolors.data_ptr(), out["means3D"].data_ptr(), self.dL_dcov3D.data_ptr(), out["sh"].data_ptr(),
            out["scales"].data_ptr(), out["rotations"].data_ptr(), ROWS_ZEROED, None, 0, None, None, sh_dtype, self._stream()),
            "gsr_rasterize_backward")

    def backward_alpha_mask_loss(self, params, cam, bg, sh_degree, gt, mask, lambda_alpha, out, scale_modifier=1.0):
        """backward() of the loss  mean|color - gt| + lambda_alpha * mean (alpha - mask)^2  of the last forward, with the loss
        gradient formed inside the blend-backward kernel (no separate loss kernel, no gradient images): same gradients, bit
        for bit, as alpha_mask_loss_backward() + backward()."""
        p = params
        sh_dtype = SH_F16 if p["shs"].dtype == torch.float16 else SH_F32
        check(lib.gsr_rasterize_backward_alpha_mask_loss(
            self.P, int(sh_degree), self.M, self.capacity, bg.data_ptr(), self.W, self.H, p["means3D"].data_ptr(),
            p["shs"].data_ptr(), None, self.alpha.data_ptr(), p["scales"].data_ptr(), float(scale_modifier),
            p["rotations"].data_ptr(), None, cam["viewmatrix"].data_ptr(), cam["projmatrix"].data_ptr(),
            cam["campos"].data_ptr(), float(cam["tanfovx"]), float(cam["tanfovy"]), self.radii.data_ptr(),
            self.geom.data_ptr(), self.bin.data_ptr(), self.img.data_ptr(), self.color.data_ptr(), gt.data_ptr(), mask.data_ptr(),
            float(lambda_alpha), self.dL_dmean2D.data_ptr(), self.dL_dconic.data_ptr(), out["opacity"].data_ptr(),
            self.dL_dcolors.data_ptr(), out["means3D"].data_ptr(), self.dL_dcov3D.data_ptr(), out["sh"].data_ptr(),
            out["scales"].data_ptr(), out["rotations"].data_ptr(), ROWS_ZEROED, sh_dtype, self._stream()),
            "gsr_rasterize_backward_alpha_mask_loss")

    # `status` may be a device tensor (default) or a pinned host tensor that the kernels write directly (ViewParallelStep)
    def _status_word(self, k):
        if not self.status.is_cuda:
            torch.cuda.synchronize(self.device)
        return int(self.status[k].item())

    def num_rendered(self):
        """R of the last forward (synchronises)."""
        return self._status_word(0) & 0xFFFFFFFF

    def overflowed(self):
        """True if the last forward needed more than `capacity` instances and therefore rendered nothing (synchronises)."""
        return bool(self._status_word(1) & 1)
