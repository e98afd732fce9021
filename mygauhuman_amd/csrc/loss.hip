// loss.hip -- fused gradient of the bench's "alpha-mask loss" L = mean|color - gt| + lambda * mean (alpha - mask)^2
// (the reference's L1(image, gt) + 0.1 * l2(alpha, mask), train.py:261-262, utils/loss_utils.py:20-24): one pass
// over the pixels instead of five elementwise torch kernels.
#include "gsr_common.h"

namespace gsr {

__global__ __launch_bounds__(256) void alpha_mask_loss_bwd_kernel(int npix, const float *__restrict__ color,
                                                                 const float *__restrict__ alpha, const float *__restrict__ gt,
                                                                 const float *__restrict__ mask, float lambda,
                                                                 float *__restrict__ dcolor, float *__restrict__ dalpha) {
  const float sc = 1.0f / (3.0f * (float)npix), sa = 2.0f * lambda / (float)npix;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float d = color[(size_t)c * npix + i] - gt[(size_t)c * npix + i];
      dcolor[(size_t)c * npix + i] = d > 0.f ? sc : (d < 0.f ? -sc : 0.f);
    }
    dalpha[i] = sa * (alpha[i] - mask[i]);
  }
}

}  // namespace gsr

extern "C" int gsr_alpha_mask_loss_backward(int width, int height, const float *color, const float *alpha, const float *gt,
                                            const float *mask, float lambda_alpha, float *dL_dcolor, float *dL_dalpha,
                                            gsr_stream_t stream_) {
  if (width <= 0 || height <= 0 || !color || !alpha || !gt || !mask || !dL_dcolor || !dL_dalpha) {
    gsr::set_error("gsr_alpha_mask_loss_backward: bad arguments");
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int npix = width * height;
  const int blocks = (npix + 255) / 256 < 2048 ? (npix + 255) / 256 : 2048;
  hipLaunchKernelGGL(gsr::alpha_mask_loss_bwd_kernel, dim3(blocks), dim3(256), 0, stream, npix, color, alpha, gt, mask,
                     lambda_alpha, dL_dcolor, dL_dalpha);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}
