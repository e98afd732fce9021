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

// ---- the phase-1 training loss of render() (train.py:261-265), value: per-workgroup partial sums, then one workgroup finishes
constexpr int P1_BLOCKS = 1024, P1_TERMS = 5;  // |image - gt|, (alpha - target)^2, |normal - gtn|, |axis - gtn|, n_bound

__global__ __launch_bounds__(256) void phase1_loss_partial_kernel(int npix, const gsr_phase1_loss l, float *__restrict__ partials) {
  float acc[P1_TERMS] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const float *nrm = l.extra_images + (size_t)(3 * l.normal_triple) * npix, *axs = l.extra_images + (size_t)(3 * l.axis_triple) * npix;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    if (l.bound[i] != 0.f) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float g = l.gt_normal[(size_t)c * npix + i];
        acc[0] += fabsf(l.color[(size_t)c * npix + i] - l.gt_image[(size_t)c * npix + i]);
        acc[2] += fabsf(nrm[(size_t)c * npix + i] - g);
        acc[3] += fabsf(axs[(size_t)c * npix + i] - g);
      }
      const float da = l.alpha[i] - l.alpha_target[i];
      acc[1] += da * da;
      acc[4] += 1.f;
    }
  }
  __shared__ float s[P1_TERMS][4];
#pragma unroll
  for (int t = 0; t < P1_TERMS; t++) {
    float v = acc[t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    if (threadIdx.x % WAVE == 0) s[t][threadIdx.x / WAVE] = v;
  }
  __syncthreads();
  if (threadIdx.x < P1_TERMS) partials[blockIdx.x * 8 + threadIdx.x] = (s[threadIdx.x][0] + s[threadIdx.x][1]) + (s[threadIdx.x][2] + s[threadIdx.x][3]);
}
__global__ __launch_bounds__(256) void phase1_loss_finish_kernel(int blocks, const float *__restrict__ partials, gsr_phase1_loss l) {
  __shared__ double s[P1_TERMS][4];
  double acc[P1_TERMS] = {0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < blocks; b += 256)
#pragma unroll
    for (int t = 0; t < P1_TERMS; t++) acc[t] += (double)partials[b * 8 + t];
#pragma unroll
  for (int t = 0; t < P1_TERMS; t++) {
    double v = acc[t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    if (threadIdx.x % WAVE == 0) s[t][threadIdx.x / WAVE] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[P1_TERMS];
#pragma unroll
    for (int t = 0; t < P1_TERMS; t++) tot[t] = (s[t][0] + s[t][1]) + (s[t][2] + s[t][3]);
    const double nb = tot[4] > 0 ? tot[4] : 1.0;  // (an empty bound mask: every term is zero, the scales must stay finite)
    const double li = tot[0] / (3.0 * nb), la = tot[1] / nb, ln = tot[2] / (3.0 * nb), lx = tot[3] / (3.0 * nb);
    l.stats[0] = (float)(l.w_image * li + l.w_alpha * la + l.w_normal * ln + l.w_axis * lx);
    l.stats[1] = (float)tot[4];
    l.stats[2] = (float)(1.0 / (3.0 * nb));
    l.stats[3] = (float)(1.0 / nb);
    l.stats[4] = (float)li, l.stats[5] = (float)la, l.stats[6] = (float)ln, l.stats[7] = (float)lx;
  }
}

}  // namespace gsr

extern "C" size_t gsr_phase1_loss_partials(void) { return (size_t)gsr::P1_BLOCKS * 8; }

extern "C" int gsr_phase1_loss_forward(int width, int height, const gsr_phase1_loss *l, float *partials, gsr_stream_t stream_) {
  if (width <= 0 || height <= 0 || !l || !partials || !l->gt_image || !l->gt_normal || !l->alpha_target || !l->bound || !l->color ||
      !l->alpha || !l->extra_images || !l->stats || l->normal_triple < 0 || l->normal_triple > 5 || l->axis_triple < 0 ||
      l->axis_triple > 5) {
    gsr::set_error("gsr_phase1_loss_forward: bad arguments (images, targets, bound mask, stats and two triple indices in 0..5 are required)");
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int npix = width * height;
  const int blocks = (npix + 255) / 256 < gsr::P1_BLOCKS ? (npix + 255) / 256 : gsr::P1_BLOCKS;
  hipLaunchKernelGGL(gsr::phase1_loss_partial_kernel, dim3(blocks), dim3(256), 0, stream, npix, *l, partials);
  hipLaunchKernelGGL(gsr::phase1_loss_finish_kernel, dim3(1), dim3(256), 0, stream, blocks, partials, *l);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

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
