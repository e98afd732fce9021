// sh_exchange.hip -- compact exchange of the SH-coefficient gradient between view-parallel ranks.
//
// dL_dsh of one view is rank one per Gaussian: dL_dsh[k][c] = w_k(dir) * dL_dRGB[c] with dir = normalise(mean - campos) and
// dL_dRGB zeroed on clamped channels (CR/backward.cu:40-116).  All-reducing it moves 12 M floats per Gaussian (192 B at
// M = 16, 81 % of the whole gradient payload).  Instead every rank contributes its masked dL_dRGB (12 B per Gaussian) and its
// camera position through ONE all-gather, and each rank rebuilds  sum_views w_k(dir_v) dL_dRGB_v  locally: the same
// numbers (fixed summation order, so replicas stay bit-identical) for 1/16 of the bytes on the xGMI links.
//   sh_view_pack:        out[i] = clamped(i, c) ? 0 : dL_dcolor[i][c]                      (sender side, P x 3 floats)
//   sh_grad_from_views:  dL_dsh[i][k][c] = scale * sum_v w_k(dir_v(i)) * packed_v[i][c]     (receiver side)
#include "gsr_common.h"
#include "sh_math.h"

namespace gsr {

__global__ void sh_view_pack_kernel(int P, const uint8_t *clamped, const float *dL_dcolor, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const uint32_t bits = clamped[i];
#pragma unroll
  for (int c = 0; c < 3; c++) out[3 * (size_t)i + c] = ((bits >> c) & 1u) ? 0.f : dL_dcolor[3 * (size_t)i + c];
}

constexpr int XSH_BLOCK = 256;
constexpr int XSH_LDS_ROW = 52;  // 48 floats + pad: 16-byte aligned, conflict-free b128 rows

// views: n_views blocks of `stride` floats each: [P*3 masked dL_dRGB | campos xyz | pad]
template <bool STAGE>
__global__ __launch_bounds__(XSH_BLOCK) void sh_grad_from_views_kernel(int P, int D, int M, int n_views, const float *means3D,
                                                                      const float *views, size_t stride, float scale_h,
                                                                      const float *dev_scale, float *dL_dsh) {
  __shared__ __attribute__((aligned(16))) float s_out[STAGE ? XSH_BLOCK * XSH_LDS_ROW : 4];
  const int i = blockIdx.x * XSH_BLOCK + threadIdx.x;
  const float scale = dev_scale ? scale_h * dev_scale[0] : scale_h;  // wave-uniform scalar load
  float acc[48];
#pragma unroll
  for (int k = 0; k < 48; k++) acc[k] = 0.f;
  if (i < P) {
    const float mx = means3D[3 * (size_t)i], my = means3D[3 * (size_t)i + 1], mz = means3D[3 * (size_t)i + 2];
    for (int v = 0; v < n_views; v++) {
      const float *blk = views + (size_t)v * stride;
      const float *cam = blk + (size_t)P * 3;
      const float g0 = blk[3 * (size_t)i], g1 = blk[3 * (size_t)i + 1], g2 = blk[3 * (size_t)i + 2];
      const float dx = mx - cam[0], dy = my - cam[1], dz = mz - cam[2];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      float w[16];
#pragma unroll
      for (int k = 0; k < 16; k++) w[k] = 0.f;
      sh_basis(D, dx / len, dy / len, dz / len, w);
#pragma unroll
      for (int k = 0; k < 16; k++) {
        acc[3 * k] += w[k] * g0;
        acc[3 * k + 1] += w[k] * g1;
        acc[3 * k + 2] += w[k] * g2;
      }
    }
  }
  if (STAGE) {
    float *row = &s_out[threadIdx.x * XSH_LDS_ROW];
#pragma unroll
    for (int k = 0; k < 48; k++) row[k] = acc[k] * scale;
    __syncthreads();
    const int first = blockIdx.x * XSH_BLOCK;
    const int nrows = min(XSH_BLOCK, P - first);
    float4 *out = reinterpret_cast<float4 *>(dL_dsh + (size_t)first * 48);
    for (int q = threadIdx.x; q < nrows * 12; q += XSH_BLOCK) {
      const int r = q / 12, k4 = q % 12;
      out[q] = *reinterpret_cast<const float4 *>(&s_out[r * XSH_LDS_ROW + 4 * k4]);
    }
  } else if (i < P) {
    for (int k = 0; k < M * 3; k++) dL_dsh[(size_t)i * M * 3 + k] = k < 48 ? acc[k] * scale : 0.f;
  }
}

// one thread: bookkeeping of a view-parallel step around the gradient all-reduce (see gsr_step_status in gsr.h)
__global__ void step_status_kernel(int phase, const uint32_t *status, float *overflow_slot, float inv_world, float *scale,
                                   uint32_t *report) {
  if (phase == 0 || phase == 2) overflow_slot[0] = status[1] ? 1.f : 0.f;
  if (phase == 1 || phase == 2) {
    const float ranks = overflow_slot[0];
    scale[0] = ranks > 0.f ? 0.f : inv_world;
    report[0] = (uint32_t)(ranks + 0.5f);
    report[1] = status[0];
    report[2] = status[1];
  }
}

}  // namespace gsr

extern "C" {

int gsr_step_status(int phase, const uint32_t *status, float *overflow_slot, float inv_world, float *scale, uint32_t *report,
                    gsr_stream_t stream_) {
  using namespace gsr;
  if (phase < 0 || phase > 2 || !status || !overflow_slot || (phase != 0 && (!scale || !report))) {
    set_error("gsr_step_status: bad arguments");
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(step_status_kernel, dim3(1), dim3(1), 0, stream, phase, status, overflow_slot, inv_world, scale, report);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_sh_view_pack(int P, const char *geom_buffer, const float *dL_dcolor, float *packed, gsr_stream_t stream_) {
  using namespace gsr;
  if (P < 0 || (P > 0 && (!geom_buffer || !dL_dcolor || !packed))) {
    set_error("gsr_sh_view_pack: bad arguments");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  GeomState geom = geom_from_chunk(const_cast<char *>(geom_buffer), (size_t)P);
  hipLaunchKernelGGL(sh_view_pack_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, geom.clamped, dL_dcolor, packed);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_sh_grad_from_views(int P, int sh_degree, int M, int n_views, const float *means3D, const float *views,
                           size_t view_stride, float scale, const float *dev_scale, float *dL_dsh, gsr_stream_t stream_) {
  using namespace gsr;
  if (P < 0 || sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1) || M > 16 || n_views < 1 ||
      view_stride < (size_t)P * 3 + 3 || (P > 0 && (!means3D || !views || !dL_dsh))) {
    set_error("gsr_sh_grad_from_views: bad arguments");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const bool stage = M == 16 && reinterpret_cast<size_t>(dL_dsh) % 16 == 0;
  const dim3 grid((P + XSH_BLOCK - 1) / XSH_BLOCK), block(XSH_BLOCK);
  if (stage)
    hipLaunchKernelGGL(sh_grad_from_views_kernel<true>, grid, block, 0, stream, P, sh_degree, M, n_views, means3D, views,
                       view_stride, scale, dev_scale, dL_dsh);
  else
    hipLaunchKernelGGL(sh_grad_from_views_kernel<false>, grid, block, 0, stream, P, sh_degree, M, n_views, means3D, views,
                       view_stride, scale, dev_scale, dL_dsh);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

}  // extern "C"
