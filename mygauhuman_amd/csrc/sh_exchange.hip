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

// views: n_views blocks of `stride` floats each: [P*3 masked dL_dRGB | ... ]; the camera position of a view sits at
// blk[cam_off .. +3]; the Gaussian positions are either shared by all views (means3D: static Gaussians, every replica holds
// the same array) or travel with the view (means_off > 0: blk[means_off + 3 i ..], the positions the view's LBS deform
// produced -- articulated Gaussians are posed differently in every view).  SPLIT: the result goes to the model's two SH
// parameter layouts, dc [P][1][3] and rest [P][15][3], instead of one [P][16][3] array.
struct ShViewsArgs {
  int P, D, M, n_views;
  const float *means3D;  // shared positions, or null
  const float *views;
  size_t stride, means_off, cam_off;
  float scale_h;
  const float *dev_scale;
  float *out, *out_rest;
};
template <bool STAGE, bool SPLIT>
__global__ __launch_bounds__(XSH_BLOCK) void sh_grad_from_views_kernel(const ShViewsArgs a) {
  __shared__ __attribute__((aligned(16))) float s_out[STAGE ? XSH_BLOCK * XSH_LDS_ROW : 4];
  const int P = a.P;
  const int i = blockIdx.x * XSH_BLOCK + threadIdx.x;
  const float scale = a.dev_scale ? a.scale_h * a.dev_scale[0] : a.scale_h;  // wave-uniform scalar load
  float acc[48];
#pragma unroll
  for (int k = 0; k < 48; k++) acc[k] = 0.f;
  if (i < P) {
    float mx = 0.f, my = 0.f, mz = 0.f;
    if (a.means3D) mx = a.means3D[3 * (size_t)i], my = a.means3D[3 * (size_t)i + 1], mz = a.means3D[3 * (size_t)i + 2];
    for (int v = 0; v < a.n_views; v++) {
      const float *blk = a.views + (size_t)v * a.stride;
      const float *cam = blk + a.cam_off;
      if (!a.means3D) {
        const float *m = blk + a.means_off + 3 * (size_t)i;
        mx = m[0], my = m[1], mz = m[2];
      }
      const float g0 = blk[3 * (size_t)i], g1 = blk[3 * (size_t)i + 1], g2 = blk[3 * (size_t)i + 2];
      const float dx = mx - cam[0], dy = my - cam[1], dz = mz - cam[2];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      float w[16];
#pragma unroll
      for (int k = 0; k < 16; k++) w[k] = 0.f;
      sh_basis(a.D, dx / len, dy / len, dz / len, w);
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
    for (int k = 0; k < 48; k++) row[k] = scale == 0.f ? 0.f : acc[k] * scale;  // a skipped step is exact zeros, whatever a blank view's block held
    __syncthreads();
    const int first = blockIdx.x * XSH_BLOCK;
    const int nrows = min(XSH_BLOCK, P - first);
    if (SPLIT) {
      // dc: nrows x 3 floats, rest: nrows x 45 floats, both contiguous for the block (256 rows x 180 B is a multiple of 16 B)
      float *dc = a.out + (size_t)first * 3;
      for (int e = threadIdx.x; e < nrows * 3; e += XSH_BLOCK) dc[e] = s_out[(e / 3) * XSH_LDS_ROW + e % 3];
      float *rest = a.out_rest + (size_t)first * 45;
      const int n4 = nrows * 45 / 4;
      float4 *rest4 = reinterpret_cast<float4 *>(rest);
      for (int q = threadIdx.x; q < n4; q += XSH_BLOCK) {
        float vv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int e = 4 * q + j;
          vv[j] = s_out[(e / 45) * XSH_LDS_ROW + 3 + e % 45];
        }
        rest4[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      }
      for (int e = 4 * n4 + (int)threadIdx.x; e < nrows * 45; e += XSH_BLOCK) rest[e] = s_out[(e / 45) * XSH_LDS_ROW + 3 + e % 45];
    } else {
      float4 *out = reinterpret_cast<float4 *>(a.out + (size_t)first * 48);
      for (int q = threadIdx.x; q < nrows * 12; q += XSH_BLOCK) {
        const int r = q / 12, k4 = q % 12;
        out[q] = *reinterpret_cast<const float4 *>(&s_out[r * XSH_LDS_ROW + 4 * k4]);
      }
    }
  } else if (i < P) {
    for (int k = 0; k < a.M * 3; k++) a.out[(size_t)i * a.M * 3 + k] = (k < 48 && scale != 0.f) ? acc[k] * scale : 0.f;
  }
}

// sender side of the articulated (render()) path: one launch fills the rank's whole view block -- masked dL_dRGB, the posed
// positions of this view and its camera position.  Mask: the colour max(SH + 0.5, 0) came out as exactly 0 = that channel was
// clamped (gaussian_renderer/__init__.py:195 `clamp_min(sh2rgb + 0.5, 0.0)`; torch lets the gradient pass where the argument is
// exactly 0 -- a measure-zero case in which all replicas still agree with each other, every rank applies this rule).
__global__ void sh_view_pack_posed_kernel(int P, const float *colors, const float *dL_dcolors, const float *means,
                                          const float *campos, float *blk, size_t means_off, size_t cam_off) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (size_t)P * 3) {
    blk[e] = colors[e] > 0.f ? dL_dcolors[e] : 0.f;
    blk[means_off + e] = means[e];
  }
  if (e < 3) blk[cam_off + e] = campos[e];
}

// after the SUM all-reduce of a step's flat gradient bucket: scale = (some rank overflowed) ? 0 : inv_world, the whole
// bucket times scale, and the bookkeeping words of gsr_step_status phase 1 -- one launch instead of a status kernel + a
// multiply.  The overflow slot itself is not scaled (every workgroup reads it).
__global__ __launch_bounds__(256) void step_finish_kernel(const uint32_t *status, float *flat, size_t n, size_t overflow_index,
                                                          float inv_world, float *scale_out, uint32_t *report) {
  const float ranks = flat[overflow_index];
  const float scale = ranks > 0.f ? 0.f : inv_world;
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n && !(overflow_index >= i && overflow_index < i + 4)) {
      float4 v = *reinterpret_cast<float4 *>(flat + i);
      v.x *= scale, v.y *= scale, v.z *= scale, v.w *= scale;
      *reinterpret_cast<float4 *>(flat + i) = v;
    } else {
      for (size_t k = i; k < n && k < i + 4; k++)
        if (k != overflow_index) flat[k] *= scale;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (scale_out) scale_out[0] = scale;
    if (report) {
      report[0] = (uint32_t)(ranks + 0.5f);
      report[1] = status ? status[0] : 0u;
      report[2] = status ? status[1] : 0u;
    }
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

static int sh_grad_from_views_impl(const char *who, int P, int sh_degree, int M, int n_views, const float *means3D,
                                   const float *views, size_t view_stride, size_t means_off, size_t cam_off, float scale,
                                   const float *dev_scale, float *out, float *out_rest, gsr_stream_t stream_) {
  using namespace gsr;
  const bool posed = means3D == nullptr;
  if (P < 0 || sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1) || M > 16 || n_views < 1 ||
      cam_off + 3 > view_stride || cam_off < (size_t)P * 3 || (posed && (means_off < (size_t)P * 3 || means_off + (size_t)P * 3 > view_stride)) ||
      (P > 0 && (!views || !out)) || (out_rest && M != 16)) {
    set_error("%s: bad arguments", who);
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ShViewsArgs a = {P, sh_degree, M, n_views, means3D, views, view_stride, means_off, cam_off, scale, dev_scale, out, out_rest};
  const dim3 grid((P + XSH_BLOCK - 1) / XSH_BLOCK), block(XSH_BLOCK);
  if (out_rest) {
    if (reinterpret_cast<size_t>(out_rest) % 16 != 0) {
      set_error("%s: the rest array must be 16-byte aligned", who);
      return GSR_EINVAL;
    }
    hipLaunchKernelGGL((sh_grad_from_views_kernel<true, true>), grid, block, 0, stream, a);
  } else if (M == 16 && reinterpret_cast<size_t>(out) % 16 == 0) {
    hipLaunchKernelGGL((sh_grad_from_views_kernel<true, false>), grid, block, 0, stream, a);
  } else {
    hipLaunchKernelGGL((sh_grad_from_views_kernel<false, false>), grid, block, 0, stream, a);
  }
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_sh_grad_from_views(int P, int sh_degree, int M, int n_views, const float *means3D, const float *views,
                           size_t view_stride, float scale, const float *dev_scale, float *dL_dsh, gsr_stream_t stream) {
  if (P > 0 && !means3D) {
    gsr::set_error("gsr_sh_grad_from_views: bad arguments");
    return GSR_EINVAL;
  }
  return sh_grad_from_views_impl("gsr_sh_grad_from_views", P, sh_degree, M, n_views, means3D, views, view_stride, 0, (size_t)(P > 0 ? P : 0) * 3,
                                 scale, dev_scale, dL_dsh, nullptr, stream);
}

int gsr_sh_view_pack_posed(int P, const float *colors, const float *dL_dcolors, const float *means3D_view, const float *campos,
                           float *view_block, size_t means_offset, size_t cam_offset, gsr_stream_t stream_) {
  using namespace gsr;
  if (P < 0 || means_offset < (size_t)(P > 0 ? P : 0) * 3 || cam_offset < means_offset + (size_t)(P > 0 ? P : 0) * 3 ||
      (P > 0 && (!colors || !dL_dcolors || !means3D_view)) || !campos || !view_block) {
    set_error("gsr_sh_view_pack_posed: bad arguments");
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const size_t n = (size_t)P * 3 > 3 ? (size_t)P * 3 : 3;
  hipLaunchKernelGGL(sh_view_pack_posed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P, colors, dL_dcolors,
                     means3D_view, campos, view_block, means_offset, cam_offset);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_sh_grad_from_views_posed(int P, int sh_degree, int n_views, const float *views, size_t view_stride, size_t means_offset,
                                 size_t cam_offset, float scale, const float *dev_scale, float *dL_dsh_dc, float *dL_dsh_rest,
                                 gsr_stream_t stream) {
  return sh_grad_from_views_impl("gsr_sh_grad_from_views_posed", P, sh_degree, 16, n_views, nullptr, views, view_stride, means_offset,
                                 cam_offset, scale, dev_scale, dL_dsh_dc, dL_dsh_rest, stream);
}

int gsr_step_finish(const uint32_t *status, float *bucket, size_t n_floats, size_t overflow_index, float inv_world, float *scale,
                    uint32_t *report, gsr_stream_t stream_) {
  using namespace gsr;
  if (!bucket || overflow_index >= n_floats || reinterpret_cast<size_t>(bucket) % 16 != 0) {
    set_error("gsr_step_finish: bad arguments");
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const size_t groups = (n_floats + 1023) / 1024;
  hipLaunchKernelGGL(step_finish_kernel, dim3((unsigned)(groups < 1024 ? groups : 1024)), dim3(256), 0, stream, status, bucket, n_floats,
                     overflow_index, inv_world, scale, report);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

}  // extern "C"
