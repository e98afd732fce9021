// geometry.hip -- per-Gaussian forward preprocess, visibility, scan, key duplication, tile ranges.
//
// Built with -ffp-contract=off: one IEEE rounding per source-level operation, the same convention as
// oracle/gsr_oracle.c, so radii / tile rects / depth bits / keys / ranges are bit-exact against the oracle.
// GLM column-major products are written out term by term in GLM's evaluation order
// ((A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2]).
//
// Replaces: preprocessCUDA (CR/forward.cu:155-256), checkFrustum (CR/rasterizer_impl.cu:54-66),
// cub::DeviceScan::InclusiveSum (:279), duplicateWithKeys (:70-111), identifyTileRanges (:116-138).
#include "expand.h"
#include "gsr_common.h"
#include "sh_math.h"

namespace gsr {

__device__ __forceinline__ float3 xform4x3(const float3 p, const float *m) {
  return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                     m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(const float3 p, const float *m) {
  return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                     m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

// CR/auxiliary.h:41-44: double arithmetic, one rounding to float
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// Sigma = (S R)^T (S R) with the quaternion used as given (CR/forward.cu:118-152)
__device__ __forceinline__ void cov3d_from_scale_rot(const float3 sc, float mod, const float4 q, float *cov6) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  float R[3][3];
  R[0][0] = 1.f - 2.f * (y * y + z * z);
  R[0][1] = 2.f * (x * y - r * z);
  R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z);
  R[1][1] = 1.f - 2.f * (x * x + z * z);
  R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y);
  R[2][1] = 2.f * (y * z + r * x);
  R[2][2] = 1.f - 2.f * (x * x + y * y);
  const float s[3] = {mod * sc.x, mod * sc.y, mod * sc.z};
  float M[3][3];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int rr = 0; rr < 3; rr++) M[c][rr] = s[rr] * R[c][rr];
#define GSR_SIG(c, rr) (M[rr][0] * M[c][0] + M[rr][1] * M[c][1] + M[rr][2] * M[c][2])
  cov6[0] = GSR_SIG(0, 0);
  cov6[1] = GSR_SIG(0, 1);
  cov6[2] = GSR_SIG(0, 2);
  cov6[3] = GSR_SIG(1, 1);
  cov6[4] = GSR_SIG(1, 2);
  cov6[5] = GSR_SIG(2, 2);
#undef GSR_SIG
}

// EWA 2D covariance (CR/forward.cu:74-113); returns (a, b, c) with the 0.3 dilation
__device__ __forceinline__ float3 cov2d(const float3 mean, float fx, float fy, float tanx, float tany, const float *c6,
                                        const float *vm) {
  float3 t = xform4x3(mean, vm);
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  const float txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
  const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
  const float W0[3] = {vm[0], vm[4], vm[8]}, W1[3] = {vm[1], vm[5], vm[9]}, W2[3] = {vm[2], vm[6], vm[10]};
  float T0[3], T1[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    T0[r] = W0[r] * J00 + W2[r] * J02;
    T1[r] = W1[r] * J11 + W2[r] * J12;
  }
  const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  float A[3][2];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    A[c][0] = T0[0] * V[c][0] + T0[1] * V[c][1] + T0[2] * V[c][2];
    A[c][1] = T1[0] * V[c][0] + T1[1] * V[c][1] + T1[2] * V[c][2];
  }
  const float c00 = A[0][0] * T0[0] + A[1][0] * T0[1] + A[2][0] * T0[2];
  const float c01 = A[0][1] * T0[0] + A[1][1] * T0[1] + A[2][1] * T0[2];
  const float c11 = A[0][1] * T1[0] + A[1][1] * T1[1] + A[2][1] * T1[2];
  return make_float3(c00 + 0.3f, c01, c11 + 0.3f);
}

constexpr int SH_M = 16;           // coefficients per Gaussian at SH degree 3
constexpr int SH_ROW = SH_M * 3;   // floats per Gaussian
constexpr int SH_HALF = SH_ROW / 2;  // floats staged at a time: 8 coefficients
constexpr int SH_LDS_ROW = 28;     // padded LDS row (112 B = 7 x 16 B, odd: conflict-free ds_read_b128)
constexpr int SH_CPT = SH_HALF / 4;  // 16-byte chunks per row half = chunks per thread per half

// STAGE_SH: the workgroup's 256 x 192-byte SH block is one contiguous 48 KB slab; it is fetched with coalesced 16-byte loads
// into LDS and each thread then evaluates its own row out of LDS (a thread-per-row global access touches 64 different cache
// lines per instruction and uses 4 bytes of each).  The slab goes through LDS in two halves of 8 coefficients (the second
// half waits in registers meanwhile): 28 KB instead of 52 KB per workgroup = 5 resident workgroups per CU instead of 3.  With
// 3, the 782 workgroups of 200k Gaussians were 14 more than the chip holds (768), and that second round cost 6 of 21 us.
template <bool STAGE_SH>
__global__ __launch_bounds__(PRE_BLOCK) void preprocess_forward_kernel(const PreprocessArgs a) {
  __shared__ uint32_t wave_tot[PRE_BLOCK / WAVE];
  __shared__ __attribute__((aligned(16))) float s_sh[STAGE_SH ? PRE_BLOCK * SH_LDS_ROW : 4];
  const int i = blockIdx.x * PRE_BLOCK + threadIdx.x;
  float4 second[STAGE_SH ? SH_CPT : 1];  // this thread's chunks of the second half (fp16 storage: 3 x 8 halves in [0..2])
  if (STAGE_SH) {
    const int first = blockIdx.x * PRE_BLOCK;
    const int nrows = min(PRE_BLOCK, a.P - first);
    if (a.sh_half) {
      // fp16 storage: 96-byte rows, 16-byte chunks of 8 halves, widened to fp32 on the way into LDS (exact)
      const uint4 *slab = reinterpret_cast<const uint4 *>(reinterpret_cast<const _Float16 *>(a.shs) + (size_t)first * SH_ROW);
#pragma unroll
      for (int j = 0; j < SH_CPT / 2; j++) {
        const int e = (int)threadIdx.x + j * PRE_BLOCK, row = e / (SH_CPT / 2), c8 = e % (SH_CPT / 2);
        if (row < nrows) {
          const uint4 v = slab[row * (SH_ROW / 8) + c8];
          second[j] = __builtin_bit_cast(float4, slab[row * (SH_ROW / 8) + SH_CPT / 2 + c8]);
          const _Float16 *hv = reinterpret_cast<const _Float16 *>(&v);
          float *dst = &s_sh[row * SH_LDS_ROW + 8 * c8];
#pragma unroll
          for (int e2 = 0; e2 < 8; e2++) dst[e2] = (float)hv[e2];
        }
      }
    } else {
      const float4 *slab = reinterpret_cast<const float4 *>(a.shs + (size_t)first * SH_ROW);
#pragma unroll
      for (int j = 0; j < SH_CPT; j++) {
        const int e = (int)threadIdx.x + j * PRE_BLOCK, row = e / SH_CPT, c4 = e % SH_CPT;
        if (row < nrows) {
          *reinterpret_cast<float4 *>(&s_sh[row * SH_LDS_ROW + 4 * c4]) = slab[row * (SH_ROW / 4) + c4];
          second[j] = slab[row * (SH_ROW / 4) + SH_CPT + c4];
        }
      }
    }
  }
  uint32_t tiles = 0;
  int my_radius = 0;
  SplatRec rec = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint32_t clamp_bits = 0;
  bool want_sh = false;  // STAGE_SH: this Gaussian's colour comes from the staged halves below
  float3 p = make_float3(0.f, 0.f, 0.f);
  if (i < a.P) {
    p = make_float3(a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]);
    const float3 pv = xform4x3(p, a.view);
    if (pv.z <= 0.2f) {  // CR/auxiliary.h:154
      // The reference prints "Point is filtered although prefiltered is set" and traps the whole context here
      // (CR/auxiliary.h:156-160).  A trap aborts the queue and the process; instead the point is culled like any other and a
      // flag word next to R is raised, which the host turns into GSR_EINVAL / a RuntimeError with the reference's message.
      if (a.prefiltered) atomicOr(&a.geom.total[1], 1u);
    } else {
      const float4 ph = xform4x4(p, a.proj);
      const float pw = 1.0f / (ph.w + 0.0000001f);
      const float pprojx = ph.x * pw, pprojy = ph.y * pw;
      float c6[6];
      if (a.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * (size_t)i + k];
      } else {
        const float3 sc = make_float3(a.scales[3 * i], a.scales[3 * i + 1], a.scales[3 * i + 2]);
        const float4 q = make_float4(a.rotations[4 * i], a.rotations[4 * i + 1], a.rotations[4 * i + 2], a.rotations[4 * i + 3]);
        cov3d_from_scale_rot(sc, a.scale_modifier, q, c6);
#pragma unroll
        for (int k = 0; k < 6; k++) a.geom.cov3D[6 * (size_t)i + k] = c6[k];
      }
      const float3 cv = cov2d(p, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c6, a.view);
      const float det = cv.x * cv.z - cv.y * cv.y;
      if (det != 0.0f) {
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (cv.x + cv.z);
        const float l1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float radf = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        const float pix = ndc2pix(pprojx, a.W), piy = ndc2pix(pprojy, a.H);
        int x0, y0, x1, y1;
        const int radi = f2i_sat(radf);
        tile_rect(pix, piy, radi, a.grid_x, a.grid_y, x0, y0, x1, y1);
        const uint32_t area = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
        if (area != 0) {
          float3 rgb = make_float3(0.f, 0.f, 0.f);
          if (a.colors_precomp) {
            rgb = make_float3(a.colors_precomp[3 * (size_t)i], a.colors_precomp[3 * (size_t)i + 1], a.colors_precomp[3 * (size_t)i + 2]);
          } else if (STAGE_SH) {
            want_sh = true;
          } else {
            rgb = sh_to_rgb(a.D, p, a.campos, a.shs + (size_t)i * a.M * 3, clamp_bits);
          }
          my_radius = radi;
          tiles = area;
          rec.x = pix;
          rec.y = piy;
          rec.conic_a = cv.z * det_inv;
          rec.conic_b = -cv.y * det_inv;
          rec.conic_c = cv.x * det_inv;
          rec.opacity = a.opacities[i];
          rec.depth = pv.z;
          rec.r = rgb.x;
          rec.g = rgb.y;
          rec.b = rgb.z;
          // Conservative cull box for the blend kernels (never changes a result): a pixel can only pass the
          // alpha >= 1/255 test if power >= -ln(255*opacity), i.e. inside the ellipse d^T conic d <= 2 tau, whose
          // bounding box has half-extents sqrt(2 tau cov_xx), sqrt(2 tau cov_yy) (cov = inverse conic).
          const float o = rec.opacity, t255 = 255.0f * o;
          float hx = __builtin_inff(), hy = __builtin_inff();
          if (o == o) {
            if (!(t255 >= 1.0f)) {
              hx = hy = -1.0e30f;  // alpha <= opacity < 1/255 everywhere
            } else if (cv.x > 0.f && cv.z > 0.f && det > 0.f) {
              const float tau2 = 2.0f * (logf(t255) + 0.02f);
              hx = sqrtf(tau2 * cv.x) * 1.0001f + 0.01f;
              hy = sqrtf(tau2 * cv.z) * 1.0001f + 0.01f;
            }
          }
          rec.hx = hx;
          rec.hy = hy;
        }
      }
    }
  }
  if (STAGE_SH) {
    float3 d0;
    const ShDir dir = sh_dir(p, a.campos, d0);
    float res[3] = {0.f, 0.f, 0.f};
    __syncthreads();  // first half staged
    if (want_sh) sh_accumulate<0, SH_M / 2>(a.D, dir, &s_sh[threadIdx.x * SH_LDS_ROW], res);
    __syncthreads();  // everybody is done reading it
    {
      const int nrows = min(PRE_BLOCK, a.P - blockIdx.x * PRE_BLOCK);
      if (a.sh_half) {
#pragma unroll
        for (int j = 0; j < SH_CPT / 2; j++) {
          const int e = (int)threadIdx.x + j * PRE_BLOCK, row = e / (SH_CPT / 2), c8 = e % (SH_CPT / 2);
          if (row < nrows) {
            const _Float16 *hv = reinterpret_cast<const _Float16 *>(&second[j]);
            float *dst = &s_sh[row * SH_LDS_ROW + 8 * c8];
#pragma unroll
            for (int e2 = 0; e2 < 8; e2++) dst[e2] = (float)hv[e2];
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < SH_CPT; j++) {
          const int e = (int)threadIdx.x + j * PRE_BLOCK, row = e / SH_CPT, c4 = e % SH_CPT;
          if (row < nrows) *reinterpret_cast<float4 *>(&s_sh[row * SH_LDS_ROW + 4 * c4]) = second[j];
        }
      }
    }
    __syncthreads();
    if (want_sh) {
      sh_accumulate<SH_M / 2, SH_M>(a.D, dir, &s_sh[threadIdx.x * SH_LDS_ROW], res);
      const float3 rgb = sh_finish(res, clamp_bits);
      rec.r = rgb.x;
      rec.g = rgb.y;
      rec.b = rgb.z;
    }
  }
  if (i < a.P) {
    a.radii[i] = my_radius;
    a.geom.tiles_touched[i] = tiles;
    a.geom.clamped[i] = (uint8_t)clamp_bits;
    float4 *dst = reinterpret_cast<float4 *>(a.geom.recs + i);
    dst[0] = make_float4(rec.x, rec.y, rec.conic_a, rec.conic_b);
    dst[1] = make_float4(rec.conic_c, rec.opacity, rec.depth, rec.r);
    dst[2] = make_float4(rec.g, rec.b, rec.hx, rec.hy);
    if (a.zero_rows) {  // GSR_FWD_ZERO_ROWS: the backward that follows finds its accumulation rows zero (no fill kernel there)
      float4 *row = reinterpret_cast<float4 *>(a.geom.grad_rows + (size_t)i * GROWX);
#pragma unroll
      for (int k = 0; k < GROWX / 4; k++) row[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // block-local inclusive scan of tiles_touched (first level of the device-wide scan)
  const uint32_t incl_w = wave_incl_scan(tiles);
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  if (lane == WAVE - 1) wave_tot[wave] = incl_w;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < PRE_BLOCK / WAVE; w++)
    if (w < wave) base += wave_tot[w];
  const uint32_t incl = base + incl_w;
  if (i < a.P) a.geom.block_incl[i] = incl;
  if (threadIdx.x == PRE_BLOCK - 1) a.geom.block_sums[blockIdx.x] = incl;
}

int launch_preprocess_forward(const PreprocessArgs &a, hipStream_t stream) {
  if (a.P <= 0) return GSR_OK;
  const bool stage = a.shs && !a.colors_precomp && a.M == SH_M && (reinterpret_cast<size_t>(a.shs) % 16 == 0);
  if (a.sh_half && a.shs && !a.colors_precomp && !stage) {
    set_error("fp16 SH storage needs 16 coefficients per Gaussian and a 16-byte aligned array");
    return GSR_EINVAL;
  }
  if (stage)
    hipLaunchKernelGGL(preprocess_forward_kernel<true>, dim3(pre_blocks(a.P)), dim3(PRE_BLOCK), 0, stream, a);
  else
    hipLaunchKernelGGL(preprocess_forward_kernel<false>, dim3(pre_blocks(a.P)), dim3(PRE_BLOCK), 0, stream, a);
  return GSR_OK;
}

__global__ void mark_visible_kernel(int P, const float *means3D, const float *view, uint8_t *present) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
  const float z = view[2] * p.x + view[6] * p.y + view[10] * p.z + view[14];
  present[i] = !(z <= 0.2f);
}
int launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t stream) {
  if (P <= 0) return GSR_OK;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, view, present);
  return GSR_OK;
}

// second level of the scan: exclusive prefix of the per-block sums, total -> *total
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(const uint32_t *sums, uint32_t *prefix, uint32_t *total, int n) {
  __shared__ uint32_t wtot[1024 / WAVE];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < n ? sums[i] : 0;
    const uint32_t incl_w = wave_incl_scan(v);
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    if (lane == WAVE - 1) wtot[wave] = incl_w;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wtot[w];
    const uint32_t carry = carry_s;
    if (i < n) prefix[i] = carry + woff + incl_w - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl_w;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}
int launch_scan_block_sums(const GeomState &g, int P, hipStream_t stream) {
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, g.block_sums, g.block_prefix, g.total,
                     pre_blocks(P));
  return GSR_OK;
}

// Key duplication (CR/rasterizer_impl.cu:70-111), load-balanced (expand.h): key/value stores are fully coalesced.
__global__ __launch_bounds__(PRE_BLOCK) void duplicate_kernel(const GeomState g, const int *radii, int P, int gx, int gy,
                                                             uint64_t *keys, uint32_t *vals) {
  expand_block_instances(g, radii, P, gx, gy, true, [&](uint32_t inst, uint32_t gid, uint32_t tile, uint32_t dbits) {
    keys[inst] = ((uint64_t)tile << 32) | (uint64_t)dbits;
    vals[inst] = gid;
  });
}
int launch_duplicate(const GeomState &g, const int *radii, int P, int grid_x, int grid_y, uint64_t *keys, uint32_t *vals,
                     hipStream_t stream) {
  if (P <= 0) return GSR_OK;
  if (grid_x >= 1024 || grid_y >= 1024) {
    set_error("image larger than 16368 px per side is not supported by the packed tile rect");
    return GSR_EINVAL;
  }
  hipLaunchKernelGGL(duplicate_kernel, dim3(pre_blocks(P)), dim3(PRE_BLOCK), 0, stream, g, radii, P, grid_x, grid_y, keys,
                     vals);
  return GSR_OK;
}

// CR/rasterizer_impl.cu:116-138 (ranges zeroed by the caller first, :312)
__global__ void tile_ranges_kernel(size_t R, const uint64_t *keys, uint2 *ranges) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R) return;
  const uint32_t cur = (uint32_t)(keys[idx] >> 32);
  if (idx == 0)
    ranges[cur].x = 0;
  else {
    const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
    if (cur != prev) {
      ranges[prev].y = (uint32_t)idx;
      ranges[cur].x = (uint32_t)idx;
    }
  }
  if (idx == R - 1) ranges[cur].y = (uint32_t)R;
}
int launch_tile_ranges(size_t R, const uint64_t *keys_sorted, uint2 *ranges, size_t tiles, hipStream_t stream) {
  hipError_t e = zero_async(ranges, tiles * sizeof(uint2), stream);
  if (e != hipSuccess) return check_hip(e, "zero_async(ranges)", __FILE__, __LINE__);
  if (R == 0) return GSR_OK;
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, R, keys_sorted, ranges);
  return GSR_OK;
}

// test introspection: unpack SplatRec fields into plain arrays
__global__ void query_recs_kernel(int what, int P, const GeomState g, void *dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const SplatRec r = g.recs[i];
  float *f = reinterpret_cast<float *>(dst);
  switch (what) {
    case GSR_Q_DEPTHS: f[i] = r.depth; break;
    case GSR_Q_MEANS2D: f[2 * i] = r.x; f[2 * i + 1] = r.y; break;
    case GSR_Q_CONIC_OPACITY: f[4 * i] = r.conic_a; f[4 * i + 1] = r.conic_b; f[4 * i + 2] = r.conic_c; f[4 * i + 3] = r.opacity; break;
    case GSR_Q_RGB: f[3 * i] = r.r; f[3 * i + 1] = r.g; f[3 * i + 2] = r.b; break;
    case GSR_Q_CLAMPED: {
      uint8_t *u = reinterpret_cast<uint8_t *>(dst);
      const uint8_t c = g.clamped[i];
      u[3 * i] = c & 1; u[3 * i + 1] = (c >> 1) & 1; u[3 * i + 2] = (c >> 2) & 1;
    } break;
    default: break;
  }
}
int launch_query_recs(int what, int P, const GeomState &g, void *dst, hipStream_t stream) {
  if (P <= 0) return GSR_OK;
  hipLaunchKernelGGL(query_recs_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, what, P, g, dst);
  return GSR_OK;
}

}  // namespace gsr
