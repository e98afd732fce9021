// knn.hip -- distCUDA2: mean squared distance to the 3 nearest neighbours (replaces SimpleKNN::knn,
// SK/simple_knn.cu:185-221).  Built with -ffp-contract=off so squared distances carry the same roundings as the
// oracle; the search is exact, so the result does not depend on the traversal order.
//
// Same geometric structure as the reference (AABB incl. the origin -> 30-bit Morton codes -> sort -> boxes of 1024
// Morton-consecutive points -> pruned exact search), re-shaped for MI355X:
//   * no host round trips: the AABB stays on the device (the reference syncs twice, SK/simple_knn.cu:197,200);
//   * no per-call allocations: one caller-provided workspace (the reference makes 7 device allocations);
//   * points are gathered into Morton order once, so a candidate box is a contiguous 12 KB run that a workgroup
//     stages into LDS with coalesced loads and every lane then reads by broadcast (the reference gathers
//     points[indices[i]] from global memory inside the innermost loop, SK/simple_knn.cu:175-180);
//   * a workgroup owns 256 Morton-consecutive query points and a box is staged when ANY of them still needs it
//     (__syncthreads_or), each lane keeps the reference's own reject/best[2] test.
#include <float.h>

#include "gsr_common.h"

namespace gsr {

constexpr int KNN_BOX = 1024;  // SK/simple_knn.cu:12 BOX_SIZE
constexpr int KNN_Q = 256;     // query points per workgroup

struct KnnWorkspace {
  float *partial;    // [nred][6]
  float *minmax;     // [6]
  uint32_t *codes, *idx, *codes_s, *idx_s, *tk, *tv, *hist;
  float *sorted;     // [P][3] points in Morton order
  float *boxes;      // [nb][6]
};
static inline int knn_red_blocks(int P) { return min(1024, (P + 1023) / 1024); }
static KnnWorkspace knn_carve(char *p, size_t P) {
  KnnWorkspace w;
  size_t n = P ? P : 1;
  carve(p, w.partial, (size_t)1024 * 6);
  carve(p, w.minmax, 8);
  carve(p, w.codes, n);
  carve(p, w.idx, n);
  carve(p, w.codes_s, n);
  carve(p, w.idx_s, n);
  carve(p, w.tk, n);
  carve(p, w.tv, n);
  carve(p, w.hist, sort_hist_words(n));
  carve(p, w.sorted, n * 3);
  carve(p, w.boxes, ((n + KNN_BOX - 1) / KNN_BOX) * 6);
  return w;
}
size_t knn_workspace_bytes(size_t P) {
  KnnWorkspace w = knn_carve(nullptr, P);
  size_t n = P ? P : 1;
  return reinterpret_cast<size_t>(w.boxes + ((n + KNN_BOX - 1) / KNN_BOX) * 6) + 512;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, WAVE));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, WAVE));
  return v;
}

// AABB of the points and the origin (init = (0,0,0), SK/simple_knn.cu:191-200), two levels
__global__ __launch_bounds__(256) void aabb_partial_kernel(int P, const float *pts, float *partial) {
  __shared__ float s[4][6];
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float v = pts[3 * (size_t)i + k];
      mn[k] = fminf(mn[k], v);
      mx[k] = fmaxf(mx[k], v);
    }
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    mn[k] = wave_min(mn[k]);
    mx[k] = wave_max(mx[k]);
    if (lane == 0) {
      s[wave][k] = mn[k];
      s[wave][3 + k] = mx[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
    partial[blockIdx.x * 6 + threadIdx.x] = v;
  }
}
__global__ void aabb_final_kernel(int nred, const float *partial, float *minmax) {
  const int k = threadIdx.x;
  if (k >= 6) return;
  float v = 0.f;
  for (int b = 0; b < nred; b++) v = k < 3 ? fminf(v, partial[b * 6 + k]) : fmaxf(v, partial[b * 6 + k]);
  minmax[k] = v;
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {  // SK/simple_knn.cu:45-52
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}
__device__ __forceinline__ uint32_t f2u_sat(float f) {  // cvt.rzi.u32.f32: truncate, saturate, NaN -> 0
  if (!(f > 0.f)) return 0u;
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
__global__ void morton_kernel(int P, const float *pts, const float *minmax, uint32_t *codes, uint32_t *idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  uint32_t c[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float mn = minmax[k], mx = minmax[3 + k];
    c[k] = prep_morton(f2u_sat(((pts[3 * (size_t)i + k] - mn) / (mx - mn)) * (float)((1 << 10) - 1)));
  }
  codes[i] = c[0] | (c[1] << 1) | (c[2] << 2);
  idx[i] = (uint32_t)i;
}

__global__ void gather_sorted_kernel(int P, const float *pts, const uint32_t *idx_s, float *sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const uint32_t g = idx_s[i];
  sorted[3 * (size_t)i + 0] = pts[3 * (size_t)g + 0];
  sorted[3 * (size_t)i + 1] = pts[3 * (size_t)g + 1];
  sorted[3 * (size_t)i + 2] = pts[3 * (size_t)g + 2];
}

// AABB of each run of 1024 Morton-consecutive points (SK/simple_knn.cu:78-117)
__global__ __launch_bounds__(256) void box_minmax_kernel(int P, const float *sorted, float *boxes) {
  __shared__ float s[4][6];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  const int lo = blockIdx.x * KNN_BOX, hi = min(P, lo + KNN_BOX);
  for (int i = lo + threadIdx.x; i < hi; i += 256)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float v = sorted[3 * (size_t)i + k];
      mn[k] = fminf(mn[k], v);
      mx[k] = fmaxf(mx[k], v);
    }
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    mn[k] = wave_min(mn[k]);
    mx[k] = wave_max(mx[k]);
    if (lane == 0) {
      s[wave][k] = mn[k];
      s[wave][3 + k] = mx[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s[0][threadIdx.x];
    for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
    boxes[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__device__ __forceinline__ void kbest3(float d, float &b0, float &b1, float &b2) {  // SK/simple_knn.cu:131-145
  if (b0 > d) { const float t = b0; b0 = d; d = t; }
  if (b1 > d) { const float t = b1; b1 = d; d = t; }
  if (b2 > d) { b2 = d; }
}
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = bx - ax, dy = by - ay, dz = bz - az;
  return dx * dx + dy * dy + dz * dz;
}

// SK/simple_knn.cu:147-183
__global__ __launch_bounds__(KNN_Q) void box_mean_dist_kernel(int P, const float *sorted, const uint32_t *idx_s,
                                                              const float *boxes, int nb, float *dists) {
  __shared__ float sx[KNN_BOX], sy[KNN_BOX], sz[KNN_BOX];
  const int idx = blockIdx.x * KNN_Q + threadIdx.x;
  const bool live = idx < P;
  float px = 0, py = 0, pz = 0;
  float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
  if (live) {
    px = sorted[3 * (size_t)idx];
    py = sorted[3 * (size_t)idx + 1];
    pz = sorted[3 * (size_t)idx + 2];
    for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
      if (i == idx) continue;
      kbest3(sqdist3(px, py, pz, sorted[3 * (size_t)i], sorted[3 * (size_t)i + 1], sorted[3 * (size_t)i + 2]), b0, b1, b2);
    }
  }
  const float reject = b2;
  b0 = b1 = b2 = FLT_MAX;
  for (int b = 0; b < nb; b++) {
    bool want = false;
    if (live) {
      const float *bx = boxes + 6 * b;
      float dfx = 0, dfy = 0, dfz = 0;
      if (px < bx[0] || px > bx[3]) dfx = fminf(fabsf(px - bx[0]), fabsf(px - bx[3]));
      if (py < bx[1] || py > bx[4]) dfy = fminf(fabsf(py - bx[1]), fabsf(py - bx[4]));
      if (pz < bx[2] || pz > bx[5]) dfz = fminf(fabsf(pz - bx[2]), fabsf(pz - bx[5]));
      const float dist = dfx * dfx + dfy * dfy + dfz * dfz;
      want = !(dist > reject || dist > b2);
    }
    if (!__syncthreads_or(want ? 1 : 0)) continue;  // also fences the previous iteration's LDS reads
    const int lo = b * KNN_BOX, cnt = min(P - lo, KNN_BOX);
    for (int i = threadIdx.x; i < cnt; i += KNN_Q) {
      sx[i] = sorted[3 * (size_t)(lo + i)];
      sy[i] = sorted[3 * (size_t)(lo + i) + 1];
      sz[i] = sorted[3 * (size_t)(lo + i) + 2];
    }
    __syncthreads();
    if (want) {
      const int self = idx - lo;  // position of this point inside the box, if it is in it
      for (int i = 0; i < cnt; i++) {
        if (i == self) continue;
        kbest3(sqdist3(px, py, pz, sx[i], sy[i], sz[i]), b0, b1, b2);
      }
    }
  }
  if (live) dists[idx_s[idx]] = (b0 + b1 + b2) / 3.0f;
}

// ---- k nearest neighbours WITH indices (k <= 3), the point itself included: what KNN_CUDA returns for ref == query
// (scene/gaussian_model.py:176,573,621,671).  Same box traversal; candidates are ranked by (squared distance, original
// index) lexicographically, so equal distances resolve to the lowest index whatever the traversal order.
__device__ __forceinline__ bool pair_less(float da, uint32_t ia, float db, uint32_t ib) { return da < db || (da == db && ia < ib); }
__device__ __forceinline__ void kbest3_idx(float d, uint32_t id, float (&bd)[3], uint32_t (&bi)[3]) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (pair_less(d, id, bd[k], bi[k])) {
      const float td = bd[k];
      const uint32_t ti = bi[k];
      bd[k] = d;
      bi[k] = id;
      d = td;
      id = ti;
    }
  }
}

__global__ __launch_bounds__(KNN_Q) void box_knn_kernel(int P, int K, const float *sorted, const uint32_t *idx_s, const float *boxes,
                                                        int nb, int *out_idx, float *out_dist) {
  __shared__ float sx[KNN_BOX], sy[KNN_BOX], sz[KNN_BOX];
  __shared__ uint32_t sid[KNN_BOX];
  const int idx = blockIdx.x * KNN_Q + threadIdx.x;
  const bool live = idx < P;
  float px = 0, py = 0, pz = 0;
  float r0 = FLT_MAX, r1 = FLT_MAX, r2 = FLT_MAX;
  if (live) {
    px = sorted[3 * (size_t)idx];
    py = sorted[3 * (size_t)idx + 1];
    pz = sorted[3 * (size_t)idx + 2];
    for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
      if (i == idx) continue;
      kbest3(sqdist3(px, py, pz, sorted[3 * (size_t)i], sorted[3 * (size_t)i + 1], sorted[3 * (size_t)i + 2]), r0, r1, r2);
    }
  }
  const float reject = r2;  // an upper bound of the 3rd-nearest distance (self excluded, so even looser than needed)
  float bd[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  uint32_t bi[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  for (int b = 0; b < nb; b++) {
    bool want = false;
    if (live) {
      const float *bx = boxes + 6 * b;
      float dfx = 0, dfy = 0, dfz = 0;
      if (px < bx[0] || px > bx[3]) dfx = fminf(fabsf(px - bx[0]), fabsf(px - bx[3]));
      if (py < bx[1] || py > bx[4]) dfy = fminf(fabsf(py - bx[1]), fabsf(py - bx[4]));
      if (pz < bx[2] || pz > bx[5]) dfz = fminf(fabsf(pz - bx[2]), fabsf(pz - bx[5]));
      const float dist = dfx * dfx + dfy * dfy + dfz * dfz;
      want = !(dist > reject || dist > bd[2]);  // boxes at exactly the bound are still visited (index tie-break)
    }
    if (!__syncthreads_or(want ? 1 : 0)) continue;
    const int lo = b * KNN_BOX, cnt = min(P - lo, KNN_BOX);
    for (int i = threadIdx.x; i < cnt; i += KNN_Q) {
      sx[i] = sorted[3 * (size_t)(lo + i)];
      sy[i] = sorted[3 * (size_t)(lo + i) + 1];
      sz[i] = sorted[3 * (size_t)(lo + i) + 2];
      sid[i] = idx_s[lo + i];
    }
    __syncthreads();
    if (want)
      for (int i = 0; i < cnt; i++) kbest3_idx(sqdist3(px, py, pz, sx[i], sy[i], sz[i]), sid[i], bd, bi);
  }
  if (live) {
    const size_t o = (size_t)idx_s[idx] * K;
    for (int k = 0; k < K; k++) {
      out_idx[o + k] = (int)bi[k];
      out_dist[o + k] = sqrtf(bd[k]);
    }
  }
}

int knn_self(int P, int K, const float *points, int *out_idx, float *out_dist, char *workspace, hipStream_t stream) {
  if (P <= 0) return GSR_OK;
  KnnWorkspace w = knn_carve(workspace, (size_t)P);
  const int nred = knn_red_blocks(P);
  hipLaunchKernelGGL(aabb_partial_kernel, dim3(nred), dim3(256), 0, stream, P, points, w.partial);
  hipLaunchKernelGGL(aabb_final_kernel, dim3(1), dim3(64), 0, stream, nred, w.partial, w.minmax);
  hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, points, w.minmax, w.codes, w.idx);
  GSR_LAUNCH_CHECK(stream, 0);
  int rc = radix_sort_u32((size_t)P, w.codes, w.idx, w.tk, w.tv, w.codes_s, w.idx_s, 32, w.hist, stream, 0);
  if (rc != GSR_OK) return rc;
  const int nb = (P + KNN_BOX - 1) / KNN_BOX;
  hipLaunchKernelGGL(gather_sorted_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, points, w.idx_s, w.sorted);
  hipLaunchKernelGGL(box_minmax_kernel, dim3(nb), dim3(256), 0, stream, P, w.sorted, w.boxes);
  hipLaunchKernelGGL(box_knn_kernel, dim3((P + KNN_Q - 1) / KNN_Q), dim3(KNN_Q), 0, stream, P, K, w.sorted, w.idx_s, w.boxes, nb,
                     out_idx, out_dist);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int knn_dist2(int P, const float *points, float *mean_dists, char *workspace, hipStream_t stream) {
  if (P <= 0) return GSR_OK;
  KnnWorkspace w = knn_carve(workspace, (size_t)P);
  const int nred = knn_red_blocks(P);
  hipLaunchKernelGGL(aabb_partial_kernel, dim3(nred), dim3(256), 0, stream, P, points, w.partial);
  hipLaunchKernelGGL(aabb_final_kernel, dim3(1), dim3(64), 0, stream, nred, w.partial, w.minmax);
  hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, points, w.minmax, w.codes, w.idx);
  GSR_LAUNCH_CHECK(stream, 0);
  // 32-bit keys like the reference's SortPairs default (SK/simple_knn.cu:213): 4 passes, result in y = (codes_s, idx_s)
  int rc = radix_sort_u32((size_t)P, w.codes, w.idx, w.tk, w.tv, w.codes_s, w.idx_s, 32, w.hist, stream, 0);
  if (rc != GSR_OK) return rc;
  const int nb = (P + KNN_BOX - 1) / KNN_BOX;
  hipLaunchKernelGGL(gather_sorted_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, points, w.idx_s, w.sorted);
  hipLaunchKernelGGL(box_minmax_kernel, dim3(nb), dim3(256), 0, stream, P, w.sorted, w.boxes);
  hipLaunchKernelGGL(box_mean_dist_kernel, dim3((P + KNN_Q - 1) / KNN_Q), dim3(KNN_Q), 0, stream, P, w.sorted, w.idx_s,
                     w.boxes, nb, mean_dists);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

}  // namespace gsr

extern "C" {
size_t gsr_dist2_workspace_bytes(int P) { return gsr::knn_workspace_bytes(P > 0 ? (size_t)P : 1); }

int gsr_dist2(int P, const float *points, float *mean_dists, char *workspace, size_t workspace_bytes, gsr_stream_t stream) {
  if (P < 0 || (P > 0 && (!points || !mean_dists))) {
    gsr::set_error("gsr_dist2: bad arguments");
    return GSR_EINVAL;
  }
  if (P > 0 && (!workspace || workspace_bytes < gsr_dist2_workspace_bytes(P))) {
    gsr::set_error("gsr_dist2: workspace too small (%zu < %zu)", workspace_bytes, gsr_dist2_workspace_bytes(P));
    return GSR_ENOMEM;
  }
  return gsr::knn_dist2(P, points, mean_dists, workspace, reinterpret_cast<hipStream_t>(stream));
}

int gsr_knn_self(int P, const float *points, int k, int *idx, float *dist, char *workspace, size_t workspace_bytes,
                 gsr_stream_t stream) {
  if (P < 0 || k < 1 || k > 3 || (P > 0 && (!points || !idx || !dist))) {
    gsr::set_error("gsr_knn_self: bad arguments (k must be 1..3)");
    return GSR_EINVAL;
  }
  if (P > 0 && (!workspace || workspace_bytes < gsr_dist2_workspace_bytes(P))) {
    gsr::set_error("gsr_knn_self: workspace too small (%zu < %zu)", workspace_bytes, gsr_dist2_workspace_bytes(P));
    return GSR_ENOMEM;
  }
  return gsr::knn_self(P, k, points, idx, dist, workspace, reinterpret_cast<hipStream_t>(stream));
}
}
