// knn.hip -- exact k-nearest-neighbour kernels:
//   distCUDA2  mean squared distance to the 3 nearest OTHER points (replaces SimpleKNN::knn, SK/simple_knn.cu:185-221)
//   knn_self   the k <= 3 nearest of every point among the same points, with indices (KNN_CUDA's knn(xyz, xyz),
//              scene/gaussian_model.py:176,573,621,671)
// Built with -ffp-contract=off so squared distances carry the same roundings as the oracle; the search is exact, so results
// do not depend on the traversal order (ties between equal distances resolve to the lowest index).
//
// The reference sorts 30-bit Morton codes and scans boxes of 1024 Morton-consecutive points with AABB pruning
// (SK/simple_knn.cu:78-183): every query visits tens of boxes x 1024 candidates.  Here the points are binned into an
// ISOTROPIC uniform grid over their bounding cube (2^L cells per axis, about two points per cell for volumetric data:
// L = ceil(log2(cbrt(P/2)))) with one short radix sort of the linear cell ids; a boundary pass turns the sorted ids
// into a (start, end) table without any scan; a query then walks rings of cells of growing Chebyshev radius around
// its own cell and stops as soon as its k-th best distance is below (r-1) h, the distance every unvisited point exceeds.
// 200k surface points: ~100 candidates per query instead of ~20k.  No host round trips, one caller-provided workspace.
#include <float.h>

#include "gsr_common.h"

namespace gsr {

struct KnnHeader {
  float mn[3], h, inv_h;
};
struct KnnWorkspace {
  float *partial;    // [1024][6]
  KnnHeader *hdr;
  uint32_t *keys, *idx, *keys_s, *idx_s, *tk, *tv, *hist;
  float4 *sorted;    // [P] points in cell order: x, y, z, original index bits
  uint2 *table;      // [res^3] (start, end) of each cell in `sorted`; (0, 0) for empty cells
};
static inline int knn_level(size_t P) {
  int L = 2;
  while (L < 8 && (double)(1u << (3 * L)) * 2.0 < (double)P) L++;
  return L;
}
static inline int knn_red_blocks(int P) { return min(1024, (P + 1023) / 1024); }
static KnnWorkspace knn_carve(char *p_, size_t P, size_t *end = nullptr) {
  KnnWorkspace w;
  uintptr_t p = carve_begin(p_);
  size_t n = P ? P : 1;
  carve(p, w.partial, (size_t)1024 * 6);
  carve(p, w.hdr, 1);
  carve(p, w.keys, n);
  carve(p, w.idx, n);
  carve(p, w.keys_s, n);
  carve(p, w.idx_s, n);
  carve(p, w.tk, n);
  carve(p, w.tv, n);
  carve(p, w.hist, sort_hist_words(n));
  carve(p, w.sorted, n);
  carve(p, w.table, (size_t)1 << (3 * knn_level(n)));
  if (end) *end = p;
  return w;
}
size_t knn_workspace_bytes(size_t P) {
  size_t end = 0;
  knn_carve(nullptr, P, &end);
  return end + 512;
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

// bounding box of the points, two levels
__global__ __launch_bounds__(256) void aabb_partial_kernel(int P, const float *pts, float *partial) {
  __shared__ float s[4][6];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
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
__global__ void aabb_final_kernel(int nred, const float *partial, KnnHeader *hdr, int res) {
  if (threadIdx.x != 0) return;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int b = 0; b < nred; b++)
    for (int k = 0; k < 3; k++) {
      mn[k] = fminf(mn[k], partial[b * 6 + k]);
      mx[k] = fmaxf(mx[k], partial[b * 6 + 3 + k]);
    }
  float ext = 0.f;
  for (int k = 0; k < 3; k++) {
    hdr->mn[k] = mn[k];
    ext = fmaxf(ext, mx[k] - mn[k]);
  }
  const float h = fmaxf(ext / (float)res, 1e-30f);
  hdr->h = h;
  hdr->inv_h = 1.0f / h;
}

__device__ __forceinline__ int cell_coord(float x, float lo, float inv_h, int res) {
  const int c = (int)floorf((x - lo) * inv_h);
  return min(max(c, 0), res - 1);
}

__global__ void cell_key_kernel(int P, const float *pts, const KnnHeader *hdr, int res, uint32_t *keys, uint32_t *idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int cx = cell_coord(pts[3 * (size_t)i], hdr->mn[0], hdr->inv_h, res);
  const int cy = cell_coord(pts[3 * (size_t)i + 1], hdr->mn[1], hdr->inv_h, res);
  const int cz = cell_coord(pts[3 * (size_t)i + 2], hdr->mn[2], hdr->inv_h, res);
  keys[i] = (uint32_t)((cz * res + cy) * res + cx);
  idx[i] = (uint32_t)i;
}

// points into cell order + the (start, end) table from the boundaries of the sorted ids (the table was zeroed)
__global__ void gather_table_kernel(int P, const float *pts, const uint32_t *keys_s, const uint32_t *idx_s, float4 *sorted,
                                    uint2 *table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const uint32_t g = idx_s[i], c = keys_s[i];
  sorted[i] = make_float4(pts[3 * (size_t)g], pts[3 * (size_t)g + 1], pts[3 * (size_t)g + 2], __uint_as_float(g));
  if (i == 0 || keys_s[i - 1] != c) table[c].x = (uint32_t)i;
  if (i == P - 1 || keys_s[i + 1] != c) table[c].y = (uint32_t)(i + 1);
}

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = bx - ax, dy = by - ay, dz = bz - az;
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ void kbest3(float d, float &b0, float &b1, float &b2) {  // SK/simple_knn.cu:131-145
  if (b0 > d) { const float t = b0; b0 = d; d = t; }
  if (b1 > d) { const float t = b1; b1 = d; d = t; }
  if (b2 > d) { b2 = d; }
}
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

// One thread per point, in cell order (neighbouring threads walk the same cells).  WITH_IDX = false: distCUDA2 (the point
// itself is skipped BY POSITION, coincident other points count, like SK/simple_knn.cu:170); true: knn_self (self included).
template <bool WITH_IDX>
__global__ __launch_bounds__(256) void grid_knn_kernel(int P, int K, const KnnHeader *hdr, int res, const float4 *sorted,
                                                       const uint2 *table, float *mean_dists, int *out_idx, float *out_dist) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float4 p = sorted[i];
  const float h = hdr->h;
  const int cx = cell_coord(p.x, hdr->mn[0], hdr->inv_h, res);
  const int cy = cell_coord(p.y, hdr->mn[1], hdr->inv_h, res);
  const int cz = cell_coord(p.z, hdr->mn[2], hdr->inv_h, res);
  float bd[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  uint32_t bi[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  const int rmax = max(max(max(cx, res - 1 - cx), max(cy, res - 1 - cy)), max(cz, res - 1 - cz));
  for (int r = 0; r <= rmax; r++) {
    if (r >= 2) {
      // every point not visited yet sits at least r cells away along some axis: farther than (r - 1) h
      const float lb = (float)(r - 1) * h * 0.9999f;
      if (bd[2] < lb * lb) break;
    }
    const int z0 = max(cz - r, 0), z1 = min(cz + r, res - 1), y0 = max(cy - r, 0), y1 = min(cy + r, res - 1);
    for (int z = z0; z <= z1; z++)
      for (int y = y0; y <= y1; y++) {
        const bool shell = (z - cz == r) || (cz - z == r) || (y - cy == r) || (cy - y == r);
        const int xs = shell ? 1 : max(2 * r, 1);  // shell rows: every x; inner rows: only x = cx - r and cx + r
        for (int x = cx - r; x <= cx + r; x += xs) {
          if (x < 0 || x >= res) continue;
          const uint2 se = table[((size_t)z * res + y) * res + x];
          for (uint32_t j = se.x; j < se.y; j++) {
            const float4 q = sorted[j];
            const float d = sqdist3(p.x, p.y, p.z, q.x, q.y, q.z);
            if (WITH_IDX) {
              kbest3_idx(d, __float_as_uint(q.w), bd, bi);
            } else if ((int)j != i) {
              kbest3(d, bd[0], bd[1], bd[2]);
            }
          }
        }
      }
  }
  const uint32_t self = __float_as_uint(p.w);
  if (WITH_IDX) {
    for (int k = 0; k < K; k++) {
      out_idx[(size_t)self * K + k] = (int)bi[k];
      out_dist[(size_t)self * K + k] = sqrtf(bd[k]);
    }
  } else {
    mean_dists[self] = (bd[0] + bd[1] + bd[2]) / 3.0f;
  }
}

static int knn_build(int P, const float *points, const KnnWorkspace &w, int L, hipStream_t stream) {
  const int res = 1 << L;
  const int nred = knn_red_blocks(P);
  hipLaunchKernelGGL(aabb_partial_kernel, dim3(nred), dim3(256), 0, stream, P, points, w.partial);
  hipLaunchKernelGGL(aabb_final_kernel, dim3(1), dim3(64), 0, stream, nred, w.partial, w.hdr, res);
  hipLaunchKernelGGL(cell_key_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, points, w.hdr, res, w.keys, w.idx);
  GSR_LAUNCH_CHECK(stream, 0);
  const int end_bit = 3 * L;
  // pass 0 reads (keys, idx) and writes (tk, tv), then ping-pongs with (keys_s, idx_s)
  int rc = radix_sort_u32((size_t)P, w.keys, w.idx, w.tk, w.tv, w.keys_s, w.idx_s, end_bit, w.hist, stream, 0);
  if (rc != GSR_OK) return rc;
  const bool in_x = radix_passes(end_bit) % 2 == 1;
  const uint32_t *ks = in_x ? w.tk : w.keys_s, *is = in_x ? w.tv : w.idx_s;
  GSR_HIP(zero_async(w.table, sizeof(uint2) * ((size_t)1 << (3 * L)), stream));
  hipLaunchKernelGGL(gather_table_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, points, ks, is, w.sorted, w.table);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int knn_dist2(int P, const float *points, float *mean_dists, char *workspace, hipStream_t stream) {
  if (P <= 0) return GSR_OK;
  KnnWorkspace w = knn_carve(workspace, (size_t)P);
  const int L = knn_level((size_t)P);
  int rc = knn_build(P, points, w, L, stream);
  if (rc != GSR_OK) return rc;
  hipLaunchKernelGGL(grid_knn_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, stream, P, 3, w.hdr, 1 << L, w.sorted, w.table,
                     mean_dists, (int *)nullptr, (float *)nullptr);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int knn_self(int P, int K, const float *points, int *out_idx, float *out_dist, char *workspace, hipStream_t stream) {
  if (P <= 0) return GSR_OK;
  KnnWorkspace w = knn_carve(workspace, (size_t)P);
  const int L = knn_level((size_t)P);
  int rc = knn_build(P, points, w, L, stream);
  if (rc != GSR_OK) return rc;
  hipLaunchKernelGGL(grid_knn_kernel<true>, dim3((P + 255) / 256), dim3(256), 0, stream, P, K, w.hdr, 1 << L, w.sorted, w.table,
                     (float *)nullptr, out_idx, out_dist);
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
