// expand.h -- load-balanced expansion of (Gaussian, tile) instances, shared by the key-duplication kernel
// (geometry.hip) and the tile-bucket binning kernels (binning_bucket.hip).
//
// The workgroup that preprocessed Gaussians [b*256, b*256+256) visits their instances with one instance per lane
// per step (the reference loops over the tile rect inside one thread, CR/rasterizer_impl.cu:98-109, which leaves
// lanes idle and scatters the stores).  Emission order is the reference's: Gaussian index, then tile row, then
// tile column; instance k of the block has global index block_prefix + k.
#pragma once
#include "gsr_common.h"

namespace gsr {

// f(global instance index, gaussian id, tile id, depth bits)
template <typename F>
__device__ __forceinline__ void expand_block_instances(const GeomState &g, const int *radii, int P, int gx, int gy,
                                                       bool write_offsets, F f) {
  __shared__ uint32_t s_incl[PRE_BLOCK];
  __shared__ uint32_t s_depth[PRE_BLOCK];
  __shared__ uint32_t s_rect[PRE_BLOCK];  // x0 | y0 << 10 | width << 20
  const int first = blockIdx.x * PRE_BLOCK;
  const int i = first + threadIdx.x;
  const uint32_t bprefix = g.block_prefix[blockIdx.x];
  uint32_t incl = 0xFFFFFFFFu, rect = 0, dbits = 0;
  if (i < P) {
    incl = g.block_incl[i];
    if (write_offsets) g.point_offsets[i] = bprefix + incl;
    const int rad = radii[i];
    if (rad > 0) {
      const float4 r0 = reinterpret_cast<const float4 *>(g.recs + i)[0];
      const float4 r1 = reinterpret_cast<const float4 *>(g.recs + i)[1];
      int x0, y0, x1, y1;
      tile_rect(r0.x, r0.y, rad, gx, gy, x0, y0, x1, y1);
      rect = (uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)(x1 - x0) << 20);
      dbits = __float_as_uint(r1.z);
    }
  }
  s_incl[threadIdx.x] = incl;
  s_depth[threadIdx.x] = dbits;
  s_rect[threadIdx.x] = rect;
  __syncthreads();
  const int nvalid = min(PRE_BLOCK, P - first);
  const uint32_t total = s_incl[nvalid - 1];
  for (uint32_t k = threadIdx.x; k < total; k += PRE_BLOCK) {
    // first j with incl[j] > k  (zero-tile Gaussians have incl[j] == incl[j-1] and are never selected)
    int lo = 0, hi = nvalid - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_incl[mid] > k)
        hi = mid;
      else
        lo = mid + 1;
    }
    const uint32_t start = lo == 0 ? 0u : s_incl[lo - 1];
    const uint32_t local = k - start;
    const uint32_t rc = s_rect[lo];
    const uint32_t w = rc >> 20, x0 = rc & 1023u, y0 = (rc >> 10) & 1023u;
    const uint32_t ty = y0 + local / w, tx = x0 + local % w;
    f(bprefix + k, (uint32_t)(first + lo), ty * (uint32_t)gx + tx, s_depth[lo]);
  }
}


// Variant for kernels whose per-instance work starts with a long-latency returning operation (an atomic): UNROLL
// instances per lane are started back to back (tok = begin(...)) before any result is consumed (finish(..., tok)), so
// UNROLL atomics per lane are in flight instead of one.
// TIGHT: an instance whose tile cannot receive alpha >= 1/255 from the Gaussian (exact ellipse-vs-tile test, the one the
// blend kernels use per quadrant) is not started at all; finish() gets CULLED_INSTANCE as its token.
constexpr uint32_t CULLED_INSTANCE = 0xFFFFFFFFu;
template <int UNROLL, bool TIGHT, typename FB, typename FE>
__device__ __forceinline__ void expand_block_instances_2phase(const GeomState &g, const int *radii, int P, int gx, int gy,
                                                              bool write_offsets, FB begin, FE finish) {
  __shared__ uint32_t s_incl[PRE_BLOCK];
  __shared__ uint32_t s_rect[PRE_BLOCK];  // x0 | y0 << 10 | width << 20
  __shared__ float4 s_geo[TIGHT ? PRE_BLOCK : 1];   // x, y, conic a, conic b
  __shared__ float2 s_geo2[TIGHT ? PRE_BLOCK : 1];  // conic c, opacity
  const int first = blockIdx.x * PRE_BLOCK;
  const int i = first + threadIdx.x;
  const uint32_t bprefix = g.block_prefix[blockIdx.x];
  uint32_t incl = 0xFFFFFFFFu, rect = 0;
  if (i < P) {
    incl = g.block_incl[i];
    if (write_offsets) g.point_offsets[i] = bprefix + incl;
    const int rad = radii[i];
    if (rad > 0) {
      const float4 r0 = reinterpret_cast<const float4 *>(g.recs + i)[0];
      int x0, y0, x1, y1;
      tile_rect(r0.x, r0.y, rad, gx, gy, x0, y0, x1, y1);
      rect = (uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)(x1 - x0) << 20);
      if (TIGHT) {
        const float4 r1 = reinterpret_cast<const float4 *>(g.recs + i)[1];
        s_geo[threadIdx.x] = r0;
        s_geo2[threadIdx.x] = make_float2(r1.x, r1.y);
      }
    }
  }
  s_incl[threadIdx.x] = incl;
  s_rect[threadIdx.x] = rect;
  __syncthreads();
  const int nvalid = min(PRE_BLOCK, P - first);
  const uint32_t total = s_incl[nvalid - 1];
  for (uint32_t k0 = threadIdx.x; k0 < total; k0 += PRE_BLOCK * UNROLL) {
    uint32_t tok[UNROLL], inst[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t k = k0 + (uint32_t)u * PRE_BLOCK;
      ok[u] = k < total;
      inst[u] = bprefix + k;
      tok[u] = 0;
      if (ok[u]) {
        int lo = 0, hi = nvalid - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s_incl[mid] > k)
            hi = mid;
          else
            lo = mid + 1;
        }
        const uint32_t start = lo == 0 ? 0u : s_incl[lo - 1];
        const uint32_t local = k - start;
        const uint32_t rc = s_rect[lo];
        const uint32_t w = rc >> 20, x0 = rc & 1023u, y0 = (rc >> 10) & 1023u;
        const uint32_t ty = y0 + local / w, tx = x0 + local % w;
        bool keep = true;
        if (TIGHT) {
          const float4 ge = s_geo[lo];
          const float2 g2 = s_geo2[lo];
          const float px0 = (float)(tx * TILE), py0 = (float)(ty * TILE);
          keep = ellipse_hits_rect(ge.x, ge.y, ge.z, ge.w, g2.x, g2.y, px0, px0 + (float)(TILE - 1), py0, py0 + (float)(TILE - 1));
        }
        tok[u] = keep ? begin(ty * (uint32_t)gx + tx) : CULLED_INSTANCE;
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
      if (ok[u]) finish(inst[u], tok[u]);
  }
}

}  // namespace gsr
