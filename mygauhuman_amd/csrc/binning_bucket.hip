// binning_bucket.hip -- tile-bucket binning back-end (GSR_BINNING_TILE_BUCKET).
//
// The reference sorts all R (Gaussian, tile) instances with a device-wide radix sort on a 64-bit key
// (tile << 32 | depth bits): 6 passes over 12-byte pairs for a 45-bit key (CR/rasterizer_impl.cu:291-320).  The sorted
// order only ever matters INSIDE a tile, and the per-tile lists are short (hundreds of entries), which is LDS-sized
// work on MI355X.  So this back-end never sorts globally:
//   1. count   : every instance increments its tile's counter with ONE returning atomic and keeps the returned
//                arrival rank                                                   (load-balanced over instances)
//   2. scan    : exclusive scan of the tile counters -> ranges[tile] directly   (replaces identifyTileRanges)
//   3. scatter : instance -> bucket[start[tile] + rank] = 64-bit sort key (depth bits << 32 | Gaussian id); no
//                atomics, arrival order is arbitrary
//   4. sort    : one WAVE per tile sorts lists of up to 512 keys (64-key runs in registers + rank merge through 4 KB of
//                LDS, no barriers); longer lists go onto a work list and get one workgroup each: every wave sorts a
//                quarter in registers, the last merge levels run through LDS.  Both write the point list
//                (+ the sorted reference keys).  Keys are unique because they contain the Gaussian id, and ordering by
//                (depth bits, Gaussian id) is exactly what a STABLE sort by (tile, depth bits) gives within a tile,
//                since the reference emits instances in Gaussian-index order.  Tiles whose list does not fit LDS
//                (> 4096 entries) are sorted by the same workgroup as LDS-sized runs merged in global memory.
// The result (point_list, ranges, sorted keys) is bit-identical to the global radix back-end; traffic drops from
// ~150 B to ~30 B per instance and the kernel count from 20 to 4.
#include <atomic>
#include <type_traits>
#include "expand.h"
#include "gsr_common.h"

namespace gsr {

constexpr int SORT_WAVE_MAX = 512;   // longest list one wave sorts by itself (8 runs of 64 + rank merge); longer ones: bucket_sort_kernel
constexpr int SORT_BIG = 2048;  // keys the workgroup sort of lists > 512 handles in LDS (two buffers of 16 KB); longer lists: chunks + global merge

// One returning atomic per instance: the value it returns is the instance's arrival rank inside its tile, kept in
// rank[instance] so that the scatter pass needs no second round of atomics.  Counter t lives at counts[t * CSTRIDE]:
// device-scope atomics execute at the memory side and serialise per 64-byte line, so neighbouring tiles should not
// share a line.
// (Options::bucket_cstride, default 4.)
// Tight tile culling (Options::tile_cull, default on): an instance of the reference's tile rectangle
// (CR/auxiliary.h:46-56, a square around 3 sigma) whose tile the ellipse {alpha >= 1/255} does not reach is dropped here --
// every pixel test of that instance would say "skip" (CR/forward.cu:344-349), so images and gradients do not change; lists,
// n_contrib positions and the per-tile ranges shrink (C3: 1.33 M -> 0.97 M instances).  rank[] keeps a sentinel for them.
template <bool TIGHT>
__global__ __launch_bounds__(PRE_BLOCK) void bucket_count_kernel(const GeomState g, const int *radii, int P, int gx, int gy,
                                                                uint32_t *counts, uint32_t *rank, uint32_t capacity, int CSTRIDE) {
  expand_block_instances_2phase<4, TIGHT>(
      g, radii, P, gx, gy, true, [&](uint32_t tile) { return atomicAdd(&counts[(size_t)tile * CSTRIDE], 1u); },
      [&](uint32_t inst, uint32_t r) {
        if (inst < capacity) rank[inst] = r;
      });
}

// exclusive scan of counts[tiles] -> ranges[t] = (start, end); cursor[t] = start
// If the instance total exceeds `capacity` (only possible when the host sized the buffer without knowing R) every
// range is emptied -- nothing is scattered, sorted or blended -- and status[1] is raised for the host to see.
// body of the tile scan for ONE workgroup of 1024 threads
__device__ __forceinline__ void tile_scan_block(const uint32_t *counts, uint32_t *cursor, uint2 *ranges, int n, const uint32_t *total,
                                                uint32_t capacity, uint32_t *status, int CSTRIDE, int check_prefilter,
                                                uint32_t *wtot, uint32_t *carry_s, uint32_t *order) {
  const uint32_t R = *total;
  if (threadIdx.x == 0 && order) order[0] = 0u;  // natural tile order for the blend kernels (only the histogram path reorders)
  if (threadIdx.x == 0 && status) {
    status[0] = R;
    status[1] = (R > capacity ? 1u : 0u) | ((check_prefilter && total[1]) ? 2u : 0u);
  }
  if (R > capacity) {
    for (int i = threadIdx.x; i < n; i += 1024) {
      cursor[i] = 0;
      ranges[i] = make_uint2(0u, 0u);
    }
    return;
  }
  if (threadIdx.x == 0) *carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < n ? counts[(size_t)i * CSTRIDE] : 0;
    const uint32_t incl_w = wave_incl_scan(v);
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    if (lane == WAVE - 1) wtot[wave] = incl_w;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wtot[w];
    const uint32_t carry = *carry_s;
    const uint32_t start = carry + woff + incl_w - v;
    if (i < n) {
      cursor[i] = start;
      // empty tiles keep (0, 0) like the reference's zero-filled ranges (CR/rasterizer_impl.cu:312)
      ranges[i] = v ? make_uint2(start, start + v) : make_uint2(0u, 0u);
    }
    __syncthreads();
    if (threadIdx.x == 1023) *carry_s = carry + woff + incl_w;
    __syncthreads();
  }
}
__global__ __launch_bounds__(1024) void bucket_scan_kernel(const uint32_t *counts, uint32_t *cursor, uint2 *ranges, int n,
                                                          const uint32_t *total, uint32_t capacity, uint32_t *status, int CSTRIDE,
                                                          int check_prefilter, uint32_t *order) {
  __shared__ uint32_t wtot[1024 / WAVE];
  __shared__ uint32_t carry_s;
  tile_scan_block(counts, cursor, ranges, n, total, capacity, status, CSTRIDE, check_prefilter, wtot, &carry_s, order);
}

__global__ __launch_bounds__(PRE_BLOCK) void bucket_scatter_kernel(const GeomState g, const int *radii, int P, int gx, int gy,
                                                                  const uint32_t *start, const uint32_t *rank,
                                                                  uint64_t *bucket, uint32_t capacity) {
  if (blockIdx.x == 0 && threadIdx.x == 0) g.total[2] = 0;  // work-list counter of the sort kernels that follow
  if (*g.total > capacity) return;  // overflow: see bucket_scan_kernel
  expand_block_instances(g, radii, P, gx, gy, false, [&](uint32_t inst, uint32_t gid, uint32_t tile, uint32_t dbits) {
    const uint32_t r = rank[inst];
    if (r != CULLED_INSTANCE) bucket[start[tile] + r] = ((uint64_t)dbits << 32) | (uint64_t)gid;
  });
}

// ---- atomics-free counting (Options::bucket_hist, default on) ---------------------------------------------------------------
// The count pass above is bound by the memory side: one scattered 4-byte returning atomic per instance is one 64-byte request
// each (~25 k requests / us chip-wide: 0.97 M instances = 34 us at C3).  Here no instance touches a global atomic:
//   hist    : a workgroup owns `hbv` <= HB consecutive Gaussians (an even split of P over a multiple of the CU count, see
//             hist_split), keeps one counter per TILE in LDS, and every instance takes
//             its rank among the workgroup's instances of that tile with an LDS returning add; per instance it leaves
//             (tile << 16 | rank) and the Gaussian id in instance order (coalesced), the counters go out as one dense row
//             table[workgroup][tile]
//   prefix  : 64 tiles x 16 waves per workgroup: every column becomes an exclusive prefix over the workgroups, tile totals out
//   (tile scan as before: ranges / start)
//   scatter : flat over the instances of a workgroup, no expansion: slot = start[tile] + table[workgroup][tile] + rank
// The arrival order inside a tile is as arbitrary as with atomics; the per-tile sort makes the result deterministic.
constexpr int HB = 1024;               // threads per histogram workgroup = the most Gaussians it can own
constexpr int HNB = HB / PRE_BLOCK + 1;  // preprocess blocks a workgroup's Gaussian range can touch
constexpr int HB_P2 = 1024;
constexpr int HU = 2;                  // owner searches in flight per lane of the histogram kernel
constexpr int HIST_MAX_TILES = 8192;   // LDS counters: 32 KB (larger tile grids take the atomic path)
__device__ __forceinline__ int pre_blocks_dev(int P) { return (P + PRE_BLOCK - 1) / PRE_BLOCK; }

// SCAN: the second level of the tiles_touched scan (scan_block_sums_kernel: block prefixes and R) is done here as well -- every
// workgroup adds the block sums in front of it (a few hundred words), writes the prefixes of its own blocks, and the last one
// writes R -- which saves a single-workgroup launch on the asynchronous path (the blocking path needs R on the host earlier).
template <bool TIGHT, bool SCAN>
__global__ __launch_bounds__(HB) void bucket_hist_kernel(const GeomState g, const int *radii, int P, int gx, int gy, int tiles,
                                                        uint32_t *table, uint32_t *rank, uint32_t *gids, uint32_t capacity, int hbv,
                                                        uint32_t *wg_start) {
  __shared__ uint32_t s_cnt[HIST_MAX_TILES];
  __shared__ uint32_t s_incl[HB_P2];  // padded with 0xFFFFFFFF for the branch-free search
  __shared__ uint32_t s_rect[HB];  // x0 | y0 << 10 | width << 20
  __shared__ float4 s_geo[TIGHT ? HB : 1];   // x, y, conic a, conic b
  __shared__ float2 s_geo2[TIGHT ? HB : 1];  // conic c, opacity
  __shared__ uint32_t s_wsum[HB / WAVE];
  __shared__ uint32_t s_boff[HNB + 2];  // [0] = instances in front of block b0, [1 + k] = offset of block b0 + k, [HNB + 1] = see below
  const int first = blockIdx.x * hbv;   // this workgroup owns the Gaussians [first, first + hbv)
  const int i = first + (int)threadIdx.x;
  const bool has = (int)threadIdx.x < hbv && i < P;
  const int b0 = first / PRE_BLOCK, n_pre = pre_blocks_dev(P);
  for (int t = threadIdx.x; t < tiles; t += HB) s_cnt[t] = 0;
  // this thread's Gaussian: issue the loads before the scan of the block sums so both latencies overlap
  uint32_t incl_local = 0;
  int rad = 0;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
  if (has) {
    incl_local = g.block_incl[i];
    rad = radii[i];
    r0 = reinterpret_cast<const float4 *>(g.recs + i)[0];
    if (TIGHT) r1 = reinterpret_cast<const float4 *>(g.recs + i)[1];
  }
  // instances of block b0 that belong to the workgroup in front (the range need not start on a block boundary)
  if (threadIdx.x == HB - 1) s_boff[HNB + 1] = (first % PRE_BLOCK) ? g.block_incl[first - 1] : 0u;
  if (SCAN) {
    // own = the sums of the blocks this range touches (lanes 0..HNB-1 of wave 0), part = every block in front of b0
    const uint32_t own = ((int)threadIdx.x < HNB && b0 + (int)threadIdx.x < n_pre) ? g.block_sums[b0 + threadIdx.x] : 0u;
    uint32_t part = 0;
    for (int b = threadIdx.x; b < b0; b += HB) part += g.block_sums[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, WAVE);
    if (threadIdx.x % WAVE == 0) s_wsum[threadIdx.x / WAVE] = part;
    __syncthreads();
    if (threadIdx.x < WAVE) {
      uint32_t run = (int)threadIdx.x < HB / WAVE ? s_wsum[threadIdx.x] : 0u;
#pragma unroll
      for (int d = WAVE / 2; d >= 1; d >>= 1) run += __shfl_xor(run, d, WAVE);  // (HB / WAVE <= 64 partial sums, zeros behind)
      uint32_t off = own;  // inclusive scan over the HNB lanes
#pragma unroll
      for (int d = 1; d < HNB; d <<= 1) {
        const uint32_t up = __shfl_up(off, d, WAVE);
        if ((int)threadIdx.x >= d) off += up;
      }
      if ((int)threadIdx.x < HNB) {
        s_boff[1 + threadIdx.x] = off - own;
        // every block prefix is written by the workgroup that owns the block's first Gaussian
        const int blk = b0 + (int)threadIdx.x, blk_first = blk * PRE_BLOCK;
        if (blk < n_pre && blk_first >= first && blk_first < first + hbv) g.block_prefix[blk] = run + off - own;
      }
      if (threadIdx.x == 0) s_boff[0] = run;
      if ((int)threadIdx.x == HNB - 1 && blockIdx.x == gridDim.x - 1) g.total[0] = run + off;  // the last blocks: this is R
    }
    __syncthreads();
  } else {
    if ((int)threadIdx.x <= HNB) {
      const uint32_t base = g.block_prefix[b0];
      if (threadIdx.x == 0) s_boff[0] = base;
      else s_boff[threadIdx.x] = (b0 + (int)threadIdx.x - 1 < n_pre ? g.block_prefix[b0 + threadIdx.x - 1] : base) - base;
    }
    __syncthreads();
  }
  const uint32_t sb_prefix = s_boff[0] + s_boff[HNB + 1];  // global index of the workgroup's first instance
  if (threadIdx.x == 0) wg_start[blockIdx.x] = sb_prefix;
  uint32_t incl = 0xFFFFFFFFu, rect = 0;
  if (has) {
    // inclusive scan of tiles_touched: block-local scan + the block's offset; relative to the workgroup for the search below
    const uint32_t global_incl = s_boff[0] + s_boff[1 + (i / PRE_BLOCK - b0)] + incl_local;
    incl = global_incl - sb_prefix;
    g.point_offsets[i] = global_incl;
    if (rad > 0) {
      int x0, y0, x1, y1;
      tile_rect(r0.x, r0.y, rad, gx, gy, x0, y0, x1, y1);
      rect = (uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)(x1 - x0) << 20);
      if (TIGHT) {
        s_geo[threadIdx.x] = r0;
        s_geo2[threadIdx.x] = make_float2(r1.x, r1.y);
      }
    }
  }
  s_incl[threadIdx.x] = incl;
  s_rect[threadIdx.x] = rect;
  __syncthreads();
  const int nvalid = min(hbv, P - first);
  const uint32_t total = s_incl[nvalid - 1];
  // owner search: branch-free descent over the inclusive scan (entries past nvalid hold 0xFFFFFFFF), HU instances per lane in
  // flight so the dependent LDS reads of one search hide behind the other's
  for (uint32_t k0 = threadIdx.x; k0 < total; k0 += HB * HU) {
    uint32_t kk[HU];
    int own[HU];
#pragma unroll
    for (int u = 0; u < HU; u++) {
      kk[u] = k0 + (uint32_t)u * HB;
      own[u] = 0;
    }
#pragma unroll
    for (int step = HB_P2 / 2; step >= 1; step >>= 1) {
#pragma unroll
      for (int u = 0; u < HU; u++) {
        const int idx = own[u] + step;
        if (s_incl[idx - 1] <= kk[u]) own[u] = idx;  // own = number of Gaussians that end at or before instance kk
      }
    }
#pragma unroll
    for (int u = 0; u < HU; u++) {
      const uint32_t k = kk[u];
      if (k >= total) continue;
      const int lo = own[u];
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
      const uint32_t tile = ty * (uint32_t)gx + tx;
      const uint32_t r = keep ? ((tile << 16) | atomicAdd(&s_cnt[tile], 1u)) : CULLED_INSTANCE;  // LDS returning add, rank < HB
      const uint32_t inst = sb_prefix + k;
      if (inst < capacity) {
        rank[inst] = r;
        gids[inst] = (uint32_t)(first + lo);
      }
    }
  }
  __syncthreads();
  uint32_t *row = table + (size_t)blockIdx.x * tiles;
  for (int t = threadIdx.x; t < tiles; t += HB) row[t] = s_cnt[t];
}

// column t of table[n_sb][tiles] -> exclusive prefix over the workgroups (in place), totals[t] = the tile's instance count.
// A workgroup takes 64 tiles (lanes) x PW waves; wave w owns the rows [w * chunk, (w + 1) * chunk): sum them (independent
// loads), exchange the wave sums through LDS, then re-walk the rows (L2 hits) and write the running prefixes.
constexpr int PW = 16;
// (Letting the last workgroup to finish scan the tile totals as well -- ticket counter + __threadfence -- was tried: 31 us instead
// of 5 + 6 for the two launches; the device-scope fences write back / invalidate the L2s of all eight XCDs.)
__global__ __launch_bounds__(PW *WAVE) void bucket_hist_prefix_kernel(uint32_t *table, int n_sb, int tiles, uint32_t *totals) {
  __shared__ uint32_t s_sum[PW][WAVE];
  const int lane = threadIdx.x % WAVE, w = threadIdx.x / WAVE;
  const int t = blockIdx.x * WAVE + lane;
  const int chunk = (n_sb + PW - 1) / PW;
  const int r0 = min(n_sb, w * chunk), r1 = min(n_sb, r0 + chunk);
  uint32_t sum = 0;
  if (t < tiles) {
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
      const uint32_t *p = table + (size_t)r * tiles + t;
      sum += (p[0] + p[tiles]) + (p[2 * (size_t)tiles] + p[3 * (size_t)tiles]);
    }
    for (; r < r1; r++) sum += table[(size_t)r * tiles + t];
  }
  s_sum[w][lane] = sum;
  __syncthreads();
  uint32_t run = 0, all = 0;
  for (int k = 0; k < PW; k++) {
    const uint32_t v = s_sum[k][lane];
    run += k < w ? v : 0u;
    all += v;
  }
  if (t < tiles) {
    for (int r = r0; r < r1; r++) {
      uint32_t *p = table + (size_t)r * tiles + t;
      const uint32_t c = *p;
      *p = run;
      run += c;
    }
    if (w == 0) totals[t] = all;
  }
}

// one workgroup per histogram workgroup, flat over its instances [wg_start[w], wg_start[w + 1]): the slot bases of all tiles are
// staged in LDS (start + this workgroup's table row), the depth comes from the Gaussian's record
// The tile scan (exclusive prefix of the tile totals -> slot bases, ranges) is part of this kernel: every workgroup needs all the
// bases in LDS anyway, so each one scans the totals itself (<= 8 per thread + one wave scan, a fraction of a microsecond, all
// workgroups in parallel) and workgroup 0 also writes ranges[] and the status words -- the separate one-workgroup scan launch
// between the prefix and the scatter kernel (6 us of launch gap and latency) is gone.
__global__ __launch_bounds__(HB) void bucket_scatter_hist_kernel(const GeomState g, const uint32_t *wg_start, int tiles,
                                                                const uint32_t *totals, const uint32_t *table, const uint32_t *rank,
                                                                const uint32_t *gids, uint64_t *bucket, uint32_t capacity,
                                                                uint2 *ranges, uint32_t *status, int check_prefilter, uint32_t *order, int order_mode,
                                                                int grid_x, uint32_t *ckpt_base, int segments, uint32_t *big_list) {
  constexpr int PER_MAX = (HIST_MAX_TILES + HB - 1) / HB;
  __shared__ uint32_t s_base[HIST_MAX_TILES];
  __shared__ uint32_t s_wtot[HB / WAVE];
  const uint32_t R = *g.total;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (R > capacity) g.total[2] = 0;  // work-list counter of the long-list sort (otherwise written by the order builder below)
    if (status) {
      status[0] = R;
      status[1] = (R > capacity ? 1u : 0u) | ((check_prefilter && g.total[1]) ? 2u : 0u);
    }
  }
  if (R > capacity) {  // overflow: nothing is binned, every tile is empty (the caller reads the status words and regrows)
    if (blockIdx.x == 0) {
      for (int t = threadIdx.x; t < tiles; t += HB) ranges[t] = make_uint2(0u, 0u);
      if (threadIdx.x == 0) order[0] = 0u;
    }
    return;
  }
  // the LAST workgroup of the launch owns no instances: it only builds the visiting order of the tiles (below), concurrently with
  // the scatter of the others -- inside the last scatter workgroup that build sat on the critical path (+2 .. +5 us per frame)
  const bool builder = blockIdx.x == gridDim.x - 1;
  const uint32_t n_scatter = gridDim.x - 1;
  const uint32_t *row = table + (size_t)(builder ? 0u : blockIdx.x) * tiles;
  const int per = (tiles + HB - 1) / HB, t0 = (int)threadIdx.x * per;
  uint32_t cnt[PER_MAX], mine[PER_MAX], local = 0;
#pragma unroll
  for (int k = 0; k < PER_MAX; k++) {
    const int t = t0 + k;
    const bool in = k < per && t < tiles;
    cnt[k] = in ? totals[t] : 0u;
    mine[k] = (in && !builder) ? row[t] : 0u;
    local += cnt[k];
  }
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  if (!builder) {
    const uint32_t incl_w = wave_incl_scan(local);
    if (lane == WAVE - 1) s_wtot[wave] = incl_w;
    const uint32_t i0 = wg_start[blockIdx.x], i1 = blockIdx.x + 1 < n_scatter ? wg_start[blockIdx.x + 1] : R;
    __syncthreads();
    uint32_t start = incl_w - local;
    for (int w = 0; w < wave; w++) start += s_wtot[w];
#pragma unroll
    for (int k = 0; k < PER_MAX; k++) {
      const int t = t0 + k;
      if (k < per && t < tiles) {
        s_base[t] = start + mine[k];
        // empty tiles keep (0, 0) like the reference's zero-filled ranges (CR/rasterizer_impl.cu:312)
        if (blockIdx.x == 0) ranges[t] = cnt[k] ? make_uint2(start, start + cnt[k]) : make_uint2(0u, 0u);
        start += cnt[k];
      }
    }
    __syncthreads();
    for (uint32_t inst = i0 + threadIdx.x; inst < i1; inst += HB) {
      const uint32_t r = rank[inst];
      if (r == CULLED_INSTANCE) continue;
      const uint32_t gid = gids[inst];
      const uint32_t dbits = __float_as_uint(g.recs[gid].depth);
      bucket[s_base[r >> 16] + (r & 0xFFFFu)] = ((uint64_t)dbits << 32) | (uint64_t)gid;
    }
    return;
  }
  // ---- work list of the long-list sort (this workgroup sees every list length): written HERE, not by 4,096 wave-sort workgroups
  // with a global atomic each
  {
    __shared__ uint32_t s_nbig;
    if (threadIdx.x == 0) s_nbig = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER_MAX; k++) {
      const int t = t0 + k;
      if (k < per && t < tiles && cnt[k] > (uint32_t)SORT_WAVE_MAX) big_list[atomicAdd(&s_nbig, 1u)] = (uint32_t)t;
    }
    __syncthreads();
    if (threadIdx.x == 0) g.total[2] = s_nbig;
  }
  // ---- visiting order of the tiles for the blend kernels (built by the extra workgroup): the tiles with the longest lists go
  // FIRST, dealt round-robin to the eight XCDs (tile_order_mode / ordered_item4 in gsr_common.h).  A wave walks its list
  // serially, so a blend kernel cannot end before its longest list has been walked from wherever that wave STARTED, and the
  // natural order with a contiguous band of tile rows per XCD leaves whole XCDs short of work when the scene is not uniform.
  // Measured: close-up of a body (1,480 busy tiles, lists up to 3x the mean) blend forward 212 -> 153 us, backward 419 -> 284;
  // C3 (uniform cloud, lists 150..355) forward 114 -> 103, backward 230 -> 211: the balance is worth more than keeping
  // neighbouring tiles on one L2.  (Longest first INSIDE each XCD's band of tile rows, to keep that locality: 0.440 vs
  // 0.415 ms per C3 step; the long lists merely moved to the front of the contiguous mapping: all on XCD 0, 377 vs 212 us.)
  __shared__ uint32_t s_red[HB / WAVE][3];
  __shared__ uint32_t s_thr, s_busy, s_kept, s_extra, s_rec;
  uint32_t busy = 0, mx = 0;
#pragma unroll
  for (int k = 0; k < PER_MAX; k++) {
    busy += cnt[k] ? 1u : 0u;
    mx = max(mx, cnt[k]);
  }
  uint32_t kept = local;  // instances in the lists (after the exact tile cull: fewer than *g.total)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    busy += __shfl_xor(busy, d, WAVE);
    kept += __shfl_xor(kept, d, WAVE);
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, WAVE));
  }
  if (lane == 0) {
    s_red[wave][0] = busy;
    s_red[wave][1] = mx;
    s_red[wave][2] = kept;
  }
  __syncthreads();
  if (wave == 0) {  // (the per-wave partials meet in the first wave's lanes: no serial loop of LDS reads on this workgroup's only path)
    static_assert(HB / WAVE <= WAVE, "one lane per wave");
    uint32_t b = lane < HB / WAVE ? s_red[lane][0] : 0u, m = lane < HB / WAVE ? s_red[lane][1] : 0u, kp = lane < HB / WAVE ? s_red[lane][2] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      b += __shfl_xor(b, d, WAVE);
      kp += __shfl_xor(kp, d, WAVE);
      m = max(m, (uint32_t)__shfl_xor((int)m, d, WAVE));
    }
    if (lane == 0) {
      s_thr = order_mode == 0 ? 0u : m;  // Options::tile_order: 0 natural order; the longest list (0: nothing to render)
      order[0] = s_thr ? ((uint32_t)order_mode | (min(m, 0xFFFFFFu) << 8)) : 0u;
      s_busy = b;
      s_kept = kp;
      s_extra = 0u;
      s_rec = 0u;
    }
  }
  __syncthreads();
  const uint32_t mxn = s_thr;  // the longest list, 0: natural order (or nothing to render)
  if (mxn == 0u) return;
  constexpr int NCLS = 64;
  __shared__ uint32_t s_ccount[NCLS], s_cbase[NCLS];
  if ((int)threadIdx.x < NCLS) s_ccount[threadIdx.x] = 0u;
  if (order_mode == 1) {
    // counting sort of the TILES by list length, longest first: 64 length classes (class of n = 1 + 62 n / max, empty tiles
    // last), the order inside a class is whatever the LDS atomics give (every tile is rendered by itself: any order of equals
    // is as good)
    //
    // List segments (gsr_common.h): a list much longer than the frame's mean is cut into pieces of about the mean length, each with
    // a visiting slot of its own (sorted by the PIECE's length) and the tile's checkpoint records reserved here.  "Much longer" =
    // at least 1.5 x max(mean over the busy tiles, 256): a uniform cloud (C3: lists 150..355) cuts nothing.
    //
    // Tail cut (segments >> 8 = sixteenths of the busy tiles, Options::blend_tail_cut): the lists that are visited LAST are cut in two
    // as well.  A blend wave is a large unit of work -- at C3 the backward is 2.7 rounds of ~75 us waves -- and the kernel ends with a
    // tail in which the last-started waves run among idle SIMDs (tools/wave_trace.py: 15 % of the span).  Half-length pieces at the
    // end of the order halve that tail; the forward of a cut tile only pays the checkpoint stores.
    __syncthreads();
    const uint32_t target = max(s_busy ? s_kept / s_busy : 0u, 256u), extra_cap = seg_extra_max((uint32_t)tiles);
    const float cls_scale = (float)(NCLS - 2) / (float)mxn;
    const uint32_t outlier = (uint32_t)segments & 0xFFu, tail16 = ((uint32_t)segments >> 8) & 0xFFu;
    uint32_t mycls[PER_MAX], mycls0[PER_MAX], mysegs[PER_MAX];
    // (classes by a float multiply: the order of near-equal lengths is of no consequence, a 64-bit division per tile is)
#pragma unroll
    for (int k = 0; k < PER_MAX; k++) {
      const int t = t0 + k;
      mycls0[k] = 0u;
      if (k < per && t < tiles) {
        mycls0[k] = cnt[k] == 0u ? 0u : 1u + min((uint32_t)(NCLS - 2), (uint32_t)((float)cnt[k] * cls_scale));  // 1 .. NCLS - 1
        if (tail16) atomicAdd(&s_ccount[mycls0[k]], 1u);
      }
    }
    uint32_t late_from = 0xFFFFFFFFu;
    if (tail16) {  // where every class of whole lists starts in the visiting order: the classes from `late_from` on are cut
      __syncthreads();
      if (wave == 0) {
        const uint32_t v = s_ccount[NCLS - 1 - lane];
        s_cbase[NCLS - 1 - lane] = wave_incl_scan(v) - v;
      }
      __syncthreads();
      late_from = s_busy - min(s_busy, (s_busy * tail16) / 16u);
    }
#pragma unroll
    for (int k = 0; k < PER_MAX; k++) {
      const int t = t0 + k;
      mycls[k] = 0u;
      mysegs[k] = 1u;
      if (k < per && t < tiles) {
        const uint32_t n = cnt[k];
        uint32_t nseg = 1u;
        if (outlier && 4u * n >= outlier * target)  // (the outlier threshold in quarters of the mean)
          nseg = max(2u, min((uint32_t)SEG_MAX, (n + target / 2u) / target));
        else if (tail16 && n >= 2u * WAVE && s_cbase[mycls0[k]] >= late_from)
          nseg = 2u;
        if (nseg > 1u && atomicAdd(&s_extra, nseg - 1u) + (nseg - 1u) > extra_cap) nseg = 1u;  // the frame's extra slots are used up: walked whole
        if (nseg > 1u) ckpt_base[t] = atomicAdd(&s_rec, nseg);  // (<= 2 records per extra slot: inside ckpt_records())
        // a cut tile's FIRST slot stays in the class of the whole list -- the forward walks the list whole from that slot and must
        // start as early as before; the other pieces are sorted by the piece's length
        mycls[k] = mycls0[k];
        if (nseg > 1u) mycls[k] = 1u + min((uint32_t)(NCLS - 2), (uint32_t)((float)segment_len((int)n, (int)nseg) * cls_scale));
        mysegs[k] = nseg;
      }
    }
    if (tail16) {  // count again, now with the pieces
      __syncthreads();
      if ((int)threadIdx.x < NCLS) s_ccount[threadIdx.x] = 0u;
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PER_MAX; k++) {
      const int t = t0 + k;
      if (k < per && t < tiles) {
        atomicAdd(&s_ccount[mycls0[k]], 1u);
        if (mysegs[k] > 1u) atomicAdd(&s_ccount[mycls[k]], mysegs[k] - 1u);
      }
    }
    __syncthreads();
    if (wave == 0) {  // class bases, longest class first: one wave scan instead of 64 dependent LDS round trips
      static_assert(NCLS == WAVE, "one lane per class");
      const uint32_t v = s_ccount[NCLS - 1 - lane], incl = wave_incl_scan(v);
      s_cbase[NCLS - 1 - lane] = incl - v;
      if (lane == WAVE - 1) order[1] = incl;  // visiting slots of this frame: tiles + extra segments
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER_MAX; k++) {
      const int t = t0 + k;
      if (k < per && t < tiles) {
        order[ORDER_HDR + atomicAdd(&s_cbase[mycls0[k]], 1u)] = order_entry((uint32_t)t, 0u, mysegs[k]);
        if (mysegs[k] > 1u) {
          const uint32_t slot = atomicAdd(&s_cbase[mycls[k]], mysegs[k] - 1u);
          for (uint32_t sgm = 1; sgm < mysegs[k]; sgm++) order[ORDER_HDR + slot + sgm - 1u] = order_entry((uint32_t)t, sgm, mysegs[k]);
        }
      }
    }
    return;
  }
  // ---- block modes (2: 2 x 2 tiles, 3: 4 x 2 tiles): the unit that is sorted and dealt to the XCDs is a BLOCK of neighbouring
  // tiles, by the sum of its list lengths; the tiles of a block go to visiting slots with the same slot % 8, i.e. to one XCD,
  // back to back in that XCD's dispatch order.  Neighbouring tiles share most of their Gaussians (a splat of mean radius 13 px
  // touches 6.6 tiles at C3): with single tiles dealt round-robin a record was fetched into up to four L2s (2 x FETCH_SIZE of
  // the blend backward 78 -> 199 MB when the per-tile order came in, profiles/r2f_pmc.csv vs r2g_pmc.csv); a block keeps it in one.
  // The per-tile totals go through LDS (s_base is free: this workgroup scatters nothing).
#pragma unroll
  for (int k = 0; k < PER_MAX; k++) {
    const int t = t0 + k;
    if (k < per && t < tiles) s_base[t] = cnt[k];
  }
  __shared__ uint32_t s_bmax;
  if (threadIdx.x == 0) s_bmax = 0u;
  __syncthreads();
  const int bx = order_mode == 3 ? 4 : 2, by = 2, T = bx * by;
  const int grid_y = tiles / grid_x;
  const int nbx = (grid_x + bx - 1) / bx, nby = (grid_y + by - 1) / by, nsb = nbx * nby;
  constexpr int SB_PER = (HIST_MAX_TILES / 2 + 2 * 128 + HB - 1) / HB;  // blocks per thread: <= tiles / 2 + border blocks of the longest grid side
  uint32_t bsum[SB_PER], bcls[SB_PER];
  uint32_t lmax = 0;
#pragma unroll
  for (int k = 0; k < SB_PER; k++) {
    const int sb = (int)threadIdx.x + k * HB;
    bsum[k] = 0u;
    if (sb < nsb) {
      const int sx = (sb % nbx) * bx, sy = (sb / nbx) * by;
      for (int j = 0; j < T; j++) {
        const int tx = sx + j % bx, ty = sy + j / bx;
        if (tx < grid_x && ty < grid_y) bsum[k] += s_base[ty * grid_x + tx];
      }
      lmax = max(lmax, bsum[k]);
    }
  }
  if (lmax) atomicMax(&s_bmax, lmax);
  __syncthreads();
  const uint32_t bmx = s_bmax;
#pragma unroll
  for (int k = 0; k < SB_PER; k++) {
    const int sb = (int)threadIdx.x + k * HB;
    bcls[k] = 0u;
    if (sb < nsb) {
      bcls[k] = bsum[k] == 0u ? 0u : 1u + (uint32_t)(((uint64_t)bsum[k] * (NCLS - 2)) / bmx);
      atomicAdd(&s_ccount[bcls[k]], 1u);
    }
  }
  __syncthreads();
  if (wave == 0) {
    const uint32_t v = s_ccount[NCLS - 1 - lane];
    s_cbase[NCLS - 1 - lane] = wave_incl_scan(v) - v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SB_PER; k++) {
    const int sb = (int)threadIdx.x + k * HB;
    if (sb < nsb) {
      const uint32_t q = atomicAdd(&s_cbase[bcls[k]], 1u);  // sorted rank of this block
      uint32_t *slot0 = order + ORDER_HDR + (size_t)(q / 8u) * 8u * (uint32_t)T + (q % 8u);
      const int sx = (sb % nbx) * bx, sy = (sb / nbx) * by;
      for (int j = 0; j < T; j++) {
        const int tx = sx + j % bx, ty = sy + j / bx;
        slot0[(size_t)j * 8u] = (tx < grid_x && ty < grid_y) ? (uint32_t)(ty * grid_x + tx) : ORDER_NO_TILE;
      }
    }
  }
  // the last, partial group of eight blocks: padding
  const int nsb8 = (nsb + 7) / 8 * 8;
  for (int e = nsb * T + (int)threadIdx.x; e < nsb8 * T; e += HB) {
    const int q = nsb + (e - nsb * T) / T, j = (e - nsb * T) % T;
    order[ORDER_HDR + (size_t)(q / 8) * 8 * T + (size_t)j * 8 + q % 8] = ORDER_NO_TILE;
  }
}

// bitonic sorting network over keys[0..npow2) by the whole workgroup, starting at merge level kstart (kstart = 2: a full sort;
// kstart = 2048: the 1024-key runs are already sorted, even runs ascending and odd runs descending)
template <typename Ptr>
__device__ __forceinline__ void bitonic_sort_block(Ptr keys, int npow2, int kstart = 2) {
  for (int k = kstart; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (npow2 >> 1); t += blockDim.x) {
        const int lo = ((t / j) * (j << 1)) + (t % j);
        const int hi = lo + j;
        const bool asc = (lo & k) == 0;
        const uint64_t a = keys[lo], b = keys[hi];
        if ((a > b) == asc) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  }
}

// ---- wave-level register bitonic sort: lists of up to 64 * NREG keys, one wave per tile, no LDS, no barriers ----
// Element i of the list lives in register i / 64 of lane i % 64.  A compare-exchange at distance j >= 64 pairs two
// registers of the same lane; at distance j < 64 it pairs lane l with lane l ^ j of the same register (two
// ds_swizzle per 64-bit key for j < 32, two ds_bpermute for j = 32).  The direction of a pair depends on bit k of the element index.
template <int NREG, int K, int J>
__device__ __forceinline__ void bitonic_stage(uint64_t (&key)[NREG], uint32_t lane) {
  if constexpr (J >= WAVE) {
    constexpr int jr = J / WAVE;
#pragma unroll
    for (int r = 0; r < NREG; r++) {
      if ((r & jr) == 0) {
        const bool asc = (((r * WAVE) & K) == 0);  // K >= 128 here: decided by the register index alone
        const uint64_t a = key[r], c = key[r | jr];
        const bool sw = (a > c) == asc;
        key[r] = sw ? c : a;
        key[r | jr] = sw ? a : c;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < NREG; r++) {
      const uint32_t i = (uint32_t)r * WAVE + lane;
      const uint64_t mine = key[r];
      uint32_t lo, hi;
      if constexpr (J < 32) {  // ds_swizzle bit-mode: lane ^ J inside each group of 32 (2.8x cheaper than ds_bpermute here)
        lo = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)mine, (J << 10) | 0x1F);
        hi = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)(mine >> 32), (J << 10) | 0x1F);
      } else {
        lo = (uint32_t)__shfl_xor((int)(uint32_t)mine, J, WAVE);
        hi = (uint32_t)__shfl_xor((int)(uint32_t)(mine >> 32), J, WAVE);
      }
      const uint64_t other = ((uint64_t)hi << 32) | lo;
      const bool lower = (lane & (uint32_t)J) == 0;
      const bool asc = (i & (uint32_t)K) == 0;
      const bool take_min = lower == asc;
      key[r] = take_min ? (mine < other ? mine : other) : (mine > other ? mine : other);
    }
  }
}
template <int NREG, int K, int J>
__device__ __forceinline__ void bitonic_merge(uint64_t (&key)[NREG], uint32_t lane) {
  bitonic_stage<NREG, K, J>(key, lane);
  if constexpr (J > 1) bitonic_merge<NREG, K, J / 2>(key, lane);
}
template <int NREG, int K>
__device__ __forceinline__ void bitonic_levels(uint64_t (&key)[NREG], uint32_t lane) {
  bitonic_merge<NREG, K, K / 2>(key, lane);
  if constexpr (K < NREG * WAVE) bitonic_levels<NREG, K * 2>(key, lane);
}
template <int NREG>
__device__ __forceinline__ void wave_bitonic_sort(uint64_t (&key)[NREG], uint32_t lane) {
  bitonic_levels<NREG, 2>(key, lane);
}

template <int NREG>
__device__ __forceinline__ void wave_sort_tile(const uint64_t *b, int n, uint32_t tile, uint32_t base, uint32_t *point_list,
                                               uint64_t *keys_sorted, uint32_t lane) {
  uint64_t key[NREG];
#pragma unroll
  for (int r = 0; r < NREG; r++) {
    const int i = r * WAVE + (int)lane;
    key[r] = i < n ? b[i] : ~0ull;
  }
  wave_bitonic_sort<NREG>(key, lane);
#pragma unroll
  for (int r = 0; r < NREG; r++) {
    const int i = r * WAVE + (int)lane;
    if (i < n) {
      point_list[base + i] = (uint32_t)key[r];
      keys_sorted[base + i] = ((uint64_t)tile << 32) | (key[r] >> 32);
    }
  }
}


// ---- sort by runs + rank merge (lists of 65 .. 512 keys) ---------------------------------------------------------------
// A full bitonic network over NREG x 64 keys costs log^2 stages over every register and needs a power-of-two size: a tile with
// 260 keys pays for 512.  Here every register is sorted ACROSS THE LANES as its own run of 64 (21 stages, all runs in lockstep),
// the runs go to LDS, and each key finds its final position as  lane + sum over the other runs of (keys smaller than it)  by a
// binary search per run (keys are unique: they contain the Gaussian id).  Work grows with the number of runs actually needed
// (5 runs for 260 keys), not with the next power of two; at C3 (lists of ~240, up to 355) this is ~2x fewer instructions.
constexpr int MERGE_MAX_RUNS = 8;

template <int J>
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v) {
  uint32_t lo, hi;
  if constexpr (J < 32) {  // ds_swizzle bit-mode: lane ^ J inside each group of 32
    lo = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)v, (J << 10) | 0x1F);
    hi = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)(v >> 32), (J << 10) | 0x1F);
  } else {
    lo = (uint32_t)__shfl_xor((int)(uint32_t)v, J, WAVE);
    hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), J, WAVE);
  }
  return ((uint64_t)hi << 32) | lo;
}
// one compare-exchange stage (block size K, distance J) of an ASCENDING bitonic sort of 64 keys held one per lane
template <int NRUN, int K, int J>
__device__ __forceinline__ void run_stage(uint64_t (&key)[NRUN], uint32_t lane) {
  const bool take_min = ((lane & (uint32_t)J) == 0) == ((lane & (uint32_t)K) == 0 || K == WAVE);
#pragma unroll
  for (int r = 0; r < NRUN; r++) {
    const uint64_t mine = key[r], other = lane_xor_u64<J>(mine);
    key[r] = take_min ? (mine < other ? mine : other) : (mine > other ? mine : other);
  }
}
template <int NRUN, int K, int J>
__device__ __forceinline__ void run_merge(uint64_t (&key)[NRUN], uint32_t lane) {
  run_stage<NRUN, K, J>(key, lane);
  if constexpr (J > 1) run_merge<NRUN, K, J / 2>(key, lane);
}
template <int NRUN, int K>
__device__ __forceinline__ void run_levels(uint64_t (&key)[NRUN], uint32_t lane) {
  run_merge<NRUN, K, K / 2>(key, lane);
  if constexpr (K < WAVE) run_levels<NRUN, K * 2>(key, lane);
}

template <int NRUN>
__device__ __forceinline__ void wave_sort_tile_runs(const uint64_t *b, int n, uint32_t tile, uint32_t base, uint32_t *point_list,
                                                    uint64_t *keys_sorted, uint32_t lane, uint64_t *s_runs) {
  uint64_t key[NRUN];
#pragma unroll
  for (int r = 0; r < NRUN; r++) {
    const int i = r * WAVE + (int)lane;
    key[r] = i < n ? b[i] : ~0ull;  // padding sorts behind every real key and is never stored
  }
  run_levels<NRUN, 2>(key, lane);  // every register: one ascending run of 64 across the lanes
#pragma unroll
  for (int r = 0; r < NRUN; r++) s_runs[r * WAVE + (int)lane] = key[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < NRUN; r++) {
    const uint64_t k = key[r];
    uint32_t rank = lane;      // keys of its own run in front of it
#pragma unroll
    for (int o = 0; o < NRUN; o++) {
      if (o == r) continue;
      const uint64_t *run = s_runs + o * WAVE;
      // branch-free: the searches of a key in the other runs (and of the lane's other keys) are independent chains of LDS
      // reads the scheduler can overlap
      uint32_t c = 0u;
#pragma unroll
      for (int step = WAVE / 2; step >= 1; step >>= 1) c += run[c + step - 1] < k ? (uint32_t)step : 0u;
      rank += run[WAVE - 1] < k ? (uint32_t)WAVE : c;  // whole run smaller?
    }
    if (k != ~0ull) {  // not padding
      point_list[base + rank] = (uint32_t)k;
      keys_sorted[base + rank] = ((uint64_t)tile << 32) | (k >> 32);
    }
  }
}

// one wave sorts the list of one tile (n <= SORT_WAVE_MAX); s_runs: MERGE_MAX_RUNS x 64 keys of LDS of its own
__device__ __forceinline__ void wave_sort_any(uint32_t tile, uint32_t lane, const uint2 r, const uint64_t *bucket, uint32_t *point_list,
                                              uint64_t *keys_sorted, uint64_t *s_runs) {
  const int n = (int)(r.y - r.x);
  const uint64_t *b = bucket + r.x;
  if (n <= 64)
    wave_sort_tile<1>(b, n, tile, r.x, point_list, keys_sorted, lane);
  else if (n <= 128)
    wave_sort_tile_runs<2>(b, n, tile, r.x, point_list, keys_sorted, lane, s_runs);
  else if (n <= 192)
    wave_sort_tile_runs<3>(b, n, tile, r.x, point_list, keys_sorted, lane, s_runs);
  else if (n <= 256)
    wave_sort_tile_runs<4>(b, n, tile, r.x, point_list, keys_sorted, lane, s_runs);
  else if (n <= 320)
    wave_sort_tile_runs<5>(b, n, tile, r.x, point_list, keys_sorted, lane, s_runs);
  else if (n <= 384)
    wave_sort_tile_runs<6>(b, n, tile, r.x, point_list, keys_sorted, lane, s_runs);
  else
    wave_sort_tile_runs<8>(b, n, tile, r.x, point_list, keys_sorted, lane, s_runs);
}

__global__ __launch_bounds__(WAVE) void bucket_sort_wave_kernel(const uint2 *ranges, const uint64_t *bucket, uint32_t *point_list,
                                                               uint64_t *keys_sorted, uint32_t *big_list, uint32_t *big_count) {
  __shared__ uint64_t s_runs[MERGE_MAX_RUNS * WAVE];
  const uint32_t tile = blockIdx.x, lane = threadIdx.x;
  const uint2 r = ranges[tile];
  const int n = (int)(r.y - r.x);
  if (n > SORT_WAVE_MAX) {  // left to bucket_sort_kernel: onto its work list unless the scatter kernel already built it (big_list null)
    if (big_list && lane == 0) big_list[atomicAdd(big_count, 1u)] = tile;
    return;
  }
  if (n == 0) return;
  wave_sort_any(tile, lane, r, bucket, point_list, keys_sorted, s_runs);
}

// CAP = LDS capacity in keys; handles tiles with LO < n <= CAP in LDS, n > CAP (TAKES_OVERSIZE) as CAP-sized chunks merged in
// global memory.  Measured in the render() frame (most tiles of the body hold 513..2048 keys): CAP 8192 (64 KB, two workgroups
// per CU) 65 us; 2048 + a second launch for the longer lists 41 + 5 us; 4096 (32 KB) 40 us in one launch.
template <int CAP, int LO, bool TAKES_OVERSIZE>
__device__ __forceinline__ void sort_big_tile(uint32_t tile, uint64_t *s_keys, const uint2 *ranges, uint64_t *bucket,
                                              uint32_t *point_list, uint64_t *keys_sorted) {
  constexpr int SORT_LDS_MAX = CAP;
  const uint2 r = ranges[tile];
  const int n = (int)(r.y - r.x);
  if (n <= LO || (!TAKES_OVERSIZE && n > CAP)) return;  // (the same for every thread of the workgroup)
  uint64_t *b = bucket + r.x;
  if (n <= SORT_LDS_MAX) {
    // Runs of 64 sorted across the lanes in registers (as bucket_sort_wave_kernel), then log2(runs) levels of PAIRWISE MERGE BY RANK
    // between two LDS buffers: every key finds how many keys of the sibling run are smaller by a binary search (keys are unique)
    // and stores itself at (its index in its own run) + (that count).  A thread carries its eight keys through a level in lockstep:
    // eight independent LDS reads per search step.  Padding (~0) sits only at the end of the LAST run, so it is never on the A side
    // of a pair with a non-empty B side and every padded key still gets a position of its own.
    // (Before: np = next power of two, four register sorts of np / 4 keys -- 45 stages over 8 keys per lane for a list of 1,100 --
    // and the last levels of the bitonic network through LDS: 39.7 us per launch in the render() frame, ~700 lists of 513..1,398 keys.
    // This form: 35 us, of which -- measured by leaving phases out -- 4.8 us the empty launch + work-list read, 1.6 us loads and
    // stores, 8.5 us the run sorts, 21 us the five merge levels: 45 search steps of 8 random 8-byte LDS reads per thread.)
    const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    constexpr int NW = 256 / WAVE;
    const int runs = (n + WAVE - 1) / WAVE, N = runs * WAVE;
    uint64_t *src = s_keys, *dst = s_keys + CAP;
    auto sort_runs = [&](auto nr_tag) {
      constexpr int NR = decltype(nr_tag)::value;
      uint64_t key[NR];
#pragma unroll
      for (int q = 0; q < NR; q++) {
        const int i = ((int)wave + q * NW) * WAVE + (int)lane;
        key[q] = i < n ? b[i] : ~0ull;
      }
      run_levels<NR, 2>(key, lane);
#pragma unroll
      for (int q = 0; q < NR; q++) {
        const int i = ((int)wave + q * NW) * WAVE + (int)lane;
        if (i < N) src[i] = key[q];
      }
    };
    const int per_wave = (runs + NW - 1) / NW;
    static_assert(CAP / WAVE / NW <= 8, "at most eight runs per wave");
    if (per_wave <= 2) sort_runs(std::integral_constant<int, 2>{});
    else if (per_wave <= 4) sort_runs(std::integral_constant<int, 4>{});
    else sort_runs(std::integral_constant<int, 8>{});
    __syncthreads();
    constexpr int KPT = CAP / 256;  // keys per thread
    for (int L = WAVE; L < N; L <<= 1) {
      uint64_t key[KPT];
      int other0[KPT], len[KPT], pos[KPT];
      uint32_t c[KPT];
#pragma unroll
      for (int q = 0; q < KPT; q++) {
        const int i = (int)threadIdx.x + q * 256;
        const int a0 = (i / (2 * L)) * (2 * L), a1 = min(N, a0 + L), b1 = min(N, a1 + L);
        const bool in_a = i < a1;
        key[q] = i < N ? src[i] : ~0ull;
        other0[q] = in_a ? a1 : a0;
        len[q] = i < N ? (in_a ? b1 - a1 : a1 - a0) : 0;
        pos[q] = a0 + (in_a ? i - a0 : i - a1);
        c[q] = 0u;
      }
      // (every read is unconditional -- the index is clamped into the list, the comparison masked -- so that a step is eight LDS
      // reads in flight and one wait, not eight round trips behind eight branches)
      for (int step = L / 2; step >= 1; step >>= 1) {
        uint64_t v[KPT];
#pragma unroll
        for (int q = 0; q < KPT; q++) v[q] = src[min(other0[q] + (int)c[q] + step - 1, N - 1)];
#pragma unroll
        for (int q = 0; q < KPT; q++) c[q] += ((int)c[q] + step - 1 < len[q] && v[q] < key[q]) ? (uint32_t)step : 0u;
      }
      {  // (the last element of the sibling run: the steps above cover indices 0 .. L - 2)
        uint64_t v[KPT];
#pragma unroll
        for (int q = 0; q < KPT; q++) v[q] = src[min(other0[q] + (int)c[q], N - 1)];
#pragma unroll
        for (int q = 0; q < KPT; q++) {
          c[q] += ((int)c[q] < len[q] && v[q] < key[q]) ? 1u : 0u;
          const int i = (int)threadIdx.x + q * 256;
          if (i < N) dst[pos[q] + (int)c[q]] = key[q];
        }
      }
      __syncthreads();
      uint64_t *t = src;
      src = dst;
      dst = t;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const uint64_t k = src[i];
      point_list[r.x + i] = (uint32_t)k;
      keys_sorted[r.x + i] = ((uint64_t)tile << 32) | (k >> 32);
    }
  } else {
    // rare: a list longer than LDS.  Sort LDS-sized chunks, then merge the sorted runs pairwise in global memory
    // (each element finds its rank in the sibling run by binary search; keys are unique).  keys_sorted's slice of
    // this tile is the ping-pong buffer.
    for (int c0 = 0; c0 < n; c0 += SORT_LDS_MAX) {
      const int m = min(SORT_LDS_MAX, n - c0);
      int np = 1;
      while (np < m) np <<= 1;
      for (int i = threadIdx.x; i < np; i += blockDim.x) s_keys[i] = i < m ? b[c0 + i] : ~0ull;
      __syncthreads();
      bitonic_sort_block(s_keys, np);
      __syncthreads();
      for (int i = threadIdx.x; i < m; i += blockDim.x) b[c0 + i] = s_keys[i];
      __syncthreads();
    }
    uint64_t *src = b, *dst = keys_sorted + r.x;
    for (int L = SORT_LDS_MAX; L < n; L <<= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int a0 = (i / (2 * L)) * (2 * L);
        const int a1 = min(n, a0 + L), b1 = min(n, a1 + L);
        const uint64_t key = src[i];
        const bool in_a = i < a1;
        const uint64_t *other = in_a ? src + a1 : src + a0;
        int len = in_a ? b1 - a1 : a1 - a0;
        int lo = 0;  // number of keys in the sibling run that are smaller than `key`
        while (len > 0) {
          const int half = len >> 1;
          if (other[lo + half] < key) {
            lo += half + 1;
            len -= half + 1;
          } else {
            len = half;
          }
        }
        dst[a0 + (in_a ? i - a0 : i - a1) + lo] = key;
      }
      __syncthreads();
      uint64_t *t = src;
      src = dst;
      dst = t;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const uint64_t k = src[i];
      point_list[r.x + i] = (uint32_t)k;
      keys_sorted[r.x + i] = ((uint64_t)tile << 32) | (k >> 32);
    }
  }
}

// Lists longer than SORT_WAVE_MAX (512): none at C3, most of the non-empty tiles of a close-up of a body.  The wave kernel leaves
// their tile ids on a work list and a fixed grid of workgroups strides over that list: nothing but one counter read when the
// list is empty, and an even share per workgroup wherever the long lists sit on the screen.  (One workgroup per tile: 5 us of
// empty workgroups at C3; workgroups striding over the TILES: 113 instead of 50 us in the render() frame, because the tiles
// of a body fall on a quarter of the workgroups.)
template <int CAP, int LO, bool TAKES_OVERSIZE>
__global__ __launch_bounds__(256) void bucket_sort_kernel(const uint2 *ranges, uint64_t *bucket, uint32_t *point_list,
                                                         uint64_t *keys_sorted, const uint32_t *big_list, const uint32_t *big_count) {
  __shared__ uint64_t s_keys[2 * CAP];  // two buffers: the merge levels go back and forth between them
  const uint32_t count = *big_count;
  for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
    sort_big_tile<CAP, LO, TAKES_OVERSIZE>(big_list[k], s_keys, ranges, bucket, point_list, keys_sorted);
    __syncthreads();  // s_keys is reused by the next tile
  }
}

// Both sorts in ONE launch (histogram path: the work list of the long lists exists before the launch).  The first n_big workgroups
// stride over the work list as bucket_sort_kernel does -- first, because in a close-up of a body they are the longest pole (35 us
// against 17 for the short lists) -- and every other workgroup sorts four tiles, a wave each, as bucket_sort_wave_kernel does
// (its 4 KB of run buffers per wave lie inside the 32 KB the long-list role needs).  One launch less per frame (the empty long-list
// launch cost 4.7 us at C3) and the two sorts overlap without a second stream (fork / join by events cost more than it gained).
template <int CAP, int LO, bool TAKES_OVERSIZE>
__global__ __launch_bounds__(256) void bucket_sort_both_kernel(const uint2 *ranges, uint64_t *bucket, uint32_t *point_list,
                                                              uint64_t *keys_sorted, const uint32_t *big_list, const uint32_t *big_count,
                                                              uint32_t n_big, uint32_t tiles) {
  __shared__ uint64_t s_keys[2 * CAP];
  static_assert(2 * CAP >= 4 * MERGE_MAX_RUNS * WAVE, "the four waves' run buffers fit the long-list buffers");
  if (blockIdx.x < n_big) {
    const uint32_t count = *big_count;
    for (uint32_t k = blockIdx.x; k < count; k += n_big) {
      sort_big_tile<CAP, LO, TAKES_OVERSIZE>(big_list[k], s_keys, ranges, bucket, point_list, keys_sorted);
      __syncthreads();  // s_keys is reused by the next tile
    }
    return;
  }
  const uint32_t wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  const uint32_t tile = (blockIdx.x - n_big) * 4u + wave;
  if (tile >= tiles) return;  // (no workgroup barrier in this role: the waves are independent)
  const uint2 r = ranges[tile];
  const int n = (int)(r.y - r.x);
  if (n == 0 || n > SORT_WAVE_MAX) return;
  wave_sort_any(tile, lane, r, bucket, point_list, keys_sorted, s_keys + wave * (MERGE_MAX_RUNS * WAVE));
}

// The histogram / scatter workgroups split the P Gaussians EVENLY over a multiple of the CU count (at most HB each, at least one
// preprocess block): 200k Gaussians on 256 CUs = 256 workgroups of 782 instead of 196 of 1024 with 60 CUs idle.
static int cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
static void hist_split(int P, int &n_sb, int &hbv) {
  const int cus = cu_count();
  const int rounds = (int)(((size_t)P + (size_t)cus * HB - 1) / ((size_t)cus * HB));
  int n = cus * (rounds > 0 ? rounds : 1);
  const int most = pre_blocks(P) > 0 ? pre_blocks(P) : 1;  // not less than one preprocess block per workgroup
  if (n > most) n = most;
  hbv = (P + n - 1) / n;
  if (hbv < 1) hbv = 1;
  n_sb = (P + hbv - 1) / hbv;
  if (n_sb < 1) n_sb = 1;
}

bool bucket_uses_hist(const Options &opt, int P, size_t tiles, size_t capacity) {
  int n_sb, hbv;
  hist_split(P, n_sb, hbv);
  // the workgroup x tile table (+ one start word per workgroup) borrows keys_s, which nothing touches before the sort kernels
  // write their final keys into it
  return opt.bucket_hist && tiles <= (size_t)HIST_MAX_TILES &&
         ((size_t)n_sb * tiles + (size_t)n_sb) * sizeof(uint32_t) <= capacity * sizeof(uint64_t);
}

int bucket_binning(const GeomState &g, const int *radii, int P, int grid_x, int grid_y, size_t capacity, bool device_sized,
                   BinningState &b, uint2 *ranges, uint32_t *order, uint32_t *ckpt_base, int segments, uint32_t *dev_status,
                   bool check_prefilter, bool scan_fused, const Options &opt, hipStream_t stream, int debug) {
  const size_t tiles = (size_t)grid_x * grid_y;
  if (grid_x >= 1024 || grid_y >= 1024) {
    set_error("image larger than 16368 px per side is not supported by the packed tile rect");
    return GSR_EINVAL;
  }
  if (!b.tile_counts) {
    set_error("binning buffer was not sized for the tile-bucket back-end");
    return GSR_EINVAL;
  }
  const uint32_t cap32 = capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity;
  // instance ranks live in the (otherwise unused in this back-end) vals_a array
  int n_sb, hbv;
  hist_split(P, n_sb, hbv);
  const bool hist = bucket_uses_hist(opt, P, tiles, capacity);
  if (scan_fused && !hist) {
    set_error("bucket_binning: the fused block scan needs the histogram path");
    return GSR_EINVAL;
  }
  if (hist) {
    uint32_t *table = reinterpret_cast<uint32_t *>(b.keys_s);
    uint32_t *wg_start = table + (size_t)n_sb * tiles;
    const dim3 hg(n_sb), hb(HB);
    if (scan_fused) {
      if (opt.tile_cull)
        hipLaunchKernelGGL((bucket_hist_kernel<true, true>), hg, hb, 0, stream, g, radii, P, grid_x, grid_y, (int)tiles, table, b.vals_a,
                           b.vals_s, cap32, hbv, wg_start);
      else
        hipLaunchKernelGGL((bucket_hist_kernel<false, true>), hg, hb, 0, stream, g, radii, P, grid_x, grid_y, (int)tiles, table, b.vals_a,
                           b.vals_s, cap32, hbv, wg_start);
    } else {
      if (opt.tile_cull)
        hipLaunchKernelGGL((bucket_hist_kernel<true, false>), hg, hb, 0, stream, g, radii, P, grid_x, grid_y, (int)tiles, table, b.vals_a,
                           b.vals_s, cap32, hbv, wg_start);
      else
        hipLaunchKernelGGL((bucket_hist_kernel<false, false>), hg, hb, 0, stream, g, radii, P, grid_x, grid_y, (int)tiles, table, b.vals_a,
                           b.vals_s, cap32, hbv, wg_start);
    }
    GSR_LAUNCH_CHECK(stream, debug);
    hipLaunchKernelGGL(bucket_hist_prefix_kernel, dim3((unsigned)((tiles + WAVE - 1) / WAVE)), dim3(PW * WAVE), 0, stream, table, n_sb,
                       (int)tiles, b.tile_counts);
    GSR_LAUNCH_CHECK(stream, debug);
    if (!device_sized && capacity == 0) {  // nothing to bin (R = 0 read by the host): only the empty ranges and the status
      hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, b.tile_counts, b.tile_cursor, ranges, (int)tiles,
                         g.total, cap32, dev_status, 1, check_prefilter ? 1 : 0, order);
      GSR_LAUNCH_CHECK(stream, debug);
      return GSR_OK;
    }
    hipLaunchKernelGGL(bucket_scatter_hist_kernel, dim3(n_sb + 1), dim3(HB), 0, stream, g, wg_start, (int)tiles, b.tile_counts, table,
                       b.vals_a, b.vals_s, b.keys_a, cap32, ranges, dev_status, check_prefilter ? 1 : 0, order, opt.tile_order, grid_x,
                       ckpt_base, segments, b.tile_cursor);
    GSR_LAUNCH_CHECK(stream, debug);
  } else {
    const int CSTRIDE = opt.bucket_cstride;
    GSR_HIP(zero_async(b.tile_counts, tiles * CSTRIDE * sizeof(uint32_t), stream));
    if (opt.tile_cull)
      hipLaunchKernelGGL(bucket_count_kernel<true>, dim3(pre_blocks(P)), dim3(PRE_BLOCK), 0, stream, g, radii, P, grid_x, grid_y,
                         b.tile_counts, b.vals_a, cap32, CSTRIDE);
    else
      hipLaunchKernelGGL(bucket_count_kernel<false>, dim3(pre_blocks(P)), dim3(PRE_BLOCK), 0, stream, g, radii, P, grid_x, grid_y,
                         b.tile_counts, b.vals_a, cap32, CSTRIDE);
    GSR_LAUNCH_CHECK(stream, debug);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, b.tile_counts, b.tile_cursor, ranges, (int)tiles,
                       g.total, cap32, dev_status, CSTRIDE, check_prefilter ? 1 : 0, order);
    GSR_LAUNCH_CHECK(stream, debug);
    if (!device_sized && capacity == 0) return GSR_OK;
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(pre_blocks(P)), dim3(PRE_BLOCK), 0, stream, g, radii, P, grid_x, grid_y,
                       b.tile_cursor, b.vals_a, b.keys_a, cap32);
    GSR_LAUNCH_CHECK(stream, debug);
  }
  // The two sort kernels.  Work list of the long lists in tile_cursor (free by now: the scatter kernels are its last readers), its
  // counter in total[2]: written by the scatter kernel's order builder on the histogram path, by the wave sort on the atomic path.
  // (Measured and dropped: the long-list sort on a side stream NEXT TO the wave sort, fork / join by events -- the kernels are
  // independent once the list is built up front.  The two event dependencies cost more than the overlap gains on this runtime: C3
  // binning 61 -> 71..77 us, render() as one graph 0.803 -> 0.836 ms.)
  const unsigned big_grid = 4u * (unsigned)cu_count();  // SORT_BIG keys = 32 KB of LDS each: up to five per CU
  const unsigned n_big = (unsigned)(tiles < big_grid ? tiles : big_grid);
  if (hist && opt.bucket_sort_merged) {  // (the work list is complete: the scatter kernel's order builder wrote it)
    hipLaunchKernelGGL((bucket_sort_both_kernel<SORT_BIG, SORT_WAVE_MAX, true>), dim3(n_big + (unsigned)((tiles + 3) / 4)), dim3(256), 0,
                       stream, ranges, b.keys_a, b.vals_s, b.keys_s, b.tile_cursor, g.total + 2, n_big, (uint32_t)tiles);
    GSR_LAUNCH_CHECK(stream, debug);
    return GSR_OK;
  }
  hipLaunchKernelGGL(bucket_sort_wave_kernel, dim3((unsigned)tiles), dim3(WAVE), 0, stream, ranges, b.keys_a, b.vals_s, b.keys_s,
                     hist ? (uint32_t *)nullptr : b.tile_cursor, g.total + 2);
  GSR_LAUNCH_CHECK(stream, debug);
  // (a separate 16 KB-LDS instantiation for 1025..2048 keys was measured: slower -- the register sorts of the runs, not the
  // LDS occupancy, bound this kernel)
  hipLaunchKernelGGL((bucket_sort_kernel<SORT_BIG, SORT_WAVE_MAX, true>), dim3(n_big), dim3(256),
                     0, stream, ranges, b.keys_a, b.vals_s, b.keys_s, b.tile_cursor, g.total + 2);
  GSR_LAUNCH_CHECK(stream, debug);
  return GSR_OK;
}

}  // namespace gsr
