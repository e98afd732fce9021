// radix_sort.hip -- device-wide STABLE least-significant-digit radix sort of (key, value) pairs for gfx950.
//
// Replaces cub::DeviceRadixSort::SortPairs at CR/rasterizer_impl.cu:305-310 (u64 keys = tile << 32 | depth bits,
// sorted on the low 32 + getHigherMsb(tiles) bits) and at SK/simple_knn.cu:213 (u32 Morton codes).
//
// Structure per 8-bit digit: (1) per-block digit histogram, (2) one exclusive scan over the digit-major
// [256][blocks] table, (3) stable scatter.  Ranking is wave64-native: the lanes of a wave that hold the same
// digit are found with 8 ballots ("match"), their rank is a popcount of the lower lanes, and one leader lane per
// digit bumps a per-wave LDS counter -- no LDS atomics, no 32-lane assumptions.  Keys are visited in wave-contiguous
// runs (wave w of a block owns 1024 consecutive keys, 64 at a time), so (block, wave, round, lane) order == input
// order, which is what makes the scatter stable.
#include "gsr_common.h"

namespace gsr {

__device__ __forceinline__ uint64_t lanemask_lt() {
  const uint32_t l = lane_id();
  return l == 0 ? 0ull : (~0ull >> (64 - l));
}

// lanes (among `valid`) holding the same 8-bit digit as this lane
__device__ __forceinline__ uint64_t digit_peers(uint32_t d, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const bool bit = (d >> b) & 1u;
    const uint64_t m = __ballot(bit && valid);
    peers &= bit ? m : ~m;
  }
  return peers;
}

template <typename K>
__global__ __launch_bounds__(SORT_BLOCK) void radix_hist_kernel(const K *__restrict__ keys, size_t n, int shift,
                                                                uint32_t *__restrict__ hist, uint32_t nblocks) {
  __shared__ volatile uint32_t h[SORT_BLOCK / WAVE][256];
  const int wave = threadIdx.x / WAVE;
  const uint32_t lane = lane_id();
  for (int w = 0; w < SORT_BLOCK / WAVE; w++) h[w][threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * SORT_TILE + (size_t)wave * (SORT_ITEMS * WAVE);
#pragma unroll 4
  for (int it = 0; it < SORT_ITEMS; it++) {
    const size_t idx = base + (size_t)it * WAVE + lane;
    const bool valid = idx < n;
    const uint32_t d = valid ? (uint32_t)((keys[idx] >> shift) & 0xFF) : 0u;
    const uint64_t peers = digit_peers(d, valid);
    if (valid) {
      const uint32_t leader = (uint32_t)__builtin_ctzll(peers);
      if (lane == leader) h[wave][d] = h[wave][d] + (uint32_t)__builtin_popcountll(peers);
    }
  }
  __syncthreads();
  uint32_t s = 0;
  for (int w = 0; w < SORT_BLOCK / WAVE; w++) s += h[w][threadIdx.x];
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = s;
}

// Scan of the digit-major [256][nblocks] histogram, one workgroup per digit row: row d becomes the exclusive
// prefix over the blocks, and totals[d] the digit's global count.  (The scatter kernel turns the 256 totals into
// digit bases itself, so a pass needs no single-workgroup scan over 256*nblocks words.)
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t *__restrict__ hist, uint32_t *__restrict__ totals,
                                                        uint32_t nblocks) {
  __shared__ uint32_t wtot[256 / WAVE];
  __shared__ uint32_t carry_s;
  uint32_t *row = hist + (size_t)blockIdx.x * nblocks;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  for (uint32_t base = 0; base < nblocks; base += 256) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblocks ? row[i] : 0u;
    const uint32_t incl_w = wave_incl_scan(v);
    if (lane == WAVE - 1) wtot[wave] = incl_w;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wtot[w];
    const uint32_t carry = carry_s;
    if (i < nblocks) row[i] = carry + woff + incl_w - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + incl_w;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

template <typename K>
__global__ __launch_bounds__(SORT_BLOCK) void radix_scatter_kernel(const K *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                                   K *__restrict__ kout, uint32_t *__restrict__ vout, size_t n,
                                                                   int shift, const uint32_t *__restrict__ hist,
                                                                   const uint32_t *__restrict__ totals, uint32_t nblocks) {
  __shared__ volatile uint32_t wcount[SORT_BLOCK / WAVE][256];
  __shared__ uint32_t dtot[SORT_BLOCK / WAVE];
  const int wave = threadIdx.x / WAVE;
  const uint32_t lane = lane_id();
  const uint64_t lt = lanemask_lt();
  for (int w = 0; w < SORT_BLOCK / WAVE; w++) wcount[w][threadIdx.x] = 0;
  // digit base = exclusive scan of the 256 digit totals (thread d owns digit d) + this block's prefix in row d
  uint32_t gbase;
  {
    const uint32_t tot = totals[threadIdx.x];
    const uint32_t incl_w = wave_incl_scan(tot);
    if (lane == WAVE - 1) dtot[wave] = incl_w;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += dtot[w];
    gbase = woff + incl_w - tot + hist[(size_t)threadIdx.x * nblocks + blockIdx.x];
  }
  __syncthreads();
  K k[SORT_ITEMS];
  uint32_t v[SORT_ITEMS];
  uint32_t rank[SORT_ITEMS];
  const size_t base = (size_t)blockIdx.x * SORT_TILE + (size_t)wave * (SORT_ITEMS * WAVE);
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; it++) {
    const size_t idx = base + (size_t)it * WAVE + lane;
    const bool valid = idx < n;
    k[it] = valid ? kin[idx] : (K)0;
    v[it] = valid ? vin[idx] : 0u;
  }
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; it++) {
    const size_t idx = base + (size_t)it * WAVE + lane;
    const bool valid = idx < n;
    const uint32_t d = (uint32_t)((k[it] >> shift) & 0xFF);
    const uint64_t peers = digit_peers(d, valid);
    uint32_t r = 0;
    if (valid) {
      const uint32_t prev = wcount[wave][d];
      r = prev + (uint32_t)__builtin_popcountll(peers & lt);
      const uint32_t leader = (uint32_t)__builtin_ctzll(peers);
      if (lane == leader) wcount[wave][d] = prev + (uint32_t)__builtin_popcountll(peers);
    }
    rank[it] = r;
  }
  __syncthreads();
  {  // per digit: global base of this block + exclusive prefix over the waves
    uint32_t run = gbase;
    for (int w = 0; w < SORT_BLOCK / WAVE; w++) {
      const uint32_t c = wcount[w][threadIdx.x];
      wcount[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; it++) {
    const size_t idx = base + (size_t)it * WAVE + lane;
    if (idx < n) {
      const uint32_t d = (uint32_t)((k[it] >> shift) & 0xFF);
      const uint32_t pos = wcount[wave][d] + rank[it];
      kout[pos] = k[it];
      vout[pos] = v[it];
    }
  }
}

template <typename K>
static int radix_sort_impl(size_t n, const K *in_k, const uint32_t *in_v, K *x_k, uint32_t *x_v, K *y_k, uint32_t *y_v,
                           int end_bit, uint32_t *hist, hipStream_t stream, int debug) {
  if (n == 0) return GSR_OK;
  if (n >= 0xFFFFFFFFull) {
    set_error("radix sort: more than 2^32-1 items");
    return GSR_EINVAL;
  }
  const int passes = radix_passes(end_bit);
  const uint32_t nblocks = (uint32_t)sort_blocks(n);
  const K *src_k = in_k;
  const uint32_t *src_v = in_v;
  for (int p = 0; p < passes; p++) {
    K *dst_k = (p % 2 == 0) ? x_k : y_k;
    uint32_t *dst_v = (p % 2 == 0) ? x_v : y_v;
    const int shift = 8 * p;
    hipLaunchKernelGGL(radix_hist_kernel<K>, dim3(nblocks), dim3(SORT_BLOCK), 0, stream, src_k, n, shift, hist, nblocks);
    GSR_LAUNCH_CHECK(stream, debug);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(256), 0, stream, hist, hist + (size_t)256 * nblocks, nblocks);
    GSR_LAUNCH_CHECK(stream, debug);
    hipLaunchKernelGGL(radix_scatter_kernel<K>, dim3(nblocks), dim3(SORT_BLOCK), 0, stream, src_k, src_v, dst_k, dst_v, n,
                       shift, hist, hist + (size_t)256 * nblocks, nblocks);
    GSR_LAUNCH_CHECK(stream, debug);
    src_k = dst_k;
    src_v = dst_v;
  }
  return GSR_OK;
}

int radix_sort_u64(size_t n, const uint64_t *in_k, const uint32_t *in_v, uint64_t *x_k, uint32_t *x_v, uint64_t *y_k,
                   uint32_t *y_v, int end_bit, uint32_t *hist, hipStream_t stream, int debug) {
  return radix_sort_impl<uint64_t>(n, in_k, in_v, x_k, x_v, y_k, y_v, end_bit, hist, stream, debug);
}
int radix_sort_u32(size_t n, const uint32_t *in_k, const uint32_t *in_v, uint32_t *x_k, uint32_t *x_v, uint32_t *y_k,
                   uint32_t *y_v, int end_bit, uint32_t *hist, hipStream_t stream, int debug) {
  return radix_sort_impl<uint32_t>(n, in_k, in_v, x_k, x_v, y_k, y_v, end_bit, hist, stream, debug);
}

}  // namespace gsr
