// blend_fwd.hip -- per-tile front-to-back alpha blending (replaces renderCUDA, CR/forward.cu:261-383).
//
// Work decomposition (wave64-first, not a 16x16-thread CUDA block with block barriers):
//   * a 16x16 tile is four 8x8 quadrants; a lane owns one pixel of a quadrant (lane = (y&7)*8 + (x&7));
//   * a wave owns SLOTS quadrants of a tile and never synchronises with another wave; with SLOTS = 1 the four waves of a tile
//     are launched as one workgroup only to share a CU (one L1 for the tile's records) (SLOTS = 1, 2 or 4 pixels per lane; tuning knob
//     "blend_fwd_waves" = 4/SLOTS waves per tile).  Waves never synchronise with each other: no __syncthreads;
//   * the wave walks the tile's depth-sorted instance list 64 entries at a time.  Lane j fetches entry j's 48-byte
//     SplatRec and tests its conservative cull box (hx, hy: outside it alpha < 1/255 for sure) against the wave's
//     pixel rectangle; a ballot + mbcnt compacts the survivors into LDS (three float4 planes, original list
//     position kept).  Only survivors reach the per-pixel loop, where every LDS read is a single-address broadcast;
//   * a second ballot in front of the exp() skips a quadrant when no lane can reach alpha >= 1/255, and the wave
//     retires as soon as all its pixels are saturated (the reference votes per 256-thread block, CR/forward.cu:314).
// Results follow the reference's rules exactly: skip power > 0, alpha = min(0.99, o*exp(power)), skip alpha < 1/255,
// stop (without blending) when T*(1-alpha) < 1e-4, alpha image = sum of weights, depth image = sum z*w.  Culling is
// conservative, so it only removes work whose outcome is "skip".
#include "gsr_common.h"

namespace gsr {

// bijective XCD-aware remap: consecutive work items (which share Gaussians) land on the same XCD / L2
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t n) {
  const uint32_t q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
  const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

// CE > 0: fused multi-feature blend -- CE extra colour channels (a.extra[P][CE]) are composited with the same weights in
// the same pass (the reference rasterises seven times per frame for them, gaussian_renderer/__init__.py:203-272).
template <int SLOTS, int CE>
__global__ __launch_bounds__(WAVE * (SLOTS == 1 ? 4 : 1)) void blend_forward_kernel(const BlendFwdArgs a) {
  // SLOTS == 1: the four quadrant waves of a tile form ONE workgroup -- still independent of each other, there is no workgroup
  // barrier anywhere -- so that they run on one CU and fetch the tile's records through one L1: 104 -> 97 us at C3, 153 -> 146 us
  // in the render() frame (the same layout changed nothing for the backward kernels, which keep one wave per workgroup)
  constexpr int WPG = SLOTS == 1 ? 4 : 1;
  const uint32_t wv = threadIdx.x / WAVE;
  constexpr int WPT = 4 / SLOTS;  // waves per tile
  __shared__ __attribute__((aligned(16))) float s_x_all[CE > 0 ? WPG * WAVE * CE : 4];  // survivors' extra channels
  __shared__ float4 s0_all[WPG * WAVE];     // x, y, qa, qb      (qa = -conic_a log2(e)/2, qb = -conic_b log2(e))
  // what the cut-off test needs sits in s0 + the first half of s1, what only a blending survivor needs in the second half of s1 +
  // s2: each branch's LDS reads are whole 8- / 16-byte accesses (a 4-byte broadcast read costs as many LDS cycles as an 8-byte one,
  // and the kernel runs the LDS at ~2/3 of its cycles)
  __shared__ float4 s1_all[WPG * WAVE];     // qc, log2(255*opacity) | list position + 1 (bits), opacity   (qc = -conic_c log2(e)/2)
  __shared__ float4 s2_all[WPG * WAVE];     // r, g, b, depth
  float *s_x = s_x_all + (CE > 0 ? wv * WAVE * CE : 0);
  float4 *s0 = s0_all + wv * WAVE, *s1 = s1_all + wv * WAVE, *s2 = s2_all + wv * WAVE;

  uint32_t tile, part;
  if constexpr (WPG == 4) {
    // (the launch covers tile_slots_max() workgroups; the frame's own mode word says how many visiting slots it has)
    const int omode = tile_order_mode(a.order);
    const uint32_t n_slots = tile_slots(a.grid_x, a.grid_y, omode);
    const uint32_t slot = omode ? blockIdx.x : (blockIdx.x < n_slots ? xcd_remap(blockIdx.x, n_slots) : n_slots);
    tile = tile_of_slot(a.order, omode, slot, n_slots);
    part = wv;
    if (tile == ORDER_NO_TILE) return;  // (workgroup-uniform)
  } else {
    const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
    tile = item / WPT, part = item % WPT;
  }
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const uint32_t lane = threadIdx.x % WAVE;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);

  float pxf[SLOTS], pyf[SLOTS], T[SLOTS], C0[SLOTS], C1[SLOTS], C2[SLOTS], Dp[SLOTS], Wt[SLOTS];
  float X[SLOTS][CE > 0 ? CE : 1];
  uint32_t last[SLOTS];
  bool inside[SLOTS];
  // "done" as a float folded into the alpha-threshold test (0 while the pixel is live, -1e30 once it is saturated or if it
  // lies outside the image): one v_add instead of a loop-carried lane mask and the scalar juggling that came with it
  float dbias[SLOTS];  // (kept as the THRESHOLD of that test: -0.02 while live, +1e30 once done -- one add less per tested survivor)
  int pixid[SLOTS];
  // pixel rectangle of this wave (pixel centres), for the cull test
  const int q0 = (int)part * SLOTS, q1 = q0 + SLOTS - 1;
  const float rx0 = (float)(tx * TILE + (q0 & 1) * 8), rx1 = (float)(tx * TILE + (q1 & 1) * 8 + 7);
  const float ry0 = (float)(ty * TILE + (q0 >> 1) * 8), ry1 = (float)(ty * TILE + (q1 >> 1) * 8 + 7);
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + s;
    const int px = tx * TILE + (q & 1) * 8 + (int)(lane & 7);
    const int py = ty * TILE + (q >> 1) * 8 + (int)(lane >> 3);
    pxf[s] = (float)px;
    pyf[s] = (float)py;
    inside[s] = px < a.W && py < a.H;
    dbias[s] = inside[s] ? -0.02f : 1e30f;
    pixid[s] = py * a.W + px;
    T[s] = 1.0f;
    C0[s] = C1[s] = C2[s] = Dp[s] = Wt[s] = 0.f;
#pragma unroll
    for (int c = 0; c < (CE > 0 ? CE : 1); c++) X[s][c] = 0.f;
    last[s] = 0;
  }
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));

  // Two-stage prefetch of the list: while batch b is blended, the records of batch b + 1 and the list entries of batch b + 2 are
  // already on their way (a close-up of a body keeps only ~800 of the 4096 tiles busy: three waves per SIMD, each walking
  // ~25 batches, cannot hide a point_list -> record load chain per batch behind one another)
  uint32_t id_a = 0, id_b = 0;  // list entries of the NEXT batch / of the one after it
  float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;  // records of the current batch (prefetched)
  uint32_t id_cur = 0;
  if ((int)lane < n) {
    id_cur = a.point_list[range.x + lane];
    const float4 *src = reinterpret_cast<const float4 *>(a.recs + id_cur);
    p0 = src[0];
    p1 = src[1];
    p2 = src[2];
  }
  if ((int)lane + WAVE < n) id_a = a.point_list[range.x + lane + WAVE];
  for (int base = 0; base < n; base += WAVE) {
    bool all_done = true;
#pragma unroll
    for (int s = 0; s < SLOTS; s++) all_done = all_done && (dbias[s] > 0.f);
    if (__ballot(!all_done) == 0ull) break;

    // ---- this batch's 64 list entries (already in registers), the next batches' loads, cull against the wave's rectangle,
    // compact the survivors into LDS
    const int idx = base + (int)lane;
    const float4 r0 = p0, r1c = p1, r2 = p2;
    const uint32_t id = id_cur;
    (void)id_b;
    bool keep = false;
    if (idx < n) keep = (r0.x + r2.z >= rx0) && (r0.x - r2.z <= rx1) && (r0.y + r2.w >= ry0) && (r0.y - r2.w <= ry1);
    float4 r1 = make_float4(0, 0, 0, 0);
    if (keep) {  // second, exact filter: ellipse {alpha >= 1/255} against the wave's pixel rectangle
      r1 = r1c;
      keep = ellipse_hits_rect(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, rx0, rx1, ry0, ry1);
    }
    const uint64_t kmask = __ballot(keep);
    const int cnt = __builtin_popcountll(kmask);
    if (keep) {
      const int slot = __builtin_popcountll(kmask & lt);
      // exponent in base 2: p2 = power * log2(e) = dx (qa dx + qb dy) + qc dy dy
      constexpr float L2E = 1.4426950408889634f;
      s0[slot] = make_float4(r0.x, r0.y, (-0.5f * L2E) * r0.z, -L2E * r0.w);
      s1[slot] = make_float4((-0.5f * L2E) * r1.x, __builtin_amdgcn_logf(255.0f * r1.y), __uint_as_float((uint32_t)(idx + 1)), r1.y);
      s2[slot] = make_float4(r1.w, r2.x, r2.y, r1.z);
      if (CE > 0) {
        const float2 *xs = reinterpret_cast<const float2 *>(a.extra + (size_t)id * CE);
#pragma unroll
        for (int q = 0; q < CE / 2; q++) reinterpret_cast<float2 *>(&s_x[slot * CE])[q] = xs[q];
      }
    }
    // the next batch's records and the list entries of the batch after it go out AFTER this batch's channel-colour loads
    // (loads retire in order: waiting for those must not wait for these).  Measured in the render() frame and dropped:
    // prefetching the channel colours of the next batch as well (18 more registers: 221 vs 209 us); staging the channel
    // colours half a batch at a time to fit a sixth wave per SIMD (amdgpu_waves_per_eu(6), no prefetch: 163 vs 153 us).
    id_cur = id_a;
    if (idx + WAVE < n) {
      const float4 *src = reinterpret_cast<const float4 *>(a.recs + id_a);
      p0 = src[0];
      p1 = src[1];
      p2 = src[2];
    }
    if (idx + 2 * WAVE < n) id_a = a.point_list[range.x + idx + 2 * WAVE];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- blend the survivors
    for (int k = 0; k < cnt; k++) {
      const float4 g0 = s0[k];
      const float4 g1 = s1[k];
      const float4 g2 = s2[k];
#pragma unroll
      for (int s = 0; s < SLOTS; s++) {
        const float dx = g0.x - pxf[s], dy = g0.y - pyf[s];
        const float p2 = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;  // power * log2(e)
        // cheap necessary condition for alpha >= 1/255:  log2(255*o) + power*log2(e) >= 0  (0.02 safety margin)
        const bool pre = !(p2 > 0.0f) && ((p2 + g1.y) >= dbias[s]);
        if (__ballot(pre) != 0ull) {
          const float alpha = fminf(0.99f, g1.w * __builtin_amdgcn_exp2f(p2));
          const bool hit = pre && !(alpha < 1.0f / 255.0f);
          const float test_T = T[s] * (1.0f - alpha);
          const bool stop = hit && test_T < 0.0001f;
          const bool blend = hit && !stop;
          dbias[s] = stop ? 1e30f : dbias[s];
          const float w = blend ? alpha * T[s] : 0.0f;
          C0[s] += g2.x * w;
          C1[s] += g2.y * w;
          C2[s] += g2.z * w;
          Dp[s] += g2.w * w;
          Wt[s] += w;
          if (CE > 0) {
#pragma unroll
            for (int c = 0; c < CE; c++) X[s][c] += s_x[k * CE + c] * w;
          }
          T[s] = blend ? test_T : T[s];
          last[s] = blend ? __float_as_uint(g1.z) : last[s];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // keep the next batch's LDS writes behind this batch's reads
  }

  const size_t plane = (size_t)a.H * a.W;
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    if (inside[s]) {
      const int p = pixid[s];
      a.final_T[p] = T[s];
      a.n_contrib[p] = last[s];
      a.out_color[p] = C0[s] + T[s] * bg0;
      a.out_color[plane + p] = C1[s] + T[s] * bg1;
      a.out_color[2 * plane + p] = C2[s] + T[s] * bg2;
      a.out_alpha[p] = Wt[s];  // CR/forward.cu:380
      a.out_depth[p] = Dp[s];
      if (CE > 0) {
#pragma unroll
        for (int c = 0; c < CE; c++) a.out_extra[(size_t)c * plane + p] = X[s][c] + T[s] * (c % 3 == 0 ? bg0 : (c % 3 == 1 ? bg1 : bg2));
      }
    }
  }
}

int launch_blend_forward(const BlendFwdArgs &a, const Options &opt, hipStream_t stream) {
  const unsigned tiles = (unsigned)(a.grid_x * a.grid_y);
  if (tiles == 0) return GSR_OK;
  const unsigned slots = tile_slots_max(a.grid_x, a.grid_y);
  if (a.CE != 0) {
    if (a.CE != CE_MAX || !a.extra || !a.out_extra) {
      set_error("fused feature blend: exactly %d extra channels with input and output arrays are required", CE_MAX);
      return GSR_EINVAL;
    }
    hipLaunchKernelGGL((blend_forward_kernel<1, CE_MAX>), dim3(slots), dim3(WAVE * 4), 0, stream, a);
    return GSR_OK;
  }
  switch (opt.blend_fwd_waves) {
    case 1: hipLaunchKernelGGL((blend_forward_kernel<4, 0>), dim3(tiles), dim3(WAVE), 0, stream, a); break;
    case 2: hipLaunchKernelGGL((blend_forward_kernel<2, 0>), dim3(tiles * 2), dim3(WAVE), 0, stream, a); break;
    default: hipLaunchKernelGGL((blend_forward_kernel<1, 0>), dim3(slots), dim3(WAVE * 4), 0, stream, a); break;
  }
  return GSR_OK;
}

}  // namespace gsr
