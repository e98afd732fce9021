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

#ifndef FWD_FULL_ROWS
#define FWD_FULL_ROWS false
#endif
#ifndef FWD_FEATURE_WPG
#define FWD_FEATURE_WPG 2
#endif

namespace gsr {

// bijective XCD-aware remap: consecutive work items (which share Gaussians) land on the same XCD / L2
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t n) {
  const uint32_t q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
  const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

// CE > 0: fused multi-feature blend -- CE extra colour channels (a.extra[P][CE]) are composited with the same weights in
// the same pass (the reference rasterises seven times per frame for them, gaussian_renderer/__init__.py:203-272).
// waves per workgroup: the quadrant waves of a tile that are launched together (SLOTS == 1)
template <int SLOTS, int CE>
constexpr int fwd_wpg() { return SLOTS == 1 ? (CE > 0 ? FWD_FEATURE_WPG : 4) : 1; }

template <int SLOTS, int CE>
__global__ __launch_bounds__((WAVE * fwd_wpg<SLOTS, CE>())) void blend_forward_kernel(const BlendFwdArgs a) {
  // SLOTS == 1: the four quadrant waves of a tile form ONE workgroup -- still independent of each other, there is no workgroup
  // barrier anywhere -- so that they run on one CU and fetch the tile's records through one L1: 104 -> 97 us at C3, 153 -> 146 us
  // in the render() frame (the same layout changed nothing for the backward kernels, which keep one wave per workgroup).
  // With 18 extra channels a workgroup is HALF a tile (two waves, 15 KB of LDS): four-wave workgroups of 30.7 KB were only placed
  // four to a CU although five fit on paper.
  constexpr int WPG = fwd_wpg<SLOTS, CE>();
  constexpr uint32_t H = SLOTS == 1 ? 4 / WPG : 1;  // workgroups per tile
  const uint32_t wv = threadIdx.x / WAVE;
  constexpr int WPT = 4 / SLOTS;  // waves per tile
  // survivors' extra channels: rows of XS2 float2, padded to whole 16-byte reads (18 channels: 80 bytes, five ds_read_b128 instead of
  // nine ds_read_b64 per contributing survivor); with two-wave workgroups the 8 bytes per row cost no resident wave
  constexpr int XS2 = CE > 0 ? (CE + 3) / 4 * 2 : 1;
  __shared__ __attribute__((aligned(16))) float2 s_x_all[CE > 0 ? WPG * WAVE * XS2 : 1];
  __shared__ float4 s0_all[WPG * WAVE];     // x, y, qa, qb      (qa = -conic_a log2(e)/2, qb = -conic_b log2(e))
  // what the cut-off test needs sits in s0 + the first half of s1, what only a blending survivor needs in the second half of s1 +
  // s2: each branch's LDS reads are whole 8- / 16-byte accesses (a 4-byte broadcast read costs as many LDS cycles as an 8-byte one,
  // and the kernel runs the LDS at ~2/3 of its cycles)
  __shared__ float4 s1_all[WPG * WAVE];     // qc, log2(255*opacity) | list position + 1 (bits), opacity   (qc = -conic_c log2(e)/2)
  __shared__ float4 s2_all[WPG * WAVE];     // r, g, b, depth
  float2 *s_x = s_x_all + (CE > 0 ? wv * WAVE * XS2 : 0);
  float4 *s0 = s0_all + wv * WAVE, *s1 = s1_all + wv * WAVE, *s2 = s2_all + wv * WAVE;

  const unsigned long long trace_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  uint32_t tile, part, nseg = 1;
  if constexpr (SLOTS == 1) {
    // (the launch covers tile_slots_max() * H workgroups; the frame's own mode word says how many visiting slots it has)
    const int omode = tile_order_mode(a.order);
    const uint32_t n_slots = tile_slots_of(a.order, a.grid_x, a.grid_y, omode);
    // workgroup -> (visiting slot, part of the tile): consecutive slots on different XCDs (workgroup % 8), the H parts of a slot on one
    uint32_t slot, sub;
    if (H == 1) {
      slot = blockIdx.x, sub = 0u;
    } else {
      const uint32_t full = (n_slots / 8u) * 8u * H;
      slot = blockIdx.x < full ? (blockIdx.x / (8u * H)) * 8u + blockIdx.x % 8u : blockIdx.x / H;
      sub = blockIdx.x < full ? (blockIdx.x / 8u) % H : blockIdx.x % H;
    }
    if (!omode) slot = slot < n_slots ? xcd_remap(slot, n_slots) : n_slots;
    const uint32_t entry = tile_of_slot(a.order, omode, slot, n_slots);
    // (workgroup-uniform) the forward walks a list whole -- the early exit decides where it ends -- from the slot of its first segment
    if (entry == ORDER_NO_TILE || order_entry_seg(entry) != 0u) return;
    tile = order_entry_tile(entry), nseg = order_entry_nseg(entry);
    part = sub * WPG + wv;
  } else {
    const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
    tile = item / WPT, part = item % WPT;
  }
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const uint32_t lane = threadIdx.x % WAVE;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);
  list_priority(a.order, n, a.list_prio);

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
  // checkpoints for the backward's list segments (gsr_common.h "list segments"): the accumulators as they stand in front of entry
  // k * seg_len, k = 1 .. nseg - 1, and once more at the end (record nseg - 1); a wave that retires early leaves its final state in
  // every record it has not reached
  const int seg_len = nseg > 1u ? segment_len(n, (int)nseg) : 0;
  uint32_t next_rec = 0;
  auto checkpoint = [&](uint32_t k) {
    if constexpr (SLOTS == 1) {
      float *r = a.ckpt + (size_t)(a.ckpt_base[tile] + k) * (CKPT_PLANES * 256) + part * 64u + lane;
      r[0] = T[0];
      r[256] = C0[0];
      r[512] = C1[0];
      r[768] = C2[0];
      r[1024] = Dp[0];
      r[1280] = Wt[0];
      if (CE > 0) {
#pragma unroll
        for (int c = 0; c < CE; c++) r[(6 + c) * 256] = X[0][c];
      }
    }
  };
  for (int base = 0; base < n; base += WAVE) {
    if (nseg > 1u && next_rec + 1u < nseg && base == (int)(next_rec + 1u) * seg_len) checkpoint(next_rec++);  // (wave-uniform)
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
    float l255 = 0.f;
    if (keep) {  // second, exact filter: ellipse {alpha >= 1/255} against the wave's pixel rectangle (hardware reciprocals, the
                 // logarithm shared with the survivor row: gsr_common.h ellipse_hits_rect_fast)
      r1 = r1c;
      l255 = __builtin_amdgcn_logf(255.0f * r1.y);
      keep = ellipse_hits_rect_fast(r0.x, r0.y, r0.z, r0.w, r1.x, l255, rx0, rx1, ry0, ry1);
    }
    const uint64_t kmask = __ballot(keep);
    const int cnt = __builtin_popcountll(kmask);
    if (keep) {
      const int slot = __builtin_popcountll(kmask & lt);
      // exponent in base 2: p2 = power * log2(e) = dx (qa dx + qb dy) + qc dy dy
      constexpr float L2E = 1.4426950408889634f;
      s0[slot] = make_float4(r0.x, r0.y, (-0.5f * L2E) * r0.z, -L2E * r0.w);
      s1[slot] = make_float4((-0.5f * L2E) * r1.x, l255, __uint_as_float((uint32_t)(idx + 1)), r1.y);
      s2[slot] = make_float4(r1.w, r2.x, r2.y, r1.z);
      if (CE > 0) {
        const float2 *xs = reinterpret_cast<const float2 *>(a.extra + (size_t)id * CE);
#pragma unroll
        for (int q = 0; q < CE / 2; q++) s_x[slot * XS2 + q] = xs[q];
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

    // ---- blend the survivors.  Software pipeline: the LDS rows of survivor k + 1 are requested BEFORE the arithmetic of survivor k.
    // A wave's walk is one dependent chain (LDS read -> cut-off test -> branch -> LDS read -> blend), ~500 cycles per survivor when
    // nothing overlaps, and in a frame of unequal lists the kernel lasts as long as the wave with the longest list however many
    // other waves fill the gaps (tools/tile_cost_census.py): the chain, not the issue rate, is the time.
    // (two register sets, A and B, used in turn -- an explicit 2x unroll: a rotating single set costs ~30 v_mov per survivor)
    // FWD_FULL_ROWS: whole rows travel ahead (no exposed LDS round trip; 32 registers per set with 18 channels).  Otherwise only the
    // cut-off rows do, and a contributing survivor fetches its colour + channel rows in one round trip (16 registers less per set).
    constexpr bool FULL = FWD_FULL_ROWS;
    struct Row {
      float4 g0, g1, g2;
      float4 x[XS2 > 1 ? XS2 / 2 : 1];
      int kk;
    };
    auto fetch_rest = [&](Row &r) {
      r.g2 = s2[r.kk];
      if (CE > 0) {
        const float4 *xr = reinterpret_cast<const float4 *>(&s_x[r.kk * XS2]);
#pragma unroll
        for (int c = 0; c < XS2 / 2; c++) r.x[c] = xr[c];
      }
    };
    auto fetch = [&](Row &r, int k) {  // (a row beyond the last survivor is stale LDS: read, never used)
      r.kk = min(k, WAVE - 1);
      r.g0 = s0[r.kk];
      r.g1 = s1[r.kk];
      if constexpr (FULL) fetch_rest(r);
    };
    auto blend_one = [&](Row &r) {
#pragma unroll
      for (int s = 0; s < SLOTS; s++) {
        const float dx = r.g0.x - pxf[s], dy = r.g0.y - pyf[s];
        const float p2 = dx * (r.g0.z * dx + r.g0.w * dy) + (r.g1.x * dy) * dy;  // power * log2(e)
        // cheap necessary condition for alpha >= 1/255:  log2(255*o) + power*log2(e) >= 0  (0.02 safety margin)
        const bool pre = !(p2 > 0.0f) && ((p2 + r.g1.y) >= dbias[s]);
        if (__ballot(pre) != 0ull) {
          if constexpr (!FULL) fetch_rest(r);
          const float alpha = fminf(0.99f, r.g1.w * __builtin_amdgcn_exp2f(p2));
          const bool hit = pre && !(alpha < 1.0f / 255.0f);
          const float test_T = T[s] * (1.0f - alpha);
          const bool stop = hit && test_T < 0.0001f;
          const bool blend = hit && !stop;
          dbias[s] = stop ? 1e30f : dbias[s];
          const float w = blend ? alpha * T[s] : 0.0f;
          C0[s] += r.g2.x * w;
          C1[s] += r.g2.y * w;
          C2[s] += r.g2.z * w;
          Dp[s] += r.g2.w * w;
          Wt[s] += w;
          if (CE > 0) {
#pragma unroll
            for (int c = 0; c < CE; c++) {
              const float4 v = r.x[c / 4];
              X[s][c] += (c % 4 == 0 ? v.x : (c % 4 == 1 ? v.y : (c % 4 == 2 ? v.z : v.w))) * w;
            }
          }
          T[s] = blend ? test_T : T[s];
          last[s] = blend ? __float_as_uint(r.g1.z) : last[s];
        }
      }
    };
    Row A, B;
    fetch(A, 0);
    int k = 0;
    for (; k + 1 < cnt; k += 2) {
      fetch(B, k + 1);
      blend_one(A);
      fetch(A, k + 2);
      blend_one(B);
    }
    if (k < cnt) blend_one(A);
    __builtin_amdgcn_wave_barrier();  // keep the next batch's LDS writes behind this batch's reads
  }
  if (nseg > 1u)
    while (next_rec < nseg) checkpoint(next_rec++);
  if (a.trace && lane == 0) {
    unsigned long long *r = a.trace + ((size_t)blockIdx.x * WPG + wv) * 4u;
    r[0] = trace_t0;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = (unsigned long long)n;
    r[3] = (unsigned long long)tile;
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

#ifdef GSR_BUILD_EXPERIMENTS  // kernels that were built, measured and NOT adopted; python -m mygauhuman_amd.build --experiments
// ---------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (knob "blend_fwd_dma" = 1, default off: parity-green and slower -- 191 vs 144 us in the render() frame, see
// profiles/r3_fwd_dma_experiment.txt): the fused multi-feature forward (CE_MAX extra channels), software-pipelined with
// global -> LDS DMA.
//
// In the render() frame blend_forward_kernel<1, 18> ran its vector ALUs only 51 % of the time with the waves in s_waitcnt half
// of it (profiles/r2f_render_kernels.txt): every batch of 64 list entries ends with a gather of the survivors' 72-byte feature
// rows that the blend of THAT batch has to wait for, and the registers of the 18 extra accumulators leave three waves per
// SIMD to hide it.  Fetching the rows a batch ahead through registers costs a wave per SIMD (measured in round 2: slower).
// gfx950 can load global memory straight into LDS (global_load_lds_dwordx4 / _dword: lane i's data lands at M0 + 16 i / 4 i,
// masked-off lanes write nothing, an 8-byte-aligned source is fine -- tools/ubench/lds_dma_probe.hip), so here
//   * the cull / compaction of batch b + 1 runs BEFORE the blend of batch b (the records of b + 1 were requested two batches
//     ago and sit in registers) and the surviving lanes send their feature rows to the other half of a double-buffered LDS
//     area -- indexed by LANE, since the DMA decides the address; a survivor's lane is kept next to its record;
//   * the blend of batch b then runs with the rows of b + 1, the records of b + 2 and the list entries of b + 3 in flight,
//     and one s_waitcnt vmcnt(0) at the top of the next iteration collects what has had a whole blend phase to arrive.
// The DMA is issued through inline assembly: the compiler does not model the LDS write of the builtin (it would drop the
// reads) and would otherwise fence every LDS read with vmcnt(0).  No wave ends with a DMA in flight (the fence precedes the
// early exit; the last batch issues none).  Same arithmetic, same order as blend_forward_kernel<1, CE_MAX>: images keep their bits.
__device__ __forceinline__ void lds_dma16(const void *g, uint32_t lds_base) {
  const uint32_t b = __builtin_amdgcn_readfirstlane(lds_base);  // (wave-uniform by construction; pins it to an SGPR)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(b) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void *g, uint32_t lds_base) {
  const uint32_t b = __builtin_amdgcn_readfirstlane(lds_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(b) : "memory");
}

constexpr int FX_PLANE16 = WAVE * 16;                    // bytes of one 16-byte-per-lane plane
constexpr int FX_BYTES = 4 * FX_PLANE16 + 2 * WAVE * 4;  // floats 0..15 of a lane's row in four planes, floats 16, 17 in two dword planes
constexpr int HALF = WAVE / 2;

// Pipeline unit = HALF a register batch (32 list entries): the records of 64 entries sit in registers (lane <-> entry), the lower
// 32 lanes are staged, blended, then the upper 32 -- so that the LDS a wave needs stays what blend_forward_kernel<1, 18> needs
// (7.9 KB: five waves per SIMD; the first version staged whole batches into double-buffered 64-cell areas, 15.9 KB per wave = two
// waves per SIMD, and ran 260 instead of 145 us: occupancy hides more latency here than the pipelining).  The DMA cell of a
// survivor is its LANE, and the two halves of a batch use the two halves of the 64 cells: the double buffer costs no extra LDS.
__global__ __launch_bounds__(WAVE * 4) void blend_forward_features_kernel(const BlendFwdArgs a) {
  constexpr int CE = CE_MAX;
  const uint32_t wv = threadIdx.x / WAVE;
  __shared__ __attribute__((aligned(16))) char s_fx_all[4 * FX_BYTES];       // [wave]: lanes 0..31 = even half-steps, 32..63 = odd
  __shared__ float4 s0_all[4 * WAVE], s1_all[4 * WAVE], s2_all[4 * WAVE];    // [wave][half-step parity][32 slots]
  __shared__ uint32_t s_lane_all[4 * WAVE];
  char *s_fx = s_fx_all + wv * FX_BYTES;
  float4 *s0w = s0_all + wv * WAVE, *s1w = s1_all + wv * WAVE, *s2w = s2_all + wv * WAVE;
  uint32_t *s_lanew = s_lane_all + wv * WAVE;
  const uint32_t fx_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_fx);   // LDS byte address of this wave's area

  const int omode = tile_order_mode(a.order);
  const uint32_t n_slots = tile_slots_of(a.order, a.grid_x, a.grid_y, omode);
  const uint32_t slot_id = omode ? blockIdx.x : (blockIdx.x < n_slots ? xcd_remap(blockIdx.x, n_slots) : n_slots);
  const uint32_t entry = tile_of_slot(a.order, omode, slot_id, n_slots), part = wv;
  if (entry == ORDER_NO_TILE || order_entry_seg(entry) != 0u) return;  // (workgroup-uniform; frames of this variant are binned without segments)
  const uint32_t tile = order_entry_tile(entry);
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const uint32_t lane = threadIdx.x % WAVE;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);
  list_priority(a.order, n, a.list_prio);

  const int px = tx * TILE + (int)(part & 1) * 8 + (int)(lane & 7);
  const int py = ty * TILE + (int)(part >> 1) * 8 + (int)(lane >> 3);
  const float pxf = (float)px, pyf = (float)py;
  const bool inside = px < a.W && py < a.H;
  float dbias = inside ? -0.02f : 1e30f;   // threshold of the alpha test: -0.02 while live, +1e30 once done
  const int pixid = py * a.W + px;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
  float X[CE];
#pragma unroll
  for (int c = 0; c < CE; c++) X[c] = 0.f;
  uint32_t last = 0;
  const float rx0 = (float)(tx * TILE + (int)(part & 1) * 8), rx1 = rx0 + 7.f;
  const float ry0 = (float)(ty * TILE + (int)(part >> 1) * 8), ry1 = ry0 + 7.f;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));

  // records of the register batch that is being staged (lane <-> list entry), its ids, the ids of the batch after it
  float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
  uint32_t id_cur = 0, id_a = 0;
  if ((int)lane < n) {
    id_cur = a.point_list[range.x + lane];
    const float4 *src = reinterpret_cast<const float4 *>(a.recs + id_cur);
    p0 = src[0];
    p1 = src[1];
    p2 = src[2];
  }
  if ((int)lane + WAVE < n) id_a = a.point_list[range.x + lane + WAVE];

  // stage half-step j (list entries 32 j .. 32 j + 31 = lanes 32 (j & 1) .. + 31 of the register batch): cull, compact the
  // survivors' records into parity j & 1, send their feature rows on their way (cell = lane); after the second half of a batch
  // request the records of the next batch.  Returns the number of survivors.
  auto stage = [&](int j) -> int {
    const int hp = j & 1;
    const int idx = (j >> 1) * WAVE + (int)lane;   // list entry this lane's registers hold
    bool keep = false;
    if ((int)(lane >> 5) == hp && idx < n)
      keep = (p0.x + p2.z >= rx0) && (p0.x - p2.z <= rx1) && (p0.y + p2.w >= ry0) && (p0.y - p2.w <= ry1);
    if (keep) keep = ellipse_hits_rect(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, rx0, rx1, ry0, ry1);
    const uint64_t kmask = __ballot(keep);
    if (keep) {
      const int slot = hp * HALF + __builtin_popcountll(kmask & lt);
      constexpr float L2E = 1.4426950408889634f;
      s0w[slot] = make_float4(p0.x, p0.y, (-0.5f * L2E) * p0.z, -L2E * p0.w);
      s1w[slot] = make_float4((-0.5f * L2E) * p1.x, __builtin_amdgcn_logf(255.0f * p1.y), __uint_as_float((uint32_t)(idx + 1)), p1.y);
      s2w[slot] = make_float4(p1.w, p2.x, p2.y, p1.z);
      s_lanew[slot] = lane;
      const float *row = a.extra + (size_t)id_cur * CE;
#pragma unroll
      for (int q = 0; q < 4; q++) lds_dma16(row + 4 * q, fx_base + (uint32_t)(q * FX_PLANE16));
      lds_dma4(row + 16, fx_base + (uint32_t)(4 * FX_PLANE16));
      lds_dma4(row + 17, fx_base + (uint32_t)(4 * FX_PLANE16 + WAVE * 4));
    }
    if (hp) {  // (uniform) the register batch is used up: the next batch's records and the list entries of the one after it
      id_cur = id_a;
      if (idx + WAVE < n) {
        const float4 *src = reinterpret_cast<const float4 *>(a.recs + id_a);
        p0 = src[0];
        p1 = src[1];
        p2 = src[2];
      }
      if (idx + 2 * WAVE < n) id_a = a.point_list[range.x + idx + 2 * WAVE];
    }
    return __builtin_popcountll(kmask);
  };

  int cnt = n > 0 ? stage(0) : 0;
  const float4 *fx16 = reinterpret_cast<const float4 *>(s_fx);
  const float *fx4 = reinterpret_cast<const float *>(s_fx + 4 * FX_PLANE16);
  for (int j = 0; j * HALF < n; j++) {
    // everything requested one blend phase ago: the feature rows of this half, the records of the next batch, list entries
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__ballot(!(dbias > 0.f)) == 0ull) break;   // every pixel of the wave is saturated (nothing is in flight here)
    const int cnt_next = (j + 1) * HALF < n ? stage(j + 1) : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int sb = (j & 1) * HALF;
    for (int k = 0; k < cnt; k++) {
      const float4 g0 = s0w[sb + k];
      const float4 g1 = s1w[sb + k];
      const float dx = g0.x - pxf, dy = g0.y - pyf;
      const float pw = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;  // power * log2(e)
      const bool pre = !(pw > 0.0f) && ((pw + g1.y) >= dbias);
      if (__ballot(pre) != 0ull) {
        const float4 g2 = s2w[sb + k];
        const uint32_t L = s_lanew[sb + k];
        const float alpha = fminf(0.99f, g1.w * __builtin_amdgcn_exp2f(pw));
        const bool hit = pre && !(alpha < 1.0f / 255.0f);
        const float test_T = T * (1.0f - alpha);
        const bool stop = hit && test_T < 0.0001f;
        const bool blend = hit && !stop;
        dbias = stop ? 1e30f : dbias;
        const float w = blend ? alpha * T : 0.0f;
        C0 += g2.x * w;
        C1 += g2.y * w;
        C2 += g2.z * w;
        Dp += g2.w * w;
        Wt += w;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 f = fx16[q * WAVE + L];
          X[4 * q + 0] += f.x * w;
          X[4 * q + 1] += f.y * w;
          X[4 * q + 2] += f.z * w;
          X[4 * q + 3] += f.w * w;
        }
        X[16] += fx4[L] * w;
        X[17] += fx4[WAVE + L] * w;
        T = blend ? test_T : T;
        last = blend ? __float_as_uint(g1.z) : last;
      }
    }
    cnt = cnt_next;
    __builtin_amdgcn_wave_barrier();
  }

  const size_t plane = (size_t)a.H * a.W;
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
  if (inside) {
    a.final_T[pixid] = T;
    a.n_contrib[pixid] = last;
    a.out_color[pixid] = C0 + T * bg0;
    a.out_color[plane + pixid] = C1 + T * bg1;
    a.out_color[2 * plane + pixid] = C2 + T * bg2;
    a.out_alpha[pixid] = Wt;
    a.out_depth[pixid] = Dp;
#pragma unroll
    for (int c = 0; c < CE; c++) a.out_extra[(size_t)c * plane + pixid] = X[c] + T * (c % 3 == 0 ? bg0 : (c % 3 == 1 ? bg1 : bg2));
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Survivor-parallel layout (knob "blend_layout" = 1): a wave owns a 4x4 pixel block and blends FOUR list entries per step.
//
//   lane = 16 g + p:  p = pixel of the block (x = p & 3, y = p >> 2),  g = which of the step's four consecutive survivors.
//
// Why: a quadrant wave (above) walks its list one survivor at a time -- one dependent chain of ~75 instruction slots per survivor --
// and a lone wave issues at most one instruction every four cycles.  In a frame of unequal lists (a body: 1,480 of 4,096 tiles busy,
// mean list 469, longest 1,398) the kernel then lasts as long as the waves of the longest tiles take BY THEMSELVES: launched with
// nothing but the tiles within 25 % of the longest list, the quadrant kernel still needs 90 of its 138 us (forward) and 214 of 283 us
// (backward) (profiles/r3d_lone_wave.txt).  Here the same instruction stream advances four entries for a quarter of the pixels:
// the chain per list entry is four times shorter, and the cull against a 4x4 block (instead of 8x8) removes a third of the
// (pixel, Gaussian) pairs on top.
//
// The reference's sequential rules are kept bit for bit where they decide something: the transmittances in front of the step's four
// survivors are the sequential products t1 = T f0, t2 = t1 f1, ... (f = 1 - alpha of a hit, 1 otherwise; every lane forms all four
// from the other rows' f, fetched with three permlane swaps), a pixel stops at the first survivor whose product falls below 1e-4
// (products only shrink, so "stopped before or at me" is "my product is below 1e-4"), T stays what it was in front of that one.
// Only the colour sums differ from the reference in association: each row adds up its own survivors, the four rows meet at the end.
typedef unsigned int fwd_u32x2 __attribute__((ext_vector_type(2)));
// v of the four 16-lane rows -> (row 0's, row 1's, row 2's, row 3's) value in every lane of the same column
__device__ __forceinline__ void gather_rows(float v, float &r0, float &r1, float &r2, float &r3) {
  const fwd_u32x2 h = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);  // [v0 v1 v0 v1], [v2 v3 v2 v3]
  const fwd_u32x2 lo = __builtin_amdgcn_permlane16_swap(h.x, h.x, false, false);                               // [v0 x4], [v1 x4]
  const fwd_u32x2 hi = __builtin_amdgcn_permlane16_swap(h.y, h.y, false, false);                               // [v2 x4], [v3 x4]
  r0 = __uint_as_float(lo.x);
  r1 = __uint_as_float(lo.y);
  r2 = __uint_as_float(hi.x);
  r3 = __uint_as_float(hi.y);
}
// sum over the four rows, in every lane
__device__ __forceinline__ float sum_rows(float v) {
  const fwd_u32x2 h = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(h.x) + __uint_as_float(h.y);  // rows [v0+v2, v1+v3, v0+v2, v1+v3]
  const fwd_u32x2 q = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(q.x) + __uint_as_float(q.y);
}
__device__ __forceinline__ uint32_t max_rows(uint32_t v) {
  const fwd_u32x2 h = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  const uint32_t s = max(h.x, h.y);
  const fwd_u32x2 q = __builtin_amdgcn_permlane16_swap(s, s, false, false);
  return max(q.x, q.y);
}

template <int CE>
__global__ __launch_bounds__(WAVE * 4) void blend_forward_sp_kernel(const BlendFwdArgs a) {
  constexpr int XS = (CE + 3) / 4 * 4;  // floats per survivor row of extra channels (whole 16-byte reads)
  __shared__ float4 s_x_all[CE > 0 ? 4 * WAVE * XS / 4 : 1];
  __shared__ float4 s0_all[4 * WAVE];  // x, y, qa, qb                        (as blend_forward_kernel)
  __shared__ float4 s1_all[4 * WAVE];  // qc, log2(255 opacity) | list position + 1 (bits), opacity
  __shared__ float4 s2_all[4 * WAVE];  // r, g, b, depth
  const uint32_t wv = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  float4 *s_x = s_x_all + (CE > 0 ? wv * WAVE * XS / 4 : 0);
  float4 *s0 = s0_all + wv * WAVE, *s1 = s1_all + wv * WAVE, *s2 = s2_all + wv * WAVE;

  // workgroup = the four 4x4 blocks of one quadrant of a tile; (slot, quadrant) as in the backward kernels
  const int omode = tile_order_mode(a.order);
  const uint32_t n_slots = tile_slots_of(a.order, a.grid_x, a.grid_y, omode);
  const uint32_t item = omode ? ordered_item4(blockIdx.x, n_slots) : (blockIdx.x < n_slots * 4u ? xcd_remap(blockIdx.x, n_slots * 4u) : n_slots * 4u);
  const uint32_t entry = tile_of_slot(a.order, omode, item / 4u, n_slots), quad = item % 4u;
  if (entry == ORDER_NO_TILE || order_entry_seg(entry) != 0u) return;  // (workgroup-uniform; frames of this variant are binned without segments)
  const uint32_t tile = order_entry_tile(entry);
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);
  list_priority(a.order, n, a.list_prio);

  // (rows beyond a batch's last survivor are read and multiplied by a zero weight: they must hold finite numbers from the start)
  s2[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (CE > 0) {
#pragma unroll
    for (int c = 0; c < XS / 4; c++) s_x[lane * (XS / 4) + c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int g = (int)(lane >> 4), p = (int)(lane & 15u);
  const int x0 = tx * TILE + (int)(quad & 1u) * 8 + (int)(wv & 1u) * 4, y0 = ty * TILE + (int)(quad >> 1) * 8 + (int)(wv >> 1) * 4;
  const int px = x0 + (p & 3), py = y0 + (p >> 2);
  const float pxf = (float)px, pyf = (float)py;
  const float rx0 = (float)x0, rx1 = (float)(x0 + 3), ry0 = (float)y0, ry1 = (float)(y0 + 3);
  const bool inside = px < a.W && py < a.H;
  const int pixid = py * a.W + px;
  float dbias = inside ? -0.02f : 1e30f;  // threshold of the alpha test: -0.02 while the pixel is live, +1e30 once it is done
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
  float X[CE > 0 ? CE : 1];
#pragma unroll
  for (int c = 0; c < (CE > 0 ? CE : 1); c++) X[c] = 0.f;
  uint32_t last = 0;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));

  // list walk exactly as blend_forward_kernel: records of the next batch and list entries of the one after it are in flight
  uint32_t id_a = 0, id_cur = 0;
  float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
  if ((int)lane < n) {
    id_cur = a.point_list[range.x + lane];
    const float4 *src = reinterpret_cast<const float4 *>(a.recs + id_cur);
    p0 = src[0];
    p1 = src[1];
    p2 = src[2];
  }
  if ((int)lane + WAVE < n) id_a = a.point_list[range.x + lane + WAVE];
  for (int base = 0; base < n; base += WAVE) {
    if (__ballot(!(dbias > 0.f)) == 0ull) break;
    const int idx = base + (int)lane;
    const float4 r0 = p0, r1c = p1, r2 = p2;
    const uint32_t id = id_cur;
    bool keep = false;
    if (idx < n) keep = (r0.x + r2.z >= rx0) && (r0.x - r2.z <= rx1) && (r0.y + r2.w >= ry0) && (r0.y - r2.w <= ry1);
    float4 r1 = make_float4(0, 0, 0, 0);
    if (keep) {
      r1 = r1c;
      keep = ellipse_hits_rect(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, rx0, rx1, ry0, ry1);
    }
    const uint64_t kmask = __ballot(keep);
    const int cnt = __builtin_popcountll(kmask);
    if (keep) {
      const int slot = __builtin_popcountll(kmask & lt);
      constexpr float L2E = 1.4426950408889634f;
      s0[slot] = make_float4(r0.x, r0.y, (-0.5f * L2E) * r0.z, -L2E * r0.w);
      s1[slot] = make_float4((-0.5f * L2E) * r1.x, __builtin_amdgcn_logf(255.0f * r1.y), __uint_as_float((uint32_t)(idx + 1)), r1.y);
      s2[slot] = make_float4(r1.w, r2.x, r2.y, r1.z);
      if (CE > 0) {
        const float2 *xs = reinterpret_cast<const float2 *>(a.extra + (size_t)id * CE);
#pragma unroll
        for (int q = 0; q < CE / 2; q++) reinterpret_cast<float2 *>(&s_x[slot * (XS / 4)])[q] = xs[q];
      }
    }
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

    // ---- blend: four survivors per step (row g of the wave takes survivor j + g); the rows of the NEXT step are requested
    // before this step's arithmetic (two register sets used in turn)
    struct Row {
      float4 g0, g1, g2;
      float4 x[CE > 0 ? XS / 4 : 1];
    };
    auto fetch = [&](Row &r, int j) {  // (rows beyond the last survivor are stale LDS: read, masked by `valid`)
      const int rr = min(j + g, WAVE - 1);
      r.g0 = s0[rr];
      r.g1 = s1[rr];
      r.g2 = s2[rr];
      if (CE > 0) {
#pragma unroll
        for (int c = 0; c < XS / 4; c++) r.x[c] = s_x[rr * (XS / 4) + c];
      }
    };
    auto step = [&](const Row &r, int j) {
      const bool valid = j + g < cnt;
      const float dx = r.g0.x - pxf, dy = r.g0.y - pyf;
      const float pw = dx * (r.g0.z * dx + r.g0.w * dy) + (r.g1.x * dy) * dy;  // power * log2(e)
      const bool pre = valid && !(pw > 0.0f) && ((pw + r.g1.y) >= dbias);
      if (__ballot(pre) != 0ull) {
        const float alpha = fminf(0.99f, r.g1.w * __builtin_amdgcn_exp2f(pw));
        const bool hit = pre && !(alpha < 1.0f / 255.0f);
        const float f = hit ? 1.0f - alpha : 1.0f;
        float f0, f1, f2, f3;
        gather_rows(f, f0, f1, f2, f3);
        const float t1 = T * f0, t2 = t1 * f1, t3 = t2 * f2, t4 = t3 * f3;  // the reference's running T after each of the four
        const float tb = g == 0 ? T : (g == 1 ? t1 : (g == 2 ? t2 : t3));  // T in front of this row's survivor
        const float tn = tb * f;                                             // (bit-identical to t1 / t2 / t3 / t4 of this row)
        const bool blend = hit && !(tn < 0.0001f);
        const float w = blend ? alpha * tb : 0.0f;
        C0 += r.g2.x * w;
        C1 += r.g2.y * w;
        C2 += r.g2.z * w;
        Dp += r.g2.w * w;
        Wt += w;
        if (CE > 0) {
#pragma unroll
          for (int c = 0; c < CE; c++) {
            const float4 v = r.x[c / 4];
            X[c] += (c % 4 == 0 ? v.x : (c % 4 == 1 ? v.y : (c % 4 == 2 ? v.z : v.w))) * w;
          }
        }
        last = blend ? __float_as_uint(r.g1.z) : last;
        // the pixel's T after the step: the last product still >= 1e-4 (a stop leaves T in front of the stopping survivor)
        float Tn = (t4 < 0.0001f) ? t3 : t4;
        Tn = (t3 < 0.0001f) ? t2 : Tn;
        Tn = (t2 < 0.0001f) ? t1 : Tn;
        Tn = (t1 < 0.0001f) ? T : Tn;
        dbias = (t4 < 0.0001f) ? 1e30f : dbias;
        T = Tn;
      }
    };
    Row A, B;
    fetch(A, 0);
    int j = 0;
    for (; j + 4 < cnt; j += 8) {
      fetch(B, j + 4);
      step(A, j);
      fetch(A, j + 8);
      step(B, j + 4);
    }
    if (j < cnt) step(A, j);
    __builtin_amdgcn_wave_barrier();  // keep the next batch's LDS writes behind this batch's reads
  }

  // ---- the four rows' partial sums meet; row g writes the planes c = g (mod 4)
  const size_t plane = (size_t)a.H * a.W;
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
  C0 = sum_rows(C0);
  C1 = sum_rows(C1);
  C2 = sum_rows(C2);
  Dp = sum_rows(Dp);
  Wt = sum_rows(Wt);
  last = max_rows(last);
#pragma unroll
  for (int c = 0; c < (CE > 0 ? CE : 0); c++) X[c] = sum_rows(X[c]);
  if (inside) {
    if (g == 0) {
      a.final_T[pixid] = T;
      a.n_contrib[pixid] = last;
      a.out_color[pixid] = C0 + T * bg0;
    } else if (g == 1) {
      a.out_color[plane + pixid] = C1 + T * bg1;
      a.out_alpha[pixid] = Wt;  // CR/forward.cu:380
    } else if (g == 2) {
      a.out_color[2 * plane + pixid] = C2 + T * bg2;
    } else {
      a.out_depth[pixid] = Dp;
    }
    if (CE > 0) {
#pragma unroll
      for (int c = 0; c < CE; c++)
        if ((c & 3) == g) a.out_extra[(size_t)c * plane + pixid] = X[c] + T * (c % 3 == 0 ? bg0 : (c % 3 == 1 ? bg1 : bg2));
    }
  }
}

#endif  // GSR_BUILD_EXPERIMENTS

int launch_blend_forward(const BlendFwdArgs &a, const Options &opt, hipStream_t stream) {
  const unsigned tiles = (unsigned)(a.grid_x * a.grid_y);
  if (tiles == 0) return GSR_OK;
  const unsigned slots = tile_slots_max(a.grid_x, a.grid_y);
  if (a.CE != 0) {
    if (a.CE != CE_MAX || !a.extra || !a.out_extra) {
      set_error("fused feature blend: exactly %d extra channels with input and output arrays are required", CE_MAX);
      return GSR_EINVAL;
    }
#ifdef GSR_BUILD_EXPERIMENTS
    if (opt.blend_layout == 1) {
      hipLaunchKernelGGL((blend_forward_sp_kernel<CE_MAX>), dim3(slots * 4), dim3(WAVE * 4), 0, stream, a);
      return GSR_OK;
    }
    if (opt.blend_fwd_dma) {
      hipLaunchKernelGGL(blend_forward_features_kernel, dim3(slots), dim3(WAVE * 4), 0, stream, a);
      return GSR_OK;
    }
#endif
    hipLaunchKernelGGL((blend_forward_kernel<1, CE_MAX>), dim3(slots * (4 / FWD_FEATURE_WPG)), dim3(WAVE * FWD_FEATURE_WPG), 0, stream, a);
    return GSR_OK;
  }
#ifdef GSR_BUILD_EXPERIMENTS
  if (opt.blend_layout == 1 && opt.blend_fwd_waves == 4) {
    hipLaunchKernelGGL((blend_forward_sp_kernel<0>), dim3(slots * 4), dim3(WAVE * 4), 0, stream, a);
    return GSR_OK;
  }
#endif
  switch (opt.blend_fwd_waves) {
    case 1: hipLaunchKernelGGL((blend_forward_kernel<4, 0>), dim3(tiles), dim3(WAVE), 0, stream, a); break;
    case 2: hipLaunchKernelGGL((blend_forward_kernel<2, 0>), dim3(tiles * 2), dim3(WAVE), 0, stream, a); break;
    default: hipLaunchKernelGGL((blend_forward_kernel<1, 0>), dim3(slots), dim3(WAVE * 4), 0, stream, a); break;
  }
  return GSR_OK;
}

}  // namespace gsr
