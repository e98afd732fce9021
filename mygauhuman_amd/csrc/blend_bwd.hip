// blend_bwd.hip -- per-pixel backward of the alpha blend (replaces renderCUDA, CR/backward.cu:399-587).
//
// Same decomposition as blend_fwd.hip: one independent wave per SLOTS quadrants of a tile, the tile's list is
// replayed back to front from each pixel's n_contrib / final_T, 64 entries at a time, culled against the wave's
// pixel rectangle (and against the wave's largest n_contrib) and compacted into LDS.
//
// Gradient accumulation, wave64-first.  The reference issues nine global atomicAdds per contributing
// (pixel, Gaussian) pair (CR/backward.cu:538,574-584).  Here survivors are processed four at a time and their
// 4 x 9 per-lane partial sums are reduced TOGETHER:
//   1. v_permlane32_swap on (a, b) and (c, d) + add   -> lanes 0-31 hold a, lanes 32-63 hold b  (resp. c, d)
//   2. v_permlane16_swap on (ab, cd) + add            -> the four 16-lane rows hold a, c, b, d
//   3. four DPP row steps (quad_perm x2, row_ror 4/8) -> every lane of a row holds that Gaussian's total
// = 10 instructions per value for FOUR Gaussians (2.5 per Gaussian-value instead of 6 for a plain wave reduction),
// and the result is already laid out for one atomic wave-instruction: lane (row r, column k < 9) adds value k of
// row r's Gaussian into that Gaussian's packed 64-byte gradient row (GROW floats: one memory-side request per
// Gaussian row instead of 9 scattered ones).  Ballots skip the exp / the reduction / the atomics whenever no lane
// is hit.
#include "gsr_common.h"

namespace gsr {

__device__ __forceinline__ uint32_t xcd_remap_b(uint32_t bid, uint32_t n) {
  const uint32_t q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
  const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// after this every lane holds the sum over its 16-lane DPP row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x124>(v);  // row_ror 4
  v += dpp_f<0x128>(v);  // row_ror 8
  return v;
}
// sum over the 8 lanes (lx = lane & 7) of each half row; both halves are reduced independently
__device__ __forceinline__ float half8_sum(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);  // row_half_mirror: lane l <-> 7 - l inside each half (the quads are uniform by now)
  return v;
}
// lower half row <- a summed over both halves, upper half row <- b summed over both halves (one DPP for two values)
__device__ __forceinline__ float pack_halves(float a, float b, bool upper) {
  const float keep = upper ? b : a, send = upper ? a : b;
  return keep + dpp_f<0x128>(send);  // row_ror 8
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// lanes 0-31: sum over the wave of a; lanes 32-63: sum over the wave of b (both still spread over 32 lanes)
__device__ __forceinline__ float fold32(float a, float b) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// x rows [x0 x1 x2 x3], y rows [y0 y1 y2 y3] -> rows [x0+x1, y0+y1, x2+x3, y2+y3]
__device__ __forceinline__ float fold16(float x, float y) {
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// the same folds, also handing back the part that came from the upper lanes / odd rows: for lane = 32 b2 + 16 b1 + 8 b0 + lx
// these are the b2 = 1 and b1 = 1 terms, from which the y-moments follow without accumulating r dy and r dy^2 per lane
__device__ __forceinline__ float fold32_parts(float a, float b, float &hi) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  hi = __uint_as_float(r.y);
  return __uint_as_float(r.x) + hi;
}
__device__ __forceinline__ float fold16_parts(float x, float y, float &odd) {
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  odd = __uint_as_float(r.y);
  return __uint_as_float(r.x) + odd;
}

// GROW row columns 0..8: with r = G * dL_dalpha(pair) and d = mean2D - pixel:
//   [0] sum r*dx  [1] sum r*dy  [2] sum r*dx*dx  [3] sum r*dx*dy  [4] sum r*dy*dy  [5] sum r (= dL_dopacity)
//   [6..8] dL_dcolor.  The geometric gradients are linear in these moments (CR/backward.cu:567-580):
//   dL_dmean2D.x = -o (A m0 + B m1) W/2, .y = -o (C m1 + B m0) H/2, dL_dconic = -o/2 (m2, m3, m4); see preprocess_bwd.hip.
constexpr int NACC = 9;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// RED selects how the four row-resident Gaussians are reduced over their 16 lanes:
//   0: four DPP row steps per value (quad_perm x2, row_ror 4/8)                       -- VALU only
//   1: two f32 MFMAs per value on the otherwise idle matrix pipe (exact f32, same sums):
//        D1 = F x Sel   (A = the folded register: A[c][r] = F[16 r + c];  Sel[r][j] = [r == j mod 4])
//           -> lane (j, g), register t holds F[16 (j mod 4) + 4 g + t]; the four registers are added (3 VALU adds)
//        D2 += P x Col_k (Col_k[g][j] = [j == k])  chained over the nine values
//           -> lane (column k, any row group), register t = total of row t's Gaussian for value k
//      which is exactly the (row, column) layout the gradient-row atomic wants.
// CE > 0 (SLOTS == 1 only): the backward of the fused multi-feature blend.  The extra channels enter e (one more dot
// product per hit) and get their own colour-gradient sums sum_pix w * dL_dpix[c]; those are formed from the four kept
// blending weights right before they are folded, so only 4 extra registers stay live during the group.  Gradient rows
// have GROWX = 32 columns then: 0..8 as usual, 9 .. 9+CE-1 the extra colour gradients (two atomic instructions).
// DET (SLOTS == 1, RED == 0, CE == 0): deterministic mode, see BlendBwdArgs::det_rows.
template <int SLOTS, int RED, int CE, bool DET = false>
__global__ __launch_bounds__(WAVE) void blend_backward_kernel(const BlendBwdArgs a) {
  static_assert(CE == 0 || SLOTS == 1, "feature channels are only built for one pixel per lane");
  static_assert(!DET || (SLOTS == 1 && RED == 0 && CE == 0), "deterministic mode is built for the default configuration");
  constexpr int WPT = 4 / SLOTS;
  constexpr int ROWF = CE > 0 ? GROWX : GROW;
  // SEP: separable reduction (one pixel per lane, lane = 8 ly + lx): per lane only sum r, sum r dy, sum r dy^2 and the three
  // colour sums are kept (6 values instead of 9); after the folds one row_ror-8 step turns them into COLUMN sums, the
  // x-weights (dx is a function of lx alone) are applied there, two values are packed into the two half rows, and three DPP
  // steps over 8 lanes finish: 18 permlane swaps + 19 DPP per four Gaussians instead of 27 + 36.
  // LFOLD (RED == 3, one pixel per lane, no feature channels): the first two fold levels go through LDS instead of
  // v_permlane32/16_swap.  A pixel lane leaves (r, w) of the group's four Gaussians in LDS (two 16-byte stores); lane
  // (row = Gaussian, c = lane & 15) reads back the four pixels c + 16 q (same column, rows 2 q + (c >> 3)) of ITS Gaussian and
  // forms the column-partial sums directly, with the y-weights as plain multiplies (ly is known per read) and the colour sums as
  // w . dL_dpix of those four pixels (12 per-lane constants).  13 swaps (~4.2 FMA slots each), the register copies the swaps
  // force and the hierarchical y-moment algebra are replaced by ~30 plain VALU instructions; the 8-lane tail is unchanged.
  constexpr bool LFOLD = (SLOTS == 1 && RED == 3 && CE == 0);
  constexpr bool SEP = (SLOTS == 1 && (RED == 0 || RED == 3));
  // HIER (with SEP): only sum r is kept per lane; with ly = 4 b2 + 2 b1 + b0 the folds over b2, b1, b0 also hand back their
  // "bit set" halves, so that  sum r ly = 4 E[b2] + 2 E[b1] + E[b0]  and  sum r ly^2 = 16 E[b2] + 4 E[b1] + E[b0] + 16 E[b2 b1]
  // + 8 E[b2 b0] + 4 E[b1 b0]  come out of 4 swaps instead of 9, and the moments about the Gaussian follow from
  // dy = Dy - ly (Dy = mean.y - top row of the quadrant).
  constexpr bool HIER = SEP && !LFOLD;
  constexpr int NA = LFOLD ? 2 : (SEP ? (HIER ? 4 : 6) : NACC);
  constexpr int LROW = 12;  // floats per pixel lane in the transposition buffer: 4 x (r, w) + 4 pad (conflict-free b64 reads)
  __shared__ __attribute__((aligned(16))) float s_rw[LFOLD ? WAVE * LROW : 4];
  // (LDS is handed out in 1,280-byte granules on this part -- 160 KB / 128: the plain kernel's 6,416 bytes took SIX of them, 21 waves per
  // CU; at <= 6,400 it takes five and the registers' six waves per SIMD fit.  Hence no spare words below.)
  __shared__ __attribute__((aligned(16))) float s_x[CE > 0 ? WAVE * CE : 1];
  __shared__ float4 s0[WAVE];     // x, y, qa, qb      (qa = -conic_a log2(e)/2, qb = -conic_b log2(e))
  // (layout as in blend_fwd.hip: the cut-off test reads s0 + s1, a contributing survivor additionally s2: whole 16-byte reads)
  __shared__ float4 s1[WAVE];     // qc, log2(255*opacity), front position (bits), opacity   (qc = -conic_c log2(e)/2)
  __shared__ float4 s2[WAVE];     // r, g, b, depth
  __shared__ uint32_t s_id[WAVE];
  __shared__ uint32_t s_slot[DET ? WAVE : 1];

  const unsigned long long trace_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  uint32_t tile, part, seg = 0, nseg = 1;
  if constexpr (WPT == 4) {
    const int omode = tile_order_mode(a.order);
    const uint32_t n_slots = tile_slots_of(a.order, a.grid_x, a.grid_y, omode);
    const uint32_t item = omode ? ordered_item4(blockIdx.x, n_slots) : (blockIdx.x < n_slots * 4u ? xcd_remap_b(blockIdx.x, n_slots * 4u) : n_slots * 4u);
    const uint32_t entry = tile_of_slot(a.order, omode, item / 4u, n_slots);
    if (entry == ORDER_NO_TILE) return;  // (wave-uniform) padding slot, or beyond this frame's slots
    tile = order_entry_tile(entry), seg = order_entry_seg(entry), nseg = order_entry_nseg(entry), part = item % 4u;
  } else {
    const uint32_t item = xcd_remap_b(blockIdx.x, gridDim.x);  // (fewer waves per tile: natural order)
    tile = item / WPT, part = item % WPT;
  }
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const uint32_t lane = threadIdx.x;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);
  // this wave's SEGMENT of the list, front positions [s_lo, s_hi): the whole list unless the frame cut it (gsr_common.h "list segments")
  const int seg_len = nseg > 1u ? segment_len(n, (int)nseg) : n;
  const int s_lo = (int)seg * seg_len, s_hi = min(n, s_lo + seg_len);
  if (seg != 0u && s_lo >= n) return;
  list_priority(a.order, s_hi - s_lo, a.list_prio);
  const size_t plane = (size_t)a.H * a.W;
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];

  // Per-pixel replay state.  The reference keeps T plus five "colour behind" recurrences (accum_rec[3], depth, alpha)
  // and their last_* operands (CR/backward.cu:454-468,529-548).  With e_j = c_j . dL_dpix + depth_j dL_ddepth + dL_dalpha
  // (one scalar per pair) they collapse into ONE suffix sum X = sum_{j behind i} w_j e_j:
  //   dL_dalpha_i = T_i e_i - (X_i + T_final (bg . dL_dpix)) / (1 - alpha_i),    w_i = alpha_i T_i,
  // which is the same value (accum_rec_i = X_i / (T_i (1 - alpha_i))) with 16 fewer instructions per hit and 10 fewer
  // live registers per pixel.
  float pxf[SLOTS], pyf[SLOTS], T[SLOTS], X[SLOTS], Tb[SLOTS];
  float dpix0[SLOTS], dpix1[SLOTS], dpix2[SLOTS], ddep[SLOTS], dalp[SLOTS];
  int lastc[SLOTS];
  float dxp[CE > 0 ? CE : 1];  // dL_dpix of the extra channels (CE > 0 implies one pixel per lane)
  const int q0 = (int)part * SLOTS, q1 = q0 + SLOTS - 1;
  const float rx0 = (float)(tx * TILE + (q0 & 1) * 8), rx1 = (float)(tx * TILE + (q1 & 1) * 8 + 7);
  const float ry0 = (float)(ty * TILE + (q0 >> 1) * 8), ry1 = (float)(ty * TILE + (q1 >> 1) * 8 + 7);
  int maxlast = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + s;
    const int px = tx * TILE + (q & 1) * 8 + (int)(lane & 7);
    const int py = ty * TILE + (q >> 1) * 8 + (int)(lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const int p = py * a.W + px;
    pxf[s] = (float)px;
    pyf[s] = (float)py;
    T[s] = inside ? a.final_T[p] : 0.f;
    lastc[s] = inside ? (int)a.n_contrib[p] : 0;
    if (a.loss_gt) {  // (kernel-uniform) fused alpha-mask loss: the pixel's gradient is formed here, see BlendBwdArgs
      dpix0[s] = dpix1[s] = dpix2[s] = ddep[s] = dalp[s] = 0.f;
      if (inside) {
        const float d0 = a.loss_color[p] - a.loss_gt[p], d1 = a.loss_color[plane + p] - a.loss_gt[plane + p];
        const float d2 = a.loss_color[2 * plane + p] - a.loss_gt[2 * plane + p];
        dpix0[s] = d0 > 0.f ? a.loss_sc : (d0 < 0.f ? -a.loss_sc : 0.f);
        dpix1[s] = d1 > 0.f ? a.loss_sc : (d1 < 0.f ? -a.loss_sc : 0.f);
        dpix2[s] = d2 > 0.f ? a.loss_sc : (d2 < 0.f ? -a.loss_sc : 0.f);
        dalp[s] = a.loss_sa * (a.loss_alpha[p] - a.loss_mask[p]);
      }
    } else {
      dpix0[s] = inside ? a.dL_dpix[p] : 0.f;
      dpix1[s] = inside ? a.dL_dpix[plane + p] : 0.f;
      dpix2[s] = inside ? a.dL_dpix[2 * plane + p] : 0.f;
      ddep[s] = inside ? a.dL_ddepth[p] : 0.f;
      dalp[s] = inside ? a.dL_dalpha[p] : 0.f;
    }
    float bgd = bg0 * dpix0[s] + bg1 * dpix1[s] + bg2 * dpix2[s];
    if (CE > 0) {
#pragma unroll
      for (int c = 0; c < CE; c++) {
        // colour triples whose image received no gradient (mask bit clear) are never read and cost nothing below
        const bool on = (a.extra_mask >> (c / 3)) & 1u;
        dxp[c] = (on && inside) ? a.dL_dextra_tri[c / 3][(size_t)(c % 3) * plane + p] : 0.f;
        bgd += (c % 3 == 0 ? bg0 : (c % 3 == 1 ? bg1 : bg2)) * dxp[c];
      }
    }
    Tb[s] = T[s] * bgd;  // T_final * (bg . dL_dpix), over every colour channel
    X[s] = 0.f;
    maxlast = max(maxlast, lastc[s]);
    if constexpr (SLOTS == 1) {
      if (nseg > 1u) {  // (wave-uniform) start in the middle of the list: the forward's checkpoint at this segment's far boundary
        const float *rb = a.ckpt + (size_t)(a.ckpt_base[tile] + seg) * (CKPT_PLANES * 256) + part * 64u + lane;
        const float *rf = a.ckpt + (size_t)(a.ckpt_base[tile] + nseg - 1u) * (CKPT_PLANES * 256) + part * 64u + lane;
        T[s] = inside ? rb[0] : 0.f;  // transmittance in front of entry s_hi (the final T for the last segment)
        // what lies behind the boundary: sum_{k >= s_hi} w_k e_k = dL_dpix . (C_final - C_prefix(s_hi)), channel by channel
        float xb = dpix0[s] * (rf[256] - rb[256]) + dpix1[s] * (rf[512] - rb[512]) + dpix2[s] * (rf[768] - rb[768]) +
                   ddep[s] * (rf[1024] - rb[1024]) + dalp[s] * (rf[1280] - rb[1280]);
        if (CE > 0) {
#pragma unroll
          for (int c = 0; c < CE; c++)
            if ((a.extra_mask >> (c / 3)) & 1u) xb += dxp[c] * (rf[(6 + c) * 256] - rb[(6 + c) * 256]);
        }
        X[s] = inside ? xb : 0.f;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxlast = max(maxlast, __shfl_xor(maxlast, d, WAVE));
  // list entries at front positions >= maxlast contribute to none of this wave's pixels; walked from the back: idx = n - 1 - position
  const int skip = max(n - maxlast, n - s_hi), walk_end = n - s_lo;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int row = (int)(lane >> 4), kcol = (int)(lane & 15);
  // separable reduction: which gradient-row column this lane feeds in the first / second atomic instruction (0xFF = none)
  const bool upper = (lane & 8u) != 0;
  const int jj = (int)(lane & 7u);
  constexpr uint64_t COLA_LO = 0x080F0C0906050200ull, COLA_UP = 0x0B0E0D0A07010403ull;  // bytes jj = 0..7
  constexpr uint64_t COLB_LO = 0xFFFFFF1411181512ull, COLB_UP = 0xFFFF1A1710191613ull;
  const int colA = (int)(((upper ? COLA_UP : COLA_LO) >> (8 * jj)) & 0xFF);
  const int colB = (int)(((upper ? COLB_UP : COLB_LO) >> (8 * jj)) & 0xFF);
  // an extra-colour column is live only if its triple is in the mask
  const bool onA = colA < NACC || ((a.extra_mask >> ((colA - NACC) / 3)) & 1u);
  const bool onB = colB != 0xFF && ((a.extra_mask >> ((colB - NACC) / 3)) & 1u);
  float dps[LFOLD ? 4 : 1][3];  // LFOLD: dL_dpix of the four pixels (lane & 15) + 16 q this lane sums as a reader
  if constexpr (LFOLD) {
    s_rw[lane * 4 + 0] = dpix0[0];
    s_rw[lane * 4 + 1] = dpix1[0];
    s_rw[lane * 4 + 2] = dpix2[0];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 t4 = *reinterpret_cast<const float4 *>(&s_rw[((lane & 15) + 16 * q) * 4]);
      dps[q][0] = t4.x;
      dps[q][1] = t4.y;
      dps[q][2] = t4.z;
    }
    __builtin_amdgcn_wave_barrier();
  }

  for (int base = skip; base < walk_end; base += WAVE) {
    // ---- fetch 64 entries (from the back), cull, compact into LDS in back-to-front order
    const int idx = base + (int)lane;
    bool keep = false;
    float4 r0 = make_float4(0, 0, 0, 0), r2 = make_float4(0, 0, 0, 0);
    const float4 *src = nullptr;
    uint32_t id = 0;
    if (idx < walk_end) {
      id = a.point_list[range.y - 1 - idx];
      src = reinterpret_cast<const float4 *>(a.recs + id);
      r0 = src[0];
      r2 = src[2];
      keep = (r0.x + r2.z >= rx0) && (r0.x - r2.z <= rx1) && (r0.y + r2.w >= ry0) && (r0.y - r2.w <= ry1);
    }
    float4 r1 = make_float4(0, 0, 0, 0);
    if (keep) {  // second, exact filter: ellipse {alpha >= 1/255} against the wave's pixel rectangle
      r1 = src[1];
      keep = ellipse_hits_rect(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, rx0, rx1, ry0, ry1);
    }
    const uint64_t kmask = __ballot(keep);
    const int cnt = __builtin_popcountll(kmask);
    if (keep) {
      const int slot = __builtin_popcountll(kmask & lt);
      // exponent in base 2: p2 = power * log2(e) = dx (qa dx + qb dy) + qc dy dy
      constexpr float L2E = 1.4426950408889634f;
      s0[slot] = make_float4(r0.x, r0.y, (-0.5f * L2E) * r0.z, -L2E * r0.w);
      s1[slot] = make_float4((-0.5f * L2E) * r1.x, __builtin_amdgcn_logf(255.0f * r1.y), __uint_as_float((uint32_t)(n - 1 - idx)), r1.y);
      s2[slot] = make_float4(r1.w, r2.x, r2.y, r1.z);
      s_id[slot] = id;
      if constexpr (DET) {  // this instance's slot: same rectangle arithmetic as the preprocess (CR/auxiliary.h:46-56)
        int bx0, by0, bx1, by1;
        tile_rect(r0.x, r0.y, a.radii[id], a.grid_x, a.grid_y, bx0, by0, bx1, by1);
        const uint32_t k = (uint32_t)((ty - by0) * (bx1 - bx0) + (tx - bx0));
        s_slot[slot] = (a.point_offsets[id] - a.tiles_touched[id] + k) * 4u + part;
      }
      if (CE > 0) {
        const float2 *xs = reinterpret_cast<const float2 *>(a.extra + (size_t)id * CE);
#pragma unroll
        for (int q = 0; q < CE / 2; q++) reinterpret_cast<float2 *>(&s_x[slot * CE])[q] = xs[q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int g = 0; g < cnt; g += 4) {
      float acc[4][NA];
      float wk[4];  // blending weight of this lane's pixel for each of the four Gaussians (0 if not hit)
      uint32_t anyhit = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        // one pixel per lane: the sums are assigned where the Gaussian is evaluated and zeroed where it is skipped; several
        // pixels per lane: zeroed up front and accumulated over the slots
        bool any_pre = false;
        if constexpr (SLOTS > 1) {
          wk[u] = 0.f;
#pragma unroll
          for (int k = 0; k < NA; k++) acc[u][k] = 0.f;
        }
        if (g + u < cnt) {  // wave-uniform
          const float4 g0 = s0[g + u];
          const float4 g1 = s1[g + u];
          const float4 g2 = s2[g + u];
          const int fpos = (int)__float_as_uint(g1.z);  // 0-based position from the front
#pragma unroll
          for (int s = 0; s < SLOTS; s++) {
            const float dx = g0.x - pxf[s], dy = g0.y - pyf[s];
            const float p2 = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;  // power * log2(e)
            const bool pre = (fpos < lastc[s]) && !(p2 > 0.0f) && (p2 + g1.y >= -0.02f);
            if (__ballot(pre) != 0ull) {  // wave-uniform: some lane may reach alpha >= 1/255
              // Select form instead of an exec-masked block: on lanes that are not hit alpha and G are forced to zero, which
              // makes every update below an exact no-op (rc = 1, Tn = T, w = 0, r = 0) -- same results, no second ballot,
              // no mask save / restore, no zero-initialisation of the sums.
              const float G0 = __builtin_amdgcn_exp2f(p2);
              const float alpha0 = fminf(0.99f, g1.w * G0);
              const bool hit = pre && !(alpha0 < 1.0f / 255.0f);
              const float alpha = hit ? alpha0 : 0.f, G = hit ? G0 : 0.f;
              const float rc = __builtin_amdgcn_rcpf(1.f - alpha);
              const float Tn = T[s] * rc;  // transmittance in front of this Gaussian
              const float w = alpha * Tn;  // blending weight = d(pixel)/d(colour)
              float e = g2.x * dpix0[s] + g2.y * dpix1[s] + g2.z * dpix2[s] + g2.w * ddep[s] + dalp[s];
              if (CE > 0) {
#pragma unroll
                for (int t = 0; t < CE / 3; t++)
                  if ((a.extra_mask >> t) & 1u) {  // wave-uniform
#pragma unroll
                    for (int c = 3 * t; c < 3 * t + 3; c++) e += s_x[(g + u) * CE + c] * dxp[c];
                  }
              }
              const float dL_dalpha = Tn * e - (X[s] + Tb[s]) * rc;
              X[s] += w * e;
              T[s] = Tn;
              // moments of r = G * dL_dalpha over the pixels; preprocess_bwd.hip turns them into dL_dmean2D / dL_dconic
              const float r = G * dL_dalpha;
              float v[NA];
              if constexpr (LFOLD) {
                v[0] = r;
                v[1] = w;
              } else if constexpr (SEP && HIER) {
                v[0] = r;  // the x- and y-weights are applied after the folds
                v[1] = w * dpix0[s];
                v[2] = w * dpix1[s];
                v[3] = w * dpix2[s];
              } else if constexpr (SEP) {
                const float ry = r * dy;  // only the y-weights go in per lane; the x-weights are applied to the COLUMN sums
                v[0] = r;
                v[1] = ry;
                v[2] = ry * dy;
                v[3] = w * dpix0[s];
                v[4] = w * dpix1[s];
                v[5] = w * dpix2[s];
              } else {
                const float rx = r * dx, ry = r * dy;
                v[0] = rx;
                v[1] = ry;
                v[2] = rx * dx;
                v[3] = rx * dy;
                v[4] = ry * dy;
                v[5] = r;
                v[6] = w * dpix0[s];
                v[7] = w * dpix1[s];
                v[8] = w * dpix2[s];
              }
#pragma unroll
              for (int k = 0; k < NA; k++) acc[u][k] = SLOTS == 1 ? v[k] : acc[u][k] + v[k];
              if (CE > 0) wk[u] = w;
              anyhit |= 1u << u;
              any_pre = true;
            }
          }
        }
        if constexpr (SLOTS == 1) {
          if (!any_pre) {  // skipped (wave-uniform): its sums must read as zero in the reduction
            wk[u] = 0.f;
#pragma unroll
            for (int k = 0; k < NA; k++) acc[u][k] = 0.f;
          }
        }
      }
      if (anyhit) {  // wave-uniform
        // after the swap folds row 0/1/2/3 = Gaussian g+0 / g+2 / g+1 / g+3; the LDS transposition reads row r = Gaussian g + r
        const int u_of_row = LFOLD ? row : (((row & 1) << 1) | (row >> 1));
        const bool row_live = ((anyhit >> u_of_row) & 1u) && (g + u_of_row < cnt);
        const uint32_t gid = row_live ? s_id[min(g + u_of_row, WAVE - 1)] : 0u;
        if constexpr (SEP) {
          const float2 gxy = *reinterpret_cast<const float2 *>(&s0[g + u_of_row]);  // this row's Gaussian centre
          const float dxr = gxy.x - pxf[0];                // against this lane's column
          float qa, qb, qc, ka, kb;
          if constexpr (LFOLD) {
            *reinterpret_cast<float4 *>(&s_rw[lane * LROW]) = make_float4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
            *reinterpret_cast<float4 *>(&s_rw[lane * LROW + 4]) = make_float4(acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float2 rw[4];
#pragma unroll
            for (int q = 0; q < 4; q++) rw[q] = *reinterpret_cast<const float2 *>(&s_rw[((int)(lane & 15) + 16 * q) * LROW + 2 * row]);
            __builtin_amdgcn_wave_barrier();  // the next group's stores stay behind these reads
            // this lane's four pixels: column lx = lane & 7, rows ly = 2 q + h, h = (lane >> 3) & 1
            const float h = upper ? 1.f : 0.f;
            const float e0 = (rw[0].x + rw[1].x) + (rw[2].x + rw[3].x);
            const float A1 = rw[1].x + 2.f * rw[2].x + 3.f * rw[3].x;   // sum q r
            const float A2 = rw[1].x + 4.f * rw[2].x + 9.f * rw[3].x;   // sum q^2 r
            const float s1 = 2.f * A1 + h * e0;                         // sum ly r
            const float s2 = 4.f * A2 + h * (4.f * A1 + e0);            // sum ly^2 r   (h^2 = h)
            const float Dy = gxy.y - (pyf[0] - (float)(lane >> 3));     // mean.y - top pixel row of the quadrant
            float c[6];
            c[0] = e0;
            c[1] = Dy * e0 - s1;                        // partial column sum of r dy
            c[2] = Dy * (Dy * e0 - 2.f * s1) + s2;      // partial column sum of r dy^2
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
              c[3 + ch] = (rw[0].y * dps[0][ch] + rw[1].y * dps[1][ch]) + (rw[2].y * dps[2][ch] + rw[3].y * dps[3][ch]);
            // from here on exactly the state the swap folds leave: lane j of a row holds the sums over its four rows of column j & 7
            qa = pack_halves(c[0], c[1], upper);
            qb = qa * dxr;
            const float c2 = c[2] + dpp_f<0x128>(c[2]);
            qc = upper ? c2 : qb * dxr;
            ka = pack_halves(c[3], c[4], upper);
            kb = c[5] + dpp_f<0x128>(c[5]);
          } else if constexpr (HIER) {
            float a1ab, a1cd, b1, d1;
            const float a0ab = fold32_parts(acc[0][0], acc[1][0], a1ab), a0cd = fold32_parts(acc[2][0], acc[3][0], a1cd);
            const float b0 = fold16_parts(a0ab, a0cd, b1);  // sum over b2, b1      | b1 = 1 part
            const float d0 = fold16_parts(a1ab, a1cd, d1);  // b2 = 1, sum over b1  | b2 = b1 = 1 part
            const float b0t = dpp_f<0x128>(b0), b1t = dpp_f<0x128>(b1), d0t = dpp_f<0x128>(d0), d1t = dpp_f<0x128>(d1);
            const float e0 = b0 + b0t, e_b0 = upper ? b0 : b0t;
            const float e_b1 = b1 + b1t, e_b1b0 = upper ? b1 : b1t;
            const float e_b2 = d0 + d0t, e_b2b0 = upper ? d0 : d0t;
            const float e_b2b1 = d1 + d1t;
            const float s1 = 4.f * e_b2 + 2.f * e_b1 + e_b0;
            const float s2 = 16.f * (e_b2 + e_b2b1) + 4.f * (e_b1 + e_b1b0) + 8.f * e_b2b0 + e_b0;
            const float Dy = gxy.y - (pyf[0] - (float)(lane >> 3));  // mean.y - top pixel row of the quadrant
            const float c1 = Dy * e0 - s1;                           // column sum of r dy
            const float c2 = Dy * (Dy * e0 - 2.f * s1) + s2;         // column sum of r dy^2
            qa = upper ? c1 : e0;
            qb = qa * dxr;
            qc = upper ? c2 : qb * dxr;
            const float k0 = fold16(fold32(acc[0][1], acc[1][1]), fold32(acc[2][1], acc[3][1]));
            const float k1 = fold16(fold32(acc[0][2], acc[1][2]), fold32(acc[2][2], acc[3][2]));
            const float k2 = fold16(fold32(acc[0][3], acc[1][3]), fold32(acc[2][3], acc[3][3]));
            ka = pack_halves(k0, k1, upper);
            kb = k2 + dpp_f<0x128>(k2);
          } else {
            float c[NA];
#pragma unroll
            for (int k = 0; k < NA; k++) c[k] = fold16(fold32(acc[0][k], acc[1][k]), fold32(acc[2][k], acc[3][k]));
            // lane j of a row now holds the sum over ly = (j >> 3) mod 2 of column lx = j & 7
            qa = pack_halves(c[0], c[1], upper);            // lower: col-sum r        upper: col-sum r dy
            qb = qa * dxr;                                  // lower: col-sum r dx     upper: col-sum r dx dy
            const float c2 = c[2] + dpp_f<0x128>(c[2]);     // col-sum r dy^2 (used by the upper half)
            qc = upper ? c2 : qb * dxr;                     // lower: col-sum r dx^2   upper: col-sum r dy^2
            ka = pack_halves(c[NA - 3], c[NA - 2], upper);  // lower: red              upper: green
            kb = c[NA - 1] + dpp_f<0x128>(c[NA - 1]);       // blue in both halves
          }
          if constexpr (LFOLD) {
            // second LDS hop instead of 5 x 3 DPP steps + a select chain: lane (row, half, jx) leaves its five half-row values as
            // [row][half][value][jx]; lane (row, column k < 9) reads the eight jx of ITS (half, value) -- two 16-byte loads -- adds them
            // and owns column k of the Gaussian's gradient row, which is the layout the atomic wants
            float *t2 = &s_rw[((row * 2 + (upper ? 1 : 0)) * 5) * 8 + jj];
            t2[0] = qa;
            t2[8] = qb;
            t2[16] = qc;
            t2[24] = ka;
            t2[32] = kb;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // column k: 0 r dx (lower qb) 1 r dy (upper qa) 2 r dx^2 (lower qc) 3 r dx dy (upper qb) 4 r dy^2 (upper qc)
            //           5 r (lower qa)    6 red (lower ka)  7 green (upper ka)   8 blue (lower kb)
            constexpr uint64_t HALF_OF_K = 0x000000000000009Aull;                  // bit k: upper half
            constexpr uint64_t VAL_OF_K = 0x0000000433021201ull >> 0;              // nibble k: which of qa..kb (0..4)
            const int k9 = kcol < 9 ? kcol : 0;
            const int hk = (int)((HALF_OF_K >> k9) & 1u), vk = (int)((VAL_OF_K >> (4 * k9)) & 0xFu);
            const float4 *src8 = reinterpret_cast<const float4 *>(&s_rw[((row * 2 + hk) * 5 + vk) * 8]);
            const float4 lo4 = src8[0], hi4 = src8[1];
            __builtin_amdgcn_wave_barrier();  // the next group's (r, w) stores stay behind these reads
            const float vsum = ((lo4.x + lo4.y) + (lo4.z + lo4.w)) + ((hi4.x + hi4.y) + (hi4.z + hi4.w));
            // (a.debug_skip_atomics: measurement only -- what the kernel costs WITHOUT its gradient-row atomics)
            if (row_live && kcol < NACC && !a.debug_skip_atomics) atomicAdd(&a.grad_rows[(size_t)gid * ROWF + kcol], vsum);
          } else {
          qa = half8_sum(qa);
          qb = half8_sum(qb);
          qc = half8_sum(qc);
          ka = half8_sum(ka);
          kb = half8_sum(kb);
          // one value per lane: jj = 0 qb, 1 qc, 2 qa, 3 ka, (7, lower) kb; colA maps (half, jj) to the gradient-row column
          float v = qb;
          v = jj == 1 ? qc : v;
          v = jj == 2 ? qa : v;
          v = jj == 3 ? ka : v;
          if constexpr (CE == 0) {
            v = jj == 7 ? kb : v;
            if constexpr (DET) {
              if (row_live && colA < NACC) a.det_rows[(size_t)s_slot[min(g + u_of_row, WAVE - 1)] * GROW + colA] = v;
            } else {
              if (row_live && colA < NACC) atomicAdd(&a.grad_rows[(size_t)gid * ROWF + colA], v);
            }
          } else {
            // extra colour sums, a triple at a time: P = (channel 3t | channel 3t+1) packed in the half rows, S = channel 3t+2
            constexpr int NT = CE / 3;
            float P[NT], S[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
              P[t] = 0.f;
              S[t] = 0.f;
              if ((a.extra_mask >> t) & 1u) {  // wave-uniform
                float e3[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                  const float d = dxp[3 * t + j];
                  e3[j] = fold16(fold32(wk[0] * d, wk[1] * d), fold32(wk[2] * d, wk[3] * d));
                }
                P[t] = half8_sum(pack_halves(e3[0], e3[1], upper));
                S[t] = half8_sum(e3[2] + dpp_f<0x128>(e3[2]));
              }
            }
            // first atomic: columns 0..15 (lower: Mx Mxx M0 K0 ch0 ch3 ch6 K2 | upper: Mxy Myy My K1 ch1 ch4 ch5 ch2)
            v = jj == 4 ? P[0] : v;
            v = jj == 5 ? P[1] : v;
            v = jj == 6 ? (upper ? S[1] : P[2]) : v;
            v = jj == 7 ? (upper ? S[0] : kb) : v;
            // second atomic: columns 16..26 (lower: ch9 ch12 ch15 ch8 ch11 | upper: ch10 ch13 ch16 ch7 ch14 ch17)
            float v1 = P[3];
            v1 = jj == 1 ? P[4] : v1;
            v1 = jj == 2 ? P[5] : v1;
            v1 = jj == 3 ? (upper ? P[2] : S[2]) : v1;
            v1 = jj == 4 ? (upper ? S[4] : S[3]) : v1;
            v1 = jj == 5 ? S[5] : v1;
            if (row_live) {
              if (onA) atomicAdd(&a.grad_rows[(size_t)gid * ROWF + colA], v);
              if (onB) atomicAdd(&a.grad_rows[(size_t)gid * ROWF + colB], v1);
            }
          }
          }  // !LFOLD
        } else {
        // four Gaussians x nine values reduced together
        float v;
        if constexpr (RED == 0) {
          float red[NACC];
#pragma unroll
          for (int k = 0; k < NACC; k++) red[k] = row16_sum(fold16(fold32(acc[0][k], acc[1][k]), fold32(acc[2][k], acc[3][k])));
          v = red[0];
#pragma unroll
          for (int k = 1; k < NACC; k++) v = (kcol == k) ? red[k] : v;
        } else {
          const float sel = (row == (kcol & 3)) ? 1.f : 0.f;  // Sel[r][j], this lane = (r = row, j = kcol)
          f32x4 d2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < NACC; k++) {
            const float f = fold16(fold32(acc[0][k], acc[1][k]), fold32(acc[2][k], acc[3][k]));
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, sel, z, 0, 0, 0);
            const float p = (d1[0] + d1[1]) + (d1[2] + d1[3]);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(p, (kcol == k) ? 1.f : 0.f, d2, 0, 0, 0);
          }
          v = row == 0 ? d2[0] : (row == 1 ? d2[1] : (row == 2 ? d2[2] : d2[3]));
        }
        static_assert(SEP || CE == 0, "the feature channels use the separable reduction");
        if (row_live && kcol < NACC) atomicAdd(&a.grad_rows[(size_t)gid * ROWF + kcol], v);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // keep the next batch's LDS writes behind this batch's reads
  }
  if constexpr (WPT == 4) {
    if (a.trace && lane == 0) {
      unsigned long long *r = a.trace + (size_t)blockIdx.x * 4u;
      r[0] = trace_t0;
      r[1] = __builtin_amdgcn_s_memrealtime();
      r[2] = (unsigned long long)(walk_end - skip > 0 ? walk_end - skip : 0) | ((unsigned long long)n << 32);
      r[3] = (unsigned long long)tile | ((unsigned long long)seg << 32);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The plain pass as the library runs it by default (one wave per 8x8 quadrant, reductions through LDS): the design of the LFOLD
// branch of blend_backward_kernel<1, 3, 0>, rewritten on its own after a census of that kernel's ISA against the measured cost of
// every instruction class (round 4, DESIGN.md section 4 "Census").
//
// What the census said.  Rounds 2-3 read the kernel as bound by vector-instruction issue ("vector ALUs 99 % busy" -- with a factor
// four cycles per instruction that this part does not have: a wave64 VALU instruction issues in two).  Taking 15 % of its VALU
// instructions out changed nothing (191.8 -> 195.0 us, same box, interleaved): it is not.  Its busiest unit is the LDS ARRAY,
// priced per wave-instruction by MI355X_MICROARCH.md (LDS): a 16-byte read costs 4 cycles even when every lane reads the same
// address, a 12-byte read 8 (!), a 16-byte store 13, and one CU's four SIMDs share one array.  Per surviving (Gaussian, quadrant)
// the round-3 kernel spent  12 (cut-off rows: b128 + b96) + 6 (colour row + opacity) + 70 / 4 (reduction per group of four: two
// b128 stores 26, two read2_b64 16, second hop 16 + 8, id + centre 4) = 35.5 LDS cycles  -- 8,867 survivors per CU x 35.5 =
// 315 k cycles = 132 of the kernel's 192 us at 2.39 GHz -- against ~100 cycles of VALU issue per SIMD.  Hence, here:
//   * every survivor row is read in whole 16-byte words (cut-off test: two b128 = 8 cycles, opacity rides along; a contributing
//     survivor: one b128 = 4), never as 12 bytes;
//   * hop 1 of the reduction is PLANAR and lane-linear: plane (value u = r or w of Gaussian 0..3) x lane, written with eight
//     ds_write_addtid_b32 (address = M0 + offset + 4 x lane: no address VGPR, 2 cycles each = 16 instead of 26), and the lane <->
//     pixel assignment of the wave is chosen so that the four pixels a reducer lane sums are four CONSECUTIVE lanes:
//         lane l:  c = l >> 2, q = l & 3  ->  pixel column lx = c & 7, row ly = 2 q + (c >> 3)
//     so that reducer lane (row = Gaussian, c) fetches r and w of its pixels with two b128 reads (8 cycles instead of 16), free of
//     bank conflicts (16 lanes x 16 bytes = the 64 banks);
//   * 17.5 -> 13 LDS cycles per survivor in the reduction, 18 -> 12 in the walk: 25 instead of 35.5.
// The instruction trims of the first rewrite stay (they cost nothing): row offsets in a VGPR the compiler cannot prove uniform
// (no v_mov of a scalar address in front of every LDS read), zeros of a skipped survivor written in the skip path only, X and
// T_final (bg . dL_dpix) as ONE state variable, dL_dalpha opening the dot product as an FMA, DEPTH = false (fused alpha-mask
// loss) without the depth term, row-live from the hit bits alone, gradient-row byte offsets instead of ids in LDS and the atomic
// in its scalar-base form, hardware reciprocals in the batch step's ellipse filter.
// Same algorithm, same order of the floating-point sums in the reduction; the per-pixel recurrence rounds e and X + Tb differently
// (within the 1e-4 contract; the deterministic kernel above is a different instantiation and unchanged).
template <typename T>
__device__ __forceinline__ T lds_at(const void *base, uint32_t byte_off) {
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
// the compiler narrows a 16-byte LDS read to the components it sees used before the next branch (12 bytes: twice the LDS cycles
// of 16); naming the fourth component as an asm operand right after the load keeps the read whole
__device__ __forceinline__ void keep_whole(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

template <bool DEPTH>
__global__ __launch_bounds__(WAVE) void blend_backward_lds_kernel(const BlendBwdArgs a) {
  // ONE block, laid out by hand: ids 256 + records 3 x 1,024 + reduction buffer 2,048 = 5,376 bytes (five 1,280-byte granules).
  // The ids come FIRST: the reducer rows of a last, partial group read up to three words past the 64 ids -- into s0, never out of
  // the allocation -- so the index needs no clamp.
  constexpr uint32_t RW_BYTES = 8 * WAVE * 4;  // hop 1: eight planes x 64 lanes; hop 2 (320 floats) and the prologue reuse it
  __shared__ __attribute__((aligned(16))) unsigned char lds[256 + 3 * 1024 + RW_BYTES];
  uint32_t *s_id = reinterpret_cast<uint32_t *>(lds);      // Gaussian id x 64: the byte offset of its gradient row
  float4 *s0 = reinterpret_cast<float4 *>(lds + 256);      // x, y, qa, qb      (qa = -conic_a log2(e)/2, qb = -conic_b log2(e))
  float4 *s1 = reinterpret_cast<float4 *>(lds + 1280);     // qc, log2(255*opacity), front position (bits), opacity   (qc = -conic_c log2(e)/2)
  float4 *s2 = reinterpret_cast<float4 *>(lds + 2304);     // r, g, b, depth
  float *s_rw = reinterpret_cast<float *>(lds + 3328);
  const uint32_t rw_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_rw);  // LDS byte address of the reduction buffer

  const unsigned long long trace_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const int omode = tile_order_mode(a.order);
  const uint32_t n_slots = tile_slots_of(a.order, a.grid_x, a.grid_y, omode);
  const uint32_t item = omode ? ordered_item4(blockIdx.x, n_slots) : (blockIdx.x < n_slots * 4u ? xcd_remap_b(blockIdx.x, n_slots * 4u) : n_slots * 4u);
  const uint32_t entry = tile_of_slot(a.order, omode, item / 4u, n_slots);
  if (entry == ORDER_NO_TILE) return;  // (wave-uniform) padding slot, or beyond this frame's slots
  const uint32_t tile = order_entry_tile(entry), seg = order_entry_seg(entry), nseg = order_entry_nseg(entry), part = item % 4u;
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const uint32_t lane = threadIdx.x;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);
  // this wave's SEGMENT of the list, front positions [s_lo, s_hi): the whole list unless the frame cut it (gsr_common.h "list segments")
  const int seg_len = nseg > 1u ? segment_len(n, (int)nseg) : n;
  const int s_lo = (int)seg * seg_len, s_hi = min(n, s_lo + seg_len);
  if (seg != 0u && s_lo >= n) return;
  list_priority(a.order, s_hi - s_lo, a.list_prio);
  const size_t plane = (size_t)a.H * a.W;

  // lane <-> pixel: c = lane >> 2, q = lane & 3 -> column c & 7, row 2 q + (c >> 3) of the quadrant (see above)
  const uint32_t wc = lane >> 2, wq = lane & 3u;
  const uint32_t lx = wc & 7u, ly = 2u * wq + (wc >> 3);
  const int x0 = tx * TILE + (int)(part & 1u) * 8, y0 = ty * TILE + (int)(part >> 1) * 8;
  const float rx0 = (float)x0, rx1 = (float)(x0 + 7), ry0 = (float)y0, ry1 = (float)(y0 + 7);
  const int px = x0 + (int)lx, py = y0 + (int)ly;
  const bool inside = px < a.W && py < a.H;
  const int p = py * a.W + px;
  const float pxf = (float)px, pyf = (float)py;
  float T = inside ? a.final_T[p] : 0.f;
  const int lastc = inside ? (int)a.n_contrib[p] : 0;
  float dpix0 = 0.f, dpix1 = 0.f, dpix2 = 0.f, ddep = 0.f, dalp = 0.f;
  if (a.loss_gt) {  // (kernel-uniform) fused alpha-mask loss: the pixel's gradient is formed here, see BlendBwdArgs
    if (inside) {
      const float d0 = a.loss_color[p] - a.loss_gt[p], d1 = a.loss_color[plane + p] - a.loss_gt[plane + p];
      const float d2 = a.loss_color[2 * plane + p] - a.loss_gt[2 * plane + p];
      dpix0 = d0 > 0.f ? a.loss_sc : (d0 < 0.f ? -a.loss_sc : 0.f);
      dpix1 = d1 > 0.f ? a.loss_sc : (d1 < 0.f ? -a.loss_sc : 0.f);
      dpix2 = d2 > 0.f ? a.loss_sc : (d2 < 0.f ? -a.loss_sc : 0.f);
      dalp = a.loss_sa * (a.loss_alpha[p] - a.loss_mask[p]);
    }
  } else if (inside) {
    dpix0 = a.dL_dpix[p];
    dpix1 = a.dL_dpix[plane + p];
    dpix2 = a.dL_dpix[2 * plane + p];
    if (DEPTH) ddep = a.dL_ddepth[p];
    dalp = a.dL_dalpha[p];
  }
  // XT = X + T_final (bg . dL_dpix): the suffix sum of w e and the background term only ever appear together
  float XT = T * (a.bg[0] * dpix0 + a.bg[1] * dpix1 + a.bg[2] * dpix2);
  if (nseg > 1u) {  // (wave-uniform) start in the middle of the list: the forward's checkpoint at this segment's far boundary
    const uint32_t fl = 8u * ly + lx;  // the forward's lane of this pixel (checkpoint planes are written in its layout)
    const float *rb = a.ckpt + (size_t)(a.ckpt_base[tile] + seg) * (CKPT_PLANES * 256) + part * 64u + fl;
    const float *rf = a.ckpt + (size_t)(a.ckpt_base[tile] + nseg - 1u) * (CKPT_PLANES * 256) + part * 64u + fl;
    // what lies behind the boundary: sum_{k >= s_hi} w_k e_k = dL_dpix . (C_final - C_prefix(s_hi)), channel by channel
    const float xb = dpix0 * (rf[256] - rb[256]) + dpix1 * (rf[512] - rb[512]) + dpix2 * (rf[768] - rb[768]) +
                     ddep * (rf[1024] - rb[1024]) + dalp * (rf[1280] - rb[1280]);
    XT += inside ? xb : 0.f;
    T = inside ? rb[0] : 0.f;  // transmittance in front of entry s_hi (the final T for the last segment)
  }
  int maxlast = lastc;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxlast = max(maxlast, __shfl_xor(maxlast, d, WAVE));
  // list entries at front positions >= maxlast contribute to none of this wave's pixels; walked from the back: idx = n - 1 - position
  const int skip = max(n - maxlast, n - s_hi), walk_end = n - s_lo;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  // reducer side: lane (row = Gaussian of the group, kcol = lane & 15) sums the pixels of writer lanes 4 kcol .. 4 kcol + 3:
  // column kcol & 7, rows 2 q + (kcol >> 3)
  const uint32_t row = lane >> 4, kcol = lane & 15u;
  const bool upper = (lane & 8u) != 0;
  const uint32_t jj = lane & 7u;
  const uint32_t rowbit = 1u << row, row16 = row * 16u;
  const float hh = upper ? 1.f : 0.f;
  const float colx = (float)(x0 + (int)jj), qtop = (float)y0;  // the reducer's pixel column; top pixel row of the quadrant
  float dps[3][4];  // dL_dpix of the reducer's four pixels
  s_rw[lane] = dpix0;
  s_rw[WAVE + lane] = dpix1;
  s_rw[2 * WAVE + lane] = dpix2;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int ch = 0; ch < 3; ch++) {
    const float4 t4 = *reinterpret_cast<const float4 *>(&s_rw[ch * WAVE + kcol * 4u]);
    dps[ch][0] = t4.x;
    dps[ch][1] = t4.y;
    dps[ch][2] = t4.z;
    dps[ch][3] = t4.w;
  }
  __builtin_amdgcn_wave_barrier();
  // hop 1, reader side: r of Gaussian `row` in plane row, w in plane 4 + row; four consecutive lanes = this reducer's pixels
  const float4 *rd_r = reinterpret_cast<const float4 *>(&s_rw[row * WAVE + kcol * 4u]);
  const float4 *rd_w = reinterpret_cast<const float4 *>(&s_rw[(4u + row) * WAVE + kcol * 4u]);
  // second hop: column k < 9 of the gradient row this lane owns
  //   column k: 0 r dx (lower qb) 1 r dy (upper qa) 2 r dx^2 (lower qc) 3 r dx dy (upper qb) 4 r dy^2 (upper qc)
  //             5 r (lower qa)    6 red (lower ka)  7 green (upper ka)   8 blue (lower kb)
  // Hop 2 is lane-linear as well: value v (qa qb qc ka kb) of lane l goes to plane v at float base[v] + l -- five ds_write_addtid_b32
  // (2 LDS cycles each, no address register) instead of five ds_write_b32 (4 each).  Lane l = 16 row + 8 half + jx, so the eight
  // partial sums of gradient-row column k of a reducer row ARE eight consecutive floats of a plane: the reader's two 16-byte reads.
  // The plane bases were found by search (tools/lds_layout_search.py) against the LDS banking rules of MI355X_MICROARCH.md --
  // 16-byte reads are served in groups of 16 lanes that MIX two reducer rows, 64 banks -- so that neither read meets a bank
  // conflict: qa 68, qb 0, qc 132, ka 224, kb 292 floats.  (Round 3's dense [row][half][value][8] rows put columns 5 and 7 -- and
  // rows 0 and 1 -- on the same banks: 8 extra LDS cycles on each of the two reads, 8.6 M conflict cycles per launch,
  // profiles/r4a_pmc.csv.)  lower values qa qb qc ka kb -> columns 5 0 2 6 8, upper -> 1 3 4 7 (its kb is the same sum: stored, unread).
  constexpr uint32_t H2_QA = 68, H2_QB = 0, H2_QC = 132, H2_KA = 224, H2_KB = 292;
  // float offset / 4 of column k's eight sums inside a reducer row's 16 lanes, 7 bits each
  constexpr uint64_t H2_P = ((uint64_t)(H2_QB / 4)) | ((uint64_t)((H2_QA + 8) / 4) << 7) | ((uint64_t)(H2_QC / 4) << 14) |
                            ((uint64_t)((H2_QB + 8) / 4) << 21) | ((uint64_t)((H2_QC + 8) / 4) << 28) | ((uint64_t)(H2_QA / 4) << 35) |
                            ((uint64_t)(H2_KA / 4) << 42) | ((uint64_t)((H2_KA + 8) / 4) << 49) | ((uint64_t)(H2_KB / 4) << 56);
  static_assert((H2_KB + WAVE) * 4 <= RW_BYTES, "hop 2 fits the reduction buffer");
  const uint32_t k9 = kcol < 9u ? kcol : 0u;
  const float4 *src8 = reinterpret_cast<const float4 *>(&s_rw[row * 16u + 4u * (uint32_t)((H2_P >> (7u * k9)) & 127u)]);

  // the list entry of the NEXT batch is requested a batch ahead, and a batch's three 16-byte record words in ONE round trip (the
  // middle word used to wait for the box test): one exposed memory round trip per batch instead of three dependent ones
  uint32_t id_next = 0;
  if (skip + (int)lane < walk_end) id_next = a.point_list[range.y - 1 - (skip + (int)lane)];
  for (int base = skip; base < walk_end; base += WAVE) {
    // ---- fetch 64 entries (from the back), cull, compact into LDS in back-to-front order
    const int idx = base + (int)lane;
    bool keep = false;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
    float l255 = 0.f;
    const uint32_t id = id_next;
    if (idx < walk_end) {
      const float4 *src = reinterpret_cast<const float4 *>(a.recs + id);
      r0 = src[0];
      r1 = src[1];
      r2 = src[2];
    }
    if (idx + WAVE < walk_end) id_next = a.point_list[range.y - 1 - (idx + WAVE)];
    if (idx < walk_end) {
      keep = (r0.x + r2.z >= rx0) && (r0.x - r2.z <= rx1) && (r0.y + r2.w >= ry0) && (r0.y - r2.w <= ry1);
      if (keep) {  // second, exact filter: ellipse {alpha >= 1/255} against the wave's pixel rectangle
        l255 = __builtin_amdgcn_logf(255.0f * r1.y);
        keep = ellipse_hits_rect_fast(r0.x, r0.y, r0.z, r0.w, r1.x, l255, rx0, rx1, ry0, ry1);
      }
    }
    const uint64_t kmask = __ballot(keep);
    const int cnt = __builtin_popcountll(kmask);
    if (keep) {
      const int slot = __builtin_popcountll(kmask & lt);
      // exponent in base 2: p2 = power * log2(e) = dx (qa dx + qb dy) + qc dy dy
      constexpr float L2E = 1.4426950408889634f;
      s0[slot] = make_float4(r0.x, r0.y, (-0.5f * L2E) * r0.z, -L2E * r0.w);
      s1[slot] = make_float4((-0.5f * L2E) * r1.x, l255, __uint_as_float((uint32_t)(n - 1 - idx)), r1.y);
      s2[slot] = make_float4(r1.w, r2.x, r2.y, r1.z);
      s_id[slot] = id * (uint32_t)(GROW * sizeof(float));  // (P < 2^25: gsr_rasterize_backward refuses more)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // byte offset of the group's first survivor row (and of this reducer row's id): wave-uniform in fact, kept in VGPRs (the asm
    // hides that from the compiler, which would otherwise move the scalar address into a VGPR in front of every LDS read)
    uint32_t vg = 0u, vi = row * 4u;
    asm volatile("" : "+v"(vg), "+v"(vi));
    for (int g = 0; g < cnt; g += 4, vg += 64u, vi += 16u) {
      float ar[4], aw[4];  // (r, w) of this lane's pixel for the group's four Gaussians (0 where it is not hit)
      uint32_t anyhit = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        bool done = false;
        if (g + u < cnt) {  // wave-uniform
          const float4 g0 = lds_at<float4>(s0, vg + 16u * u);
          float4 g1 = lds_at<float4>(s1, vg + 16u * u);
          keep_whole(g1);
          const int fpos = (int)__float_as_uint(g1.z);  // 0-based position from the front
          const float dx = g0.x - pxf, dy = g0.y - pyf;
          const float p2 = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;  // power * log2(e)
          const bool pre = (fpos < lastc) && !(p2 > 0.0f) && (p2 + g1.y >= -0.02f);
          if (__ballot(pre) != 0ull) {  // wave-uniform: some lane may reach alpha >= 1/255
            // select form: on lanes that are not hit alpha and G are forced to zero, which makes every update below an exact no-op
            // (rc = 1, Tn = T, w = 0, r = 0)
            float4 g2 = lds_at<float4>(s2, vg + 16u * u);
            keep_whole(g2);
            const float G0 = __builtin_amdgcn_exp2f(p2);
            const float alpha0 = fminf(0.99f, g1.w * G0);
            const bool hit = pre && !(alpha0 < 1.0f / 255.0f);
            const float alpha = hit ? alpha0 : 0.f, G = hit ? G0 : 0.f;
            const float rc = __builtin_amdgcn_rcpf(1.f - alpha);
            const float Tn = T * rc;  // transmittance in front of this Gaussian
            const float w = alpha * Tn;  // blending weight = d(pixel)/d(colour)
            float e = __builtin_fmaf(g2.x, dpix0, dalp);
            e = __builtin_fmaf(g2.y, dpix1, e);
            e = __builtin_fmaf(g2.z, dpix2, e);
            if (DEPTH) e = __builtin_fmaf(g2.w, ddep, e);
            const float dL_dalpha = Tn * e - XT * rc;
            XT = __builtin_fmaf(w, e, XT);
            T = Tn;
            ar[u] = G * dL_dalpha;  // r = G dL_dalpha: its moments over the pixels become dL_dmean2D / dL_dconic in preprocess_bwd.hip
            aw[u] = w;
            anyhit |= 1u << u;
            done = true;
          }
        }
        if (!done) {  // (wave-uniform) skipped: its sums must read as zero in the reduction -- two moves on THIS path only
          asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(ar[u]), "=v"(aw[u]));
        }
      }
      if (anyhit) {  // wave-uniform
        // the hit bits already say g + row < cnt (bit u is set inside that test)
        const bool row_live = (anyhit & rowbit) != 0u;
        const uint32_t goff = lds_at<uint32_t>(s_id, vi);   // (stale LDS beyond cnt: never used)
        const float2 gxy = lds_at<float2>(s0, vg + row16);  // this row's Gaussian centre
        // hop 1: eight lane-linear planes, ds_write_addtid_b32 (address = M0 + offset + 4 x lane)
        asm volatile(
            "s_mov_b32 m0, %8\n\t"
            "s_nop 0\n\t"  // (an M0 write needs one wait state before an add-TID LDS instruction reads it)
            "ds_write_addtid_b32 %0 offset:0\n\t"
            "ds_write_addtid_b32 %1 offset:256\n\t"
            "ds_write_addtid_b32 %2 offset:512\n\t"
            "ds_write_addtid_b32 %3 offset:768\n\t"
            "ds_write_addtid_b32 %4 offset:1024\n\t"
            "ds_write_addtid_b32 %5 offset:1280\n\t"
            "ds_write_addtid_b32 %6 offset:1536\n\t"
            "ds_write_addtid_b32 %7 offset:1792"
            :
            : "v"(ar[0]), "v"(ar[1]), "v"(ar[2]), "v"(ar[3]), "v"(aw[0]), "v"(aw[1]), "v"(aw[2]), "v"(aw[3]), "s"(rw_base)
            : "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4 r4 = *rd_r, w4 = *rd_w;
        __builtin_amdgcn_wave_barrier();  // the second hop's stores stay behind these reads
        // this lane's four pixels: column kcol & 7, rows ly = 2 q + h, h = kcol >> 3
        const float dxr = gxy.x - colx;                       // against this lane's column
        const float e0 = (r4.x + r4.y) + (r4.z + r4.w);
        const float A1 = r4.y + 2.f * r4.z + 3.f * r4.w;      // sum q r
        const float A2 = r4.y + 4.f * r4.z + 9.f * r4.w;      // sum q^2 r
        const float sy1 = 2.f * A1 + hh * e0;                 // sum ly r
        const float sy2 = 4.f * A2 + hh * (4.f * A1 + e0);    // sum ly^2 r   (h^2 = h)
        const float Dy = gxy.y - qtop;                        // mean.y - top pixel row of the quadrant
        float c[6];
        c[0] = e0;
        c[1] = Dy * e0 - sy1;                        // partial column sum of r dy
        c[2] = Dy * (Dy * e0 - 2.f * sy1) + sy2;     // partial column sum of r dy^2
#pragma unroll
        for (int ch = 0; ch < 3; ch++) c[3 + ch] = (w4.x * dps[ch][0] + w4.y * dps[ch][1]) + (w4.z * dps[ch][2] + w4.w * dps[ch][3]);
        const float qa = pack_halves(c[0], c[1], upper);
        const float qb = qa * dxr;
        const float c2 = c[2] + dpp_f<0x128>(c[2]);
        const float qc = upper ? c2 : qb * dxr;
        const float ka = pack_halves(c[3], c[4], upper);
        const float kb = c[5] + dpp_f<0x128>(c[5]);
        // second hop: lane (row, half, jx) leaves its half-row values at [row][column][jx] (layout: H2_P above); lane (row, column
        // k < 9) reads the eight jx of ITS column -- two 16-byte loads -- adds them and owns column k of the Gaussian's gradient row
        asm volatile(
            "s_mov_b32 m0, %5\n\t"
            "s_nop 0\n\t"
            "ds_write_addtid_b32 %0 offset:%6\n\t"   // qa  lower: column 5 (r)        upper: column 1 (r dy)
            "ds_write_addtid_b32 %1 offset:%7\n\t"   // qb  lower: column 0 (r dx)     upper: column 3 (r dx dy)
            "ds_write_addtid_b32 %2 offset:%8\n\t"   // qc  lower: column 2 (r dx^2)   upper: column 4 (r dy^2)
            "ds_write_addtid_b32 %3 offset:%9\n\t"   // ka  lower: column 6 (red)      upper: column 7 (green)
            "ds_write_addtid_b32 %4 offset:%10"        // kb  column 8 (blue)
            :
            : "v"(qa), "v"(qb), "v"(qc), "v"(ka), "v"(kb), "s"(rw_base), "n"(H2_QA * 4), "n"(H2_QB * 4), "n"(H2_QC * 4), "n"(H2_KA * 4),
              "n"(H2_KB * 4)
            : "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4 lo4 = src8[0], hi4 = src8[1];
        __builtin_amdgcn_wave_barrier();  // the next group's hop-1 stores stay behind these reads
        const float vsum = ((lo4.x + lo4.y) + (lo4.z + lo4.w)) + ((hi4.x + hi4.y) + (hi4.z + hi4.w));
        // (a.debug_skip_atomics: measurement only -- what the kernel costs WITHOUT its gradient-row atomics)
        if (row_live && kcol < (uint32_t)NACC && !a.debug_skip_atomics)
          atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(a.grad_rows) + (size_t)(goff + kcol * 4u)), vsum);
      }
    }
    __builtin_amdgcn_wave_barrier();  // keep the next batch's LDS writes behind this batch's reads
  }
  if (a.trace && lane == 0) {
    unsigned long long *r = a.trace + (size_t)blockIdx.x * 4u;
    r[0] = trace_t0;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = (unsigned long long)(walk_end - skip > 0 ? walk_end - skip : 0) | ((unsigned long long)n << 32);
    r[3] = (unsigned long long)tile | ((unsigned long long)seg << 32);
  }
}


#ifdef GSR_BUILD_EXPERIMENTS  // built, measured, NOT adopted (DESIGN.md section 4 "MFMA"); python -m mygauhuman_amd.build --experiments
// ---------------------------------------------------------------------------------------------------------------------
// RED = 2: the per-Gaussian sums over the 64 pixels of a quadrant as a matrix product on the (otherwise idle) matrix pipe.
//
// For one survivor the nine gradient sums are  sum_p r[p] m_j(p)  (j = 0..5: moments 1, lx, ly, lx^2, lx ly, ly^2 of the
// QUADRANT-LOCAL pixel coordinate, p = 8 ly + lx) and  sum_p w[p] dL_dpix_c(p)  (c = 0..2): per pixel only TWO values (r, w)
// depend on the Gaussian, every weight is a per-pixel constant of the wave.  That is  D[16 x 16] = A[16 x 64] B[64 x 16]  with
//   A rows 0..7  = r of 8 survivors, rows 8..15 = w of the same 8 survivors (pixels along the contraction index),
//   B cols 0..5  = the moment weights, cols 6..8 = dL_dpix of the wave's pixels, cols 9..15 = 0,
// sixteen v_mfma_f32_16x16x4_f32 per 8 survivors (exact fp32 FMA chains), in the shadow of other waves' VALU work.  MFMA
// contracts over lane>>4 and registers, never over the 64 lanes, so (r, w) take one trip through LDS: the pixel lanes write
// two dwords per survivor, lane (i = l & 15, k = l >> 4) reads back row i, pixels 16k .. 16k+15 (4 ds_read_b128; row stride
// TRS / pixel-group offset TKO make them bank-conflict free).  The result comes out as (row = lane>>4 *4 + reg, col = lane&15):
// a DPP row of 16 lanes holds one survivor's 16 columns -- the layout the packed 64-byte gradient row wants.  The raw
// moments about the quadrant origin are turned into moments about the Gaussian centre (dx = Dx - lx, Dx = mean.x - x0):
//   sum r (A - la)(B - lb) = A B S0 - B S_la - A S_lb + S_lalb   with per-column choices of A, B in {Dx, Dy, 1}
// by one more LDS hop of the 16 x 16 tile (each lane fetches the 4 raw sums its column needs).
// VALU cost of the whole reduction: ~10 instructions per survivor instead of ~42 FMA-equivalents of permlane swaps + DPP.
constexpr int TRS = 88;  // floats per transposition row (64 pixels in 4 groups of 16 at offsets 0, 20, 40, 60)
constexpr int TKO = 20;
constexpr int MG = 8;    // survivors per MFMA group

__global__ __launch_bounds__(WAVE) void blend_backward_mfma_kernel(const BlendBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float s_t[16 * TRS];
  __shared__ float4 s0[WAVE];     // x, y, qa, qb      (qa = -conic_a log2(e)/2, qb = -conic_b log2(e))
  __shared__ float4 s1[WAVE];     // qc, opacity, depth, r   (qc = -conic_c log2(e)/2)
  __shared__ float4 s2[WAVE];     // g, b, log2(255*opacity), front position (bits)
  __shared__ uint32_t s_id[WAVE + 4];

  const int omode = tile_order_mode(a.order);
  const uint32_t n_slots = tile_slots_of(a.order, a.grid_x, a.grid_y, omode);
  const uint32_t item = omode ? ordered_item4(blockIdx.x, n_slots) : (blockIdx.x < n_slots * 4u ? xcd_remap_b(blockIdx.x, n_slots * 4u) : n_slots * 4u);
  const uint32_t entry = tile_of_slot(a.order, omode, item >> 2, n_slots), part = item & 3;
  if (entry == ORDER_NO_TILE || order_entry_seg(entry) != 0u) return;  // (this variant walks a cut list whole, from its first segment's slot)
  const uint32_t tile = order_entry_tile(entry);
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const uint32_t lane = threadIdx.x;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);
  list_priority(a.order, n, a.list_prio);
  const size_t plane = (size_t)a.H * a.W;
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];

  const int x0 = tx * TILE + (int)(part & 1) * 8, y0 = ty * TILE + (int)(part >> 1) * 8;
  const float rx0 = (float)x0, rx1 = (float)(x0 + 7), ry0 = (float)y0, ry1 = (float)(y0 + 7);
  const int px = x0 + (int)(lane & 7), py = y0 + (int)(lane >> 3);
  const bool inside = px < a.W && py < a.H;
  const int p = py * a.W + px;
  const float pxf = (float)px, pyf = (float)py;
  float T = inside ? a.final_T[p] : 0.f;
  const int lastc = inside ? (int)a.n_contrib[p] : 0;
  const float dpix0 = inside ? a.dL_dpix[p] : 0.f;
  const float dpix1 = inside ? a.dL_dpix[plane + p] : 0.f;
  const float dpix2 = inside ? a.dL_dpix[2 * plane + p] : 0.f;
  const float ddep = inside ? a.dL_ddepth[p] : 0.f;
  const float dalp = inside ? a.dL_dalpha[p] : 0.f;
  const float Tb = T * (bg0 * dpix0 + bg1 * dpix1 + bg2 * dpix2);  // T_final * (bg . dL_dpix)
  float X = 0.f;
  int maxlast = lastc;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxlast = max(maxlast, __shfl_xor(maxlast, d, WAVE));
  if (maxlast == 0) return;  // (wave-uniform) nothing was blended into this quadrant
  const int skip = n - maxlast;  // list entries at front positions >= maxlast contribute to none of this wave's pixels
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));

  // ---- B operand (16 registers): column j = lane & 15 of the weights of pixels 16 k + t, k = lane >> 4
  const int mi = (int)(lane & 15), mk = (int)(lane >> 4);
  s_t[lane] = dpix0;
  s_t[64 + lane] = dpix1;
  s_t[128 + lane] = dpix2;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float bw[16];
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const int q = 16 * mk + t;                        // pixel (= lane index of its owner) inside the quadrant
    const float lx = (float)(q & 7), ly = (float)(q >> 3);
    float w = 0.f;
    w = mi == 0 ? 1.f : w;
    w = mi == 1 ? lx : w;
    w = mi == 2 ? ly : w;
    w = mi == 3 ? lx * lx : w;
    w = mi == 4 ? lx * ly : w;
    w = mi == 5 ? ly * ly : w;
    if (mi >= 6 && mi < 9) w = s_t[(mi - 6) * 64 + q];
    bw[t] = w;
  }
  __builtin_amdgcn_wave_barrier();
  // position of this lane's pixel inside a transposition row, and the row segment this lane reads back
  const int wpos = TKO * (int)(lane >> 4) + (int)(lane & 15);
  const float4 *arow = reinterpret_cast<const float4 *>(&s_t[mi * TRS + TKO * mk]);
  // conversion to moments about the Gaussian centre: which raw sums column `mi` needs, and its A / B factors
  const bool mom_lane = mk < 2 && mi < 6, col_lane = mk >= 2 && mi >= 6 && mi < 9;
  const int ipP = mi == 1 || mi == 4 ? 2 : (mi == 5 ? 15 : 1);            // S_la: lx for columns 0, 2, 3; ly for 1, 4
  const int ipQ = mi == 2 ? 1 : (mi == 3 || mi == 4 ? 2 : 15);             // S_lb
  const int ipR = mi >= 2 && mi <= 4 ? mi + 1 : 15;                        // S_lalb (column 15 of D is exactly zero)
  const float aX = (mi == 0 || mi == 2 || mi == 3) ? 1.f : 0.f, aY = (mi == 1 || mi == 4) ? 1.f : 0.f, a1 = mi == 5 ? 1.f : 0.f;
  const float bX = mi == 2 ? 1.f : 0.f, bY = (mi == 3 || mi == 4) ? 1.f : 0.f, b1 = (mi < 2 || mi == 5) ? 1.f : 0.f;
  const float fx0 = (float)x0, fy0 = (float)y0;

  for (int base = skip; base < n; base += WAVE) {
    // ---- fetch 64 entries (from the back), cull, compact into LDS in back-to-front order
    const int idx = base + (int)lane;
    bool keep = false;
    float4 r0 = make_float4(0, 0, 0, 0), r2 = make_float4(0, 0, 0, 0);
    const float4 *src = nullptr;
    uint32_t id = 0;
    if (idx < n) {
      id = a.point_list[range.y - 1 - idx];
      src = reinterpret_cast<const float4 *>(a.recs + id);
      r0 = src[0];
      r2 = src[2];
      keep = (r0.x + r2.z >= rx0) && (r0.x - r2.z <= rx1) && (r0.y + r2.w >= ry0) && (r0.y - r2.w <= ry1);
    }
    float4 r1 = make_float4(0, 0, 0, 0);
    if (keep) {  // second, exact filter: ellipse {alpha >= 1/255} against the wave's pixel rectangle
      r1 = src[1];
      keep = ellipse_hits_rect(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, rx0, rx1, ry0, ry1);
    }
    const uint64_t kmask = __ballot(keep);
    const int cnt = __builtin_popcountll(kmask);
    if (keep) {
      const int slot = __builtin_popcountll(kmask & lt);
      constexpr float L2E = 1.4426950408889634f;
      s0[slot] = make_float4(r0.x, r0.y, (-0.5f * L2E) * r0.z, -L2E * r0.w);
      s1[slot] = make_float4((-0.5f * L2E) * r1.x, r1.y, r1.z, r1.w);
      s2[slot] = make_float4(r2.x, r2.y, __builtin_amdgcn_logf(255.0f * r1.y), __uint_as_float((uint32_t)(n - 1 - idx)));
      s_id[slot] = id;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int g = 0; g < cnt; g += MG) {
      const int ng = min(MG, cnt - g);
      uint32_t anyhit = 0;
      for (int u = 0; u < ng; u++) {  // a real loop: nothing of a survivor stays in registers
        const float4 g0 = s0[g + u];
        const float4 g1 = s1[g + u];
        const float4 g2 = s2[g + u];
        const int fpos = (int)__float_as_uint(g2.w);  // 0-based position from the front
        const float dx = g0.x - pxf, dy = g0.y - pyf;
        const float p2 = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;  // power * log2(e)
        const bool pre = (fpos < lastc) && !(p2 > 0.0f) && (p2 + g2.z >= -0.02f);
        if (__ballot(pre) != 0ull) {
          const float G = __builtin_amdgcn_exp2f(p2);
          const float alpha = fminf(0.99f, g1.y * G);
          const bool hit = pre && !(alpha < 1.0f / 255.0f);
          if (__ballot(hit) != 0ull) {
            float r = 0.f, w = 0.f;
            if (hit) {  // exec-masked body: state changes on hit lanes only
              const float rc = __builtin_amdgcn_rcpf(1.f - alpha);
              const float Tn = T * rc;  // transmittance in front of this Gaussian
              w = alpha * Tn;           // blending weight = d(pixel)/d(colour)
              const float e = g1.w * dpix0 + g2.x * dpix1 + g2.y * dpix2 + g1.z * ddep + dalp;
              const float dL_dalpha = Tn * e - (X + Tb) * rc;
              X += w * e;
              T = Tn;
              r = G * dL_dalpha;
            }
            s_t[u * TRS + wpos] = r;
            s_t[(MG + u) * TRS + wpos] = w;
            anyhit |= 1u << u;
          }
        }
      }
      if (anyhit) {  // wave-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4 a0 = arow[0], a1v = arow[1], a2 = arow[2], a3 = arow[3];
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};  // two chains: a dependent MFMA waits 40 cycles, an independent one 32
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bw[0], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bw[1], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bw[2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bw[3], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.x, bw[4], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.y, bw[5], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.z, bw[6], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v.w, bw[7], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, bw[8], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, bw[9], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, bw[10], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, bw[11], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.x, bw[12], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.y, bw[13], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.z, bw[14], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3.w, bw[15], d1, 0, 0, 0);
        d0 += d1;
        // rows 0..7 of D (lanes with mk < 2): raw moments of survivors g .. g+7; rows 8..15: their colour sums
        __builtin_amdgcn_wave_barrier();
        if (mk < 2) {
#pragma unroll
          for (int r = 0; r < 4; r++) s_t[(4 * mk + r) * 16 + mi] = d0[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int R = 4 * (mk & 1) + r;  // survivor of the group this lane's register r belongs to
          const bool live = (anyhit >> R) & 1u;
          float v = d0[r];
          if (mom_lane) {
            const float *S = &s_t[R * 16];
            const float2 c = *reinterpret_cast<const float2 *>(&s0[g + R]);
            const float Dx = c.x - fx0, Dy = c.y - fy0;
            const float A = aX * Dx + (aY * Dy + a1), B = bX * Dx + (bY * Dy + b1);
            v = A * (B * S[0] - S[ipQ]) - B * S[ipP] + S[ipR];
          }
          if (live && (mom_lane || col_lane)) atomicAdd(&a.grad_rows[(size_t)s_id[g + R] * GROW + mi], v);
        }
        __builtin_amdgcn_wave_barrier();  // the next group's (r, w) rows overwrite the tile
      }
    }
    __builtin_amdgcn_wave_barrier();  // keep the next batch's LDS writes behind this batch's reads
  }
}

#endif  // GSR_BUILD_EXPERIMENTS

// deterministic mode, second step: one thread per (Gaussian, column) adds the Gaussian's slots in index order
__global__ void reduce_det_rows_kernel(int P, const uint32_t *point_offsets, const uint32_t *tiles_touched, const float *det_rows,
                                       size_t n_slots, float *grad_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = i >> 4, col = i & 15;
  if (g >= P) return;
  uint32_t n = tiles_touched[g] * 4u;
  if ((size_t)point_offsets[g] * 4u > n_slots) n = 0;  // a forward that overflowed its capacity rendered nothing: no slots exist
  const float *src = det_rows + (size_t)(point_offsets[g] - tiles_touched[g]) * 4u * GROW + col;
  float s = 0.f;
  for (uint32_t k = 0; k < n; k++) s += src[(size_t)k * GROW];
  grad_rows[(size_t)g * GROW + col] = col < NACC ? s : 0.f;
}
int launch_reduce_det_rows(int P, const uint32_t *point_offsets, const uint32_t *tiles_touched, const float *det_rows,
                           size_t n_slots, float *grad_rows, hipStream_t stream) {
  if (P == 0) return GSR_OK;
  hipLaunchKernelGGL(reduce_det_rows_kernel, dim3((P * 16 + 255) / 256), dim3(256), 0, stream, P, point_offsets, tiles_touched,
                     det_rows, n_slots, grad_rows);
  return GSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The 18-channel (fused multi-feature) backward with its reductions through LDS (knob blend_bwd_reduce = 3, the default).
//
// blend_backward_kernel<1, 0, 18> above folds every channel sum with permlane swaps and DPP (~80 FMA slots per colour triple
// per four Gaussians on top of ~130 for the nine base sums) and sits at 108 VGPRs / 8 KB of LDS.  Here the base sums take the
// two LDS hops of the plain kernel, and the channel sums reuse the first hop: the reducer lane (row = Gaussian, c = column)
// already holds the blending weights w of "its" four pixels c + 16 q, so a channel's column-partial sum is four FMAs against
// that channel's image gradient at those four pixels -- a per-wave table s_dx[channel][c][q] written once in the prologue
// (only the channels of the 6-bit image mask, dynamic LDS) -- and the 16 column partials of a row meet in a second hop
// (one triple at a time: three 4-byte stores, then lanes (row, k < 3) add sixteen values each).  LDS budget: survivors are
// staged 32 at a time (half the plain kernel's batch) so that records + channel colours + transposition buffer + table stay
// below 10 KB per wave for up to ~8 live channels: the occupancy curve in DESIGN.md punishes every wave lost.
constexpr int FB = 32;                 // survivors staged at a time
constexpr int FX_LROW = 10;            // floats per pixel lane in the transposition buffer: 4 x (r, w) + 2 pad (12 as in LFOLD above would
                                       // cost 256 bytes that decide between six and seven LDS granules per wave, see s_rw)
constexpr int FX_BASE2 = 4 * 2 * 5 * 8;  // floats of the base sums' second hop
constexpr int FX_TRI2 = 4 * 3 * 16 + 4;  // floats of one colour triple's second hop (two triples per pass); + 4 (round 4): the second
                                         // triple's 16-float rows start four banks after the first's -- the readers of both (lanes k < 3
                                         // and 3 <= k < 6 of a row, one 16-byte read each) hit the same banks otherwise (2-way conflict)
#ifndef FX_FULL_ROWS
#define FX_FULL_ROWS false
#endif
#ifndef FX_SPLIT_DOT
#define FX_SPLIT_DOT true
#endif

// NL = how many of the six channel triples carry a gradient (popcount of the image mask).  Everything per-channel is sized by it:
// the survivor rows in LDS hold the LIVE triples only (gathered as such: 12 bytes per live triple instead of the 72-byte row), and so
// do the per-pixel gradient registers, the dot product per contributing pair and the second-hop passes.  Phase 1 of the training
// loss (train.py:256-286: image, alpha, normal, axis) has two live triples: 6 channels instead of 18 everywhere -- 24 registers and
// 12 FMAs per pair less than the all-channel kernel, one more wave per SIMD.
template <int NL>
__device__ __forceinline__ void blend_backward_features_body(const BlendBwdArgs &a) {
  constexpr int CE = CE_MAX, CL = 3 * NL;
  // LDS budget (1,280-byte granules: 160 KB / 128): with two live triples 2,816 + 768 + 1,536 + 128 + 2,304 (table) = 7,552 bytes = six
  // granules = 21 waves per CU, so that the registers' five waves per SIMD fit (8,080 bytes took seven: 18 waves per CU)
  constexpr int RW_FLOATS = (WAVE * FX_LROW > FX_BASE2 + 2 * FX_TRI2) ? WAVE * FX_LROW : FX_BASE2 + 2 * FX_TRI2;
  __shared__ __attribute__((aligned(16))) float s_rw[RW_FLOATS];  // hop 1: 64 x FX_LROW; hop 2: base [0, 320) + two triples [320, 704)
  constexpr int XS2 = CL > 0 ? (CL + 1) / 2 : 1;  // float2s per survivor row of live channel colours
  __shared__ float2 s_x[FB * XS2];
  __shared__ float4 s0[FB], s1[FB], s2[FB];
  __shared__ uint32_t s_id[FB];
  extern __shared__ __attribute__((aligned(16))) float s_dx[];  // [live channel][c = pixel & 15][q = pixel >> 4]
  static_assert(FX_LROW % 2 == 0 && FX_LROW >= 8, "hop 1 stores four (r, w) pairs per lane, read back as float2");
  const unsigned long long trace_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;

  const int omode = tile_order_mode(a.order);
  const uint32_t n_slots = tile_slots_of(a.order, a.grid_x, a.grid_y, omode);
  const uint32_t item = omode ? ordered_item4(blockIdx.x, n_slots) : (blockIdx.x < n_slots * 4u ? xcd_remap_b(blockIdx.x, n_slots * 4u) : n_slots * 4u);
  const uint32_t entry = tile_of_slot(a.order, omode, item / 4, n_slots), part = item % 4;
  if (entry == ORDER_NO_TILE) return;
  const uint32_t tile = order_entry_tile(entry), seg = order_entry_seg(entry), nseg = order_entry_nseg(entry);
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const uint32_t lane = threadIdx.x;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);
  // this wave's SEGMENT of the list, front positions [s_lo, s_hi) (gsr_common.h "list segments")
  const int seg_len = nseg > 1u ? segment_len(n, (int)nseg) : n;
  const int s_lo = (int)seg * seg_len, s_hi = min(n, s_lo + seg_len);
  if (seg != 0u && s_lo >= n) return;
  list_priority(a.order, s_hi - s_lo, a.list_prio);
  const size_t plane = (size_t)a.H * a.W;
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
  const uint32_t mask = a.extra_mask;

  const int q0 = (int)part;
  const float rx0 = (float)(tx * TILE + (q0 & 1) * 8), rx1 = rx0 + 7.f;
  const float ry0 = (float)(ty * TILE + (q0 >> 1) * 8), ry1 = ry0 + 7.f;
  const int px = tx * TILE + (q0 & 1) * 8 + (int)(lane & 7), py = ty * TILE + (q0 >> 1) * 8 + (int)(lane >> 3);
  const bool inside = px < a.W && py < a.H;
  const int p = py * a.W + px;
  const float pxf = (float)px, pyf = (float)py;
  float T = inside ? a.final_T[p] : 0.f;
  const int lastc = inside ? (int)a.n_contrib[p] : 0;
  // image gradients: read (gsr_rasterize_backward_ex), or -- fused phase-1 loss, a.p1 -- FORMED here from the forward's images and
  // the targets, any gradient that does arrive (other loss terms) added on top
  const bool fused = a.p1.gt_image != nullptr;  // (kernel-uniform)
  float dpix0 = (inside && a.dL_dpix) ? a.dL_dpix[p] : 0.f, dpix1 = (inside && a.dL_dpix) ? a.dL_dpix[plane + p] : 0.f;
  float dpix2 = (inside && a.dL_dpix) ? a.dL_dpix[2 * plane + p] : 0.f;
  const float ddep = (inside && a.dL_ddepth) ? a.dL_ddepth[p] : 0.f;
  float dalp = (inside && a.dL_dalpha) ? a.dL_dalpha[p] : 0.f;
  float p1_s3 = 0.f;  // bound . upstream / (3 n_bound): the scale of the L1 terms at this pixel (0 outside the bound mask)
  if (fused) {
    const float up = a.p1.upstream ? a.p1.upstream[0] : 1.f;
    const float bnd = (inside && a.p1.bound[p] != 0.f) ? up : 0.f;
    p1_s3 = bnd * a.p1.stats[2];
    const float s3 = a.p1.w_image * p1_s3, s1 = 2.f * a.p1.w_alpha * bnd * a.p1.stats[3];
    if (inside) {
      const float d0 = a.p1.color[p] - a.p1.gt_image[p], d1 = a.p1.color[plane + p] - a.p1.gt_image[plane + p];
      const float d2 = a.p1.color[2 * plane + p] - a.p1.gt_image[2 * plane + p];
      dpix0 += d0 > 0.f ? s3 : (d0 < 0.f ? -s3 : 0.f);
      dpix1 += d1 > 0.f ? s3 : (d1 < 0.f ? -s3 : 0.f);
      dpix2 += d2 > 0.f ? s3 : (d2 < 0.f ? -s3 : 0.f);
      dalp += s1 * (a.p1.alpha[p] - a.p1.alpha_target[p]);
    }
  }
  float bgd = bg0 * dpix0 + bg1 * dpix1 + bg2 * dpix2;
  uint32_t live_t[NL > 0 ? NL : 1];  // the live triples, ascending (wave-uniform)
  {
    uint32_t m = mask;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      live_t[j] = (uint32_t)__builtin_ctz(m | 0x40u);
      m &= m - 1u;
    }
  }
  float dxp[CL > 0 ? CL : 1];
  {
    // table rows 0..2: the colour image's gradient (always live); then the live extra channels
    s_dx[0 * WAVE + (int)(lane & 15u) * 4 + (int)(lane >> 4)] = dpix0;
    s_dx[1 * WAVE + (int)(lane & 15u) * 4 + (int)(lane >> 4)] = dpix1;
    s_dx[2 * WAVE + (int)(lane & 15u) * 4 + (int)(lane >> 4)] = dpix2;
#pragma unroll
    for (int c = 0; c < CL; c++) {
      const uint32_t tri = live_t[c / 3] % (CE / 3);
      const float *img = a.dL_dextra_tri[tri];
      dxp[c] = (inside && img) ? img[(size_t)(c % 3) * plane + p] : 0.f;
      if (fused && inside && (tri == (uint32_t)a.p1.normal_triple || tri == (uint32_t)a.p1.axis_triple)) {  // (wave-uniform test)
        const float wgt = p1_s3 * (tri == (uint32_t)a.p1.normal_triple ? a.p1.w_normal : a.p1.w_axis);
        const float d = a.p1.extra_images[(size_t)(3u * tri + (uint32_t)(c % 3)) * plane + p] - a.p1.gt_normal[(size_t)(c % 3) * plane + p];
        dxp[c] += d > 0.f ? wgt : (d < 0.f ? -wgt : 0.f);
      }
      bgd += (c % 3 == 0 ? bg0 : (c % 3 == 1 ? bg1 : bg2)) * dxp[c];
      s_dx[(3 + c) * WAVE + (int)(lane & 15u) * 4 + (int)(lane >> 4)] = dxp[c];
    }
  }
  const float Tb = T * bgd;  // T_final * (bg . dL_dpix), over every colour channel
  float X = 0.f;
  if (nseg > 1u) {  // (wave-uniform) start in the middle of the list: the forward's checkpoint at this segment's far boundary
    const float *rb = a.ckpt + (size_t)(a.ckpt_base[tile] + seg) * (CKPT_PLANES * 256) + part * 64u + lane;
    const float *rf = a.ckpt + (size_t)(a.ckpt_base[tile] + nseg - 1u) * (CKPT_PLANES * 256) + part * 64u + lane;
    T = inside ? rb[0] : 0.f;  // transmittance in front of entry s_hi (the final T for the last segment)
    // what lies behind the boundary: sum_{k >= s_hi} w_k e_k = dL_dpix . (C_final - C_prefix(s_hi)), channel by channel
    float xb = dpix0 * (rf[256] - rb[256]) + dpix1 * (rf[512] - rb[512]) + dpix2 * (rf[768] - rb[768]) + ddep * (rf[1024] - rb[1024]) +
               dalp * (rf[1280] - rb[1280]);
#pragma unroll
    for (int c = 0; c < CL; c++) {
      const uint32_t pl = (6u + 3u * live_t[c / 3] + (uint32_t)(c % 3)) * 256u;
      xb += dxp[c] * (rf[pl] - rb[pl]);
    }
    X = inside ? xb : 0.f;
  }
  int maxlast = lastc;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxlast = max(maxlast, __shfl_xor(maxlast, d, WAVE));
  // list entries at front positions >= maxlast contribute to none of this wave's pixels; walked from the back: idx = n - 1 - position
  const int skip = max(n - maxlast, n - s_hi), walk_end = n - s_lo;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int row = (int)(lane >> 4), kcol = (int)(lane & 15);
  const bool upper = (lane & 8u) != 0;
  const int jj = (int)(lane & 7u);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // (the table is complete before any reducer lane reads it)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // FULL_ROWS: the whole row (cut-off test, colour, 18 channels = 32 registers) is fetched a survivor ahead: no exposed LDS round
  // trip, 160 VGPRs = three waves per SIMD.  Otherwise only the cut-off rows travel ahead and a contributing survivor fetches its
  // colour + channel rows in ONE round trip: ~128 VGPRs = four waves per SIMD.
  constexpr bool FULL_ROWS = FX_FULL_ROWS;
  struct Row {
    float4 g0, g1, g2;
    float2 x[XS2];
    int k;
  };
  auto fetch_rest = [&](Row &r) {
    r.g2 = s2[r.k];
    if (CL > 0) {
#pragma unroll
      for (int c = 0; c < XS2; c++) r.x[c] = s_x[r.k * XS2 + c];
    }
  };
  auto fetch = [&](Row &r, int k) {  // (a row beyond the last survivor is stale LDS: read, never used)
    r.k = min(k, FB - 1);
    r.g0 = s0[r.k];
    r.g1 = s1[r.k];
    if constexpr (FULL_ROWS) fetch_rest(r);
  };
  for (int base = skip; base < walk_end; base += WAVE) {
    // ---- fetch 64 entries (from the back), cull; the survivors go through LDS in back-to-front order, FB at a time
    const int idx = base + (int)lane;
    bool keep = false;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
    uint32_t id = 0;
    if (idx < walk_end) {
      id = a.point_list[range.y - 1 - idx];
      const float4 *src = reinterpret_cast<const float4 *>(a.recs + id);
      r0 = src[0];
      r2 = src[2];
      keep = (r0.x + r2.z >= rx0) && (r0.x - r2.z <= rx1) && (r0.y + r2.w >= ry0) && (r0.y - r2.w <= ry1);
      if (keep) {
        r1 = src[1];
        keep = ellipse_hits_rect(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, rx0, rx1, ry0, ry1);
      }
    }
    const uint64_t kmask = __ballot(keep);
    const int cnt = __builtin_popcountll(kmask);
    const int slot = __builtin_popcountll(kmask & lt);
    for (int h0 = 0; h0 < cnt; h0 += FB) {
      if (keep && slot >= h0 && slot < h0 + FB) {
        const int sl = slot - h0;
        constexpr float L2E = 1.4426950408889634f;
        s0[sl] = make_float4(r0.x, r0.y, (-0.5f * L2E) * r0.z, -L2E * r0.w);
        s1[sl] = make_float4((-0.5f * L2E) * r1.x, __builtin_amdgcn_logf(255.0f * r1.y), __uint_as_float((uint32_t)(n - 1 - idx)), r1.y);
        s2[sl] = make_float4(r1.w, r2.x, r2.y, r1.z);
        s_id[sl] = id;
        if (CL > 0) {  // the LIVE triples of the Gaussian's channel row only
          const float *xs = a.extra + (size_t)id * CE;
          float *dst = reinterpret_cast<float *>(&s_x[sl * XS2]);
#pragma unroll
          for (int j = 0; j < NL; j++) {
            const float *t3 = xs + 3u * live_t[j];
            dst[3 * j + 0] = t3[0];
            dst[3 * j + 1] = t3[1];
            dst[3 * j + 2] = t3[2];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int m = min(FB, cnt - h0);
      // The survivors' LDS rows travel through registers ONE SURVIVOR AHEAD of the arithmetic (two register sets, A and B, used in
      // turn: A always holds the first survivor of a group).  A wave's walk is a dependent chain and the kernel lasts as long as its
      // slowest chains (tools/wave_trace.py): read where they are used -- the cut-off rows, then the colour row, then one read per live
      // channel triple, each behind its own branch and its own s_waitcnt -- the rows cost four LDS round trips per survivor.
      Row A, B;
      fetch(A, 0);
      for (int g = 0; g < m; g += 4) {
        float accr[4], accw[4];
        uint32_t anyhit = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          accr[u] = 0.f;
          accw[u] = 0.f;
        }
        auto one = [&](Row &r, int u) {
          const int fpos = (int)__float_as_uint(r.g1.z);  // 0-based position from the front
          const float dx = r.g0.x - pxf, dy = r.g0.y - pyf;
          const float p2 = dx * (r.g0.z * dx + r.g0.w * dy) + (r.g1.x * dy) * dy;  // power * log2(e)
          const bool pre = (fpos < lastc) && !(p2 > 0.0f) && (p2 + r.g1.y >= -0.02f);
          if (__ballot(pre) != 0ull) {  // wave-uniform: some lane may reach alpha >= 1/255
            if constexpr (!FULL_ROWS) fetch_rest(r);
            // select form (see blend_backward_kernel): lanes that are not hit run with alpha = G = 0, every update a no-op
            const float G0 = __builtin_amdgcn_exp2f(p2);
            const float alpha0 = fminf(0.99f, r.g1.w * G0);
            const bool hit = pre && !(alpha0 < 1.0f / 255.0f);
            const float alpha = hit ? alpha0 : 0.f, G = hit ? G0 : 0.f;
            const float rc = __builtin_amdgcn_rcpf(1.f - alpha);
            const float Tn = T * rc;  // transmittance in front of this Gaussian
            const float w = alpha * Tn;  // blending weight = d(pixel)/d(colour)
            // the dot product over the live channels in three independent partial sums: as ONE chain it is 5 + CL dependent FMAs
            // (23 with all six triples live, ~9 cycles each for a wave on its own) in front of dL_dalpha
            float e = r.g2.x * dpix0 + r.g2.y * dpix1 + r.g2.z * dpix2 + r.g2.w * ddep + dalp;
            if constexpr (FX_SPLIT_DOT && CL >= 6) {
              float ea = 0.f, eb = 0.f;
#pragma unroll
              for (int c = 0; c < CL; c++) {
                const float xv = (c % 2 == 0 ? r.x[c / 2].x : r.x[c / 2].y);
                if (c % 3 == 0) e = __builtin_fmaf(xv, dxp[c], e);
                else if (c % 3 == 1) ea = __builtin_fmaf(xv, dxp[c], ea);
                else eb = __builtin_fmaf(xv, dxp[c], eb);
              }
              e += ea + eb;
            } else {
#pragma unroll
              for (int c = 0; c < CL; c++) e += (c % 2 == 0 ? r.x[c / 2].x : r.x[c / 2].y) * dxp[c];
            }
            const float dL_dalpha = Tn * e - (X + Tb) * rc;
            X += w * e;
            T = Tn;
            accr[u] = G * dL_dalpha;
            accw[u] = w;
            anyhit |= 1u << u;
          }
        };
        fetch(B, g + 1);
        one(A, 0);
        if (g + 1 < m) {  // (wave-uniform)
          fetch(A, g + 2);
          one(B, 1);
        }
        if (g + 2 < m) {
          fetch(B, g + 3);
          one(A, 2);
        }
        if (g + 3 < m) {
          fetch(A, g + 4);
          one(B, 3);
        }
        if (anyhit) {  // wave-uniform
          const bool row_live = ((anyhit >> row) & 1u) && (g + row < m);
          const uint32_t gid = row_live ? s_id[min(g + row, FB - 1)] : 0u;
          const float2 gxy = *reinterpret_cast<const float2 *>(&s0[min(g + row, FB - 1)]);  // this row's Gaussian centre
          const float dxr = gxy.x - pxf;                                                      // against this lane's column
          // ---- hop 1: (r, w) of the four Gaussians, pixel lanes -> reducer lanes (row = Gaussian, c = lane & 15)
#pragma unroll
          for (int u = 0; u < 4; u++) *reinterpret_cast<float2 *>(&s_rw[lane * FX_LROW + 2 * u]) = make_float2(accr[u], accw[u]);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          float2 rw[4];
#pragma unroll
          for (int q = 0; q < 4; q++) rw[q] = *reinterpret_cast<const float2 *>(&s_rw[((int)(lane & 15) + 16 * q) * FX_LROW + 2 * row]);
          __builtin_amdgcn_wave_barrier();  // the second-hop stores below stay behind these reads
          // ---- base sums (as the LFOLD branch of blend_backward_kernel)
          const float hh = upper ? 1.f : 0.f;
          const float e0 = (rw[0].x + rw[1].x) + (rw[2].x + rw[3].x);
          const float A1 = rw[1].x + 2.f * rw[2].x + 3.f * rw[3].x;   // sum q r
          const float A2 = rw[1].x + 4.f * rw[2].x + 9.f * rw[3].x;   // sum q^2 r
          const float s1y = 2.f * A1 + hh * e0;                       // sum ly r
          const float s2y = 4.f * A2 + hh * (4.f * A1 + e0);          // sum ly^2 r   (h^2 = h)
          const float Dy = gxy.y - (pyf - (float)(lane >> 3));        // mean.y - top pixel row of the quadrant
          float c[6];
          c[0] = e0;
          c[1] = Dy * e0 - s1y;
          c[2] = Dy * (Dy * e0 - 2.f * s1y) + s2y;
#pragma unroll
          for (int ch = 0; ch < 3; ch++) {  // (the colour image's gradient at this reader's four pixels: table rows 0..2, not 12 registers)
            const float4 d4 = *reinterpret_cast<const float4 *>(&s_dx[ch * WAVE + (int)(lane & 15u) * 4]);
            c[3 + ch] = (rw[0].y * d4.x + rw[1].y * d4.y) + (rw[2].y * d4.z + rw[3].y * d4.w);
          }
          const float qa = pack_halves(c[0], c[1], upper);
          const float qb = qa * dxr;
          const float c2 = c[2] + dpp_f<0x128>(c[2]);
          const float qc = upper ? c2 : qb * dxr;
          const float ka = pack_halves(c[3], c[4], upper);
          const float kb = c[5] + dpp_f<0x128>(c[5]);
          float *t2 = &s_rw[((row * 2 + (upper ? 1 : 0)) * 5) * 8 + jj];
          t2[0] = qa;
          t2[8] = qb;
          t2[16] = qc;
          t2[24] = ka;
          t2[32] = kb;
          // ---- channel sums: the live triples take the second hop TWO at a time (the first pair rides with the base sums)
          auto base_atomic = [&]() {
            constexpr uint64_t HALF_OF_K = 0x000000000000009Aull;  // bit k: upper half (see blend_backward_kernel)
            constexpr uint64_t VAL_OF_K = 0x0000000433021201ull;   // nibble k: which of qa..kb (0..4)
            const int k9 = kcol < 9 ? kcol : 0;
            const int hk = (int)((HALF_OF_K >> k9) & 1u), vk = (int)((VAL_OF_K >> (4 * k9)) & 0xFu);
            const float4 *src8 = reinterpret_cast<const float4 *>(&s_rw[((row * 2 + hk) * 5 + vk) * 8]);
            const float4 lo4 = src8[0], hi4 = src8[1];
            const float vsum = ((lo4.x + lo4.y) + (lo4.z + lo4.w)) + ((hi4.x + hi4.y) + (hi4.z + hi4.w));
            if (row_live && kcol < NACC) atomicAdd(&a.grad_rows[(size_t)gid * GROWX + kcol], vsum);
          };
#pragma unroll
          for (int j0 = 0; j0 < NL; j0 += 2) {
            const int npair = j0 + 1 < NL ? 2 : 1;
#pragma unroll
            for (int jj2 = 0; jj2 < 2; jj2++) {
              if (jj2 < npair) {
                float x3[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                  const float4 d4 = *reinterpret_cast<const float4 *>(&s_dx[(3 + 3 * (j0 + jj2) + k) * WAVE + (int)(lane & 15u) * 4]);
                  x3[k] = (rw[0].y * d4.x + rw[1].y * d4.y) + (rw[2].y * d4.z + rw[3].y * d4.w);
                }
                float *t3 = &s_rw[FX_BASE2 + jj2 * FX_TRI2 + (row * 3) * 16 + (int)(lane & 15u)];
                t3[0] = x3[0];
                t3[16] = x3[1];
                t3[32] = x3[2];
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (j0 == 0) base_atomic();
            {  // lane (row, k < 6): column k % 3 of the pair's triple k / 3
              const int which = kcol >= 3 ? 1 : 0, k3 = kcol < 6 ? kcol - 3 * which : 0;
              const float4 *src16 = reinterpret_cast<const float4 *>(&s_rw[FX_BASE2 + which * FX_TRI2 + (row * 3 + k3) * 16]);
              const float4 v0 = src16[0], v1 = src16[1], v2 = src16[2], v3 = src16[3];
              const float xs = (((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w))) +
                               (((v2.x + v2.y) + (v2.z + v2.w)) + ((v3.x + v3.y) + (v3.z + v3.w)));
              const uint32_t tt = (which && npair == 2) ? live_t[(j0 + 1) % (NL > 0 ? NL : 1)] : live_t[j0 % (NL > 0 ? NL : 1)];
              if (row_live && kcol < 3 * npair) atomicAdd(&a.grad_rows[(size_t)gid * GROWX + NACC + 3 * tt + k3], xs);
            }
            __builtin_amdgcn_wave_barrier();  // the next pair's (or the next group's) stores stay behind these reads
          }
          if (NL == 0) {  // no live triple at all: the base sums' second hop alone
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            base_atomic();
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
      __builtin_amdgcn_wave_barrier();  // keep the next half's LDS writes behind this half's reads
    }
  }
  if (a.trace && lane == 0) {
    unsigned long long *r = a.trace + (size_t)blockIdx.x * 4u;
    r[0] = trace_t0;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = (unsigned long long)(walk_end - skip > 0 ? walk_end - skip : 0) | ((unsigned long long)n << 32);
    r[3] = (unsigned long long)tile | ((unsigned long long)seg << 32);
  }
}


// one kernel per number of live triples, each told how many waves per SIMD its registers should leave room for (the register
// allocator then fits 96 / 128 VGPRs instead of stopping two registers above the step)
// (five and six live triples: the gradient table makes 11.1 / 12.2 KB of LDS = 9 / 10 granules = three waves per SIMD whatever the
// registers do, so the allocator is told three and keeps everything in registers -- asked for four it spilled)
#define GSR_FEATURES_KERNEL(NL, WAVES)                                                                                        \
  __global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void blend_backward_features_kernel_##NL( \
      const BlendBwdArgs a) {                                                                                                 \
    blend_backward_features_body<NL>(a);                                                                                      \
  }
GSR_FEATURES_KERNEL(0, 5)
GSR_FEATURES_KERNEL(1, 5)
GSR_FEATURES_KERNEL(2, 5)
GSR_FEATURES_KERNEL(3, 4)
GSR_FEATURES_KERNEL(4, 4)
GSR_FEATURES_KERNEL(5, 3)
GSR_FEATURES_KERNEL(6, 3)
#undef GSR_FEATURES_KERNEL

int launch_blend_backward(const BlendBwdArgs &a, const Options &opt, hipStream_t stream) {
  const unsigned tiles = (unsigned)(a.grid_x * a.grid_y);
  if (tiles == 0) return GSR_OK;
  const unsigned slots = tile_slots_max(a.grid_x, a.grid_y);  // (the kernels bound themselves by the frame's own mode word)
  if (a.det_rows) {
    if (a.CE != 0) {
      set_error("deterministic backward: only the plain pass (no extra feature channels) is built");
      return GSR_EINVAL;
    }
    hipLaunchKernelGGL((blend_backward_kernel<1, 0, 0, true>), dim3(slots * 4), dim3(WAVE), 0, stream, a);
    return GSR_OK;
  }
  if (a.CE != 0) {
    if (a.CE != CE_MAX || !a.extra) {
      set_error("fused feature blend backward: exactly %d extra channels with their arrays are required", CE_MAX);
      return GSR_EINVAL;
    }
    if (opt.blend_bwd_reduce == 3) {  // reductions through LDS; dynamic LDS = the image-gradient table of the live channels
      const unsigned live = 3u * (unsigned)__builtin_popcount(a.extra_mask & 0x3Fu);
      const dim3 grid(slots * 4), block(WAVE);
      const size_t lds = (live + 3u) * WAVE * sizeof(float);
      switch (live / 3u) {  // (the kernel is built per number of live triples: rows, registers and passes sized by it)
        case 0: hipLaunchKernelGGL(blend_backward_features_kernel_0, grid, block, lds, stream, a); break;
        case 1: hipLaunchKernelGGL(blend_backward_features_kernel_1, grid, block, lds, stream, a); break;
        case 2: hipLaunchKernelGGL(blend_backward_features_kernel_2, grid, block, lds, stream, a); break;
        case 3: hipLaunchKernelGGL(blend_backward_features_kernel_3, grid, block, lds, stream, a); break;
        case 4: hipLaunchKernelGGL(blend_backward_features_kernel_4, grid, block, lds, stream, a); break;
        case 5: hipLaunchKernelGGL(blend_backward_features_kernel_5, grid, block, lds, stream, a); break;
        default: hipLaunchKernelGGL(blend_backward_features_kernel_6, grid, block, lds, stream, a); break;
      }
      return GSR_OK;
    }
    hipLaunchKernelGGL((blend_backward_kernel<1, 0, CE_MAX>), dim3(slots * 4), dim3(WAVE), 0, stream, a);
    return GSR_OK;
  }
  if (opt.blend_bwd_reduce == 3 && opt.blend_bwd_waves == 4) {  // the default
    if (a.loss_gt) hipLaunchKernelGGL(blend_backward_lds_kernel<false>, dim3(slots * 4), dim3(WAVE), 0, stream, a);  // (no depth term)
    else hipLaunchKernelGGL(blend_backward_lds_kernel<true>, dim3(slots * 4), dim3(WAVE), 0, stream, a);
    return GSR_OK;
  }
#ifdef GSR_BUILD_EXPERIMENTS
  if (opt.blend_bwd_reduce == 4 && opt.blend_bwd_waves == 4) {  // round 3's instantiation of the LDS-fold design (A/B against the rewrite)
    hipLaunchKernelGGL((blend_backward_kernel<1, 3, 0>), dim3(slots * 4), dim3(WAVE), 0, stream, a);
    return GSR_OK;
  }
  if (opt.blend_bwd_reduce == 2 && opt.blend_bwd_waves == 4 && !a.loss_gt) {  // (the MFMA experiment reads its image gradients)
    hipLaunchKernelGGL(blend_backward_mfma_kernel, dim3(slots * 4), dim3(WAVE), 0, stream, a);
    return GSR_OK;
  }
  if (opt.blend_bwd_reduce == 1) {
    switch (opt.blend_bwd_waves) {
      case 1: hipLaunchKernelGGL((blend_backward_kernel<4, 1, 0>), dim3(tiles), dim3(WAVE), 0, stream, a); break;
      case 2: hipLaunchKernelGGL((blend_backward_kernel<2, 1, 0>), dim3(tiles * 2), dim3(WAVE), 0, stream, a); break;
      default: hipLaunchKernelGGL((blend_backward_kernel<1, 1, 0>), dim3(slots * 4), dim3(WAVE), 0, stream, a); break;
    }
    return GSR_OK;
  }
#endif
  switch (opt.blend_bwd_waves) {
    case 1: hipLaunchKernelGGL((blend_backward_kernel<4, 0, 0>), dim3(tiles), dim3(WAVE), 0, stream, a); break;
    case 2: hipLaunchKernelGGL((blend_backward_kernel<2, 0, 0>), dim3(tiles * 2), dim3(WAVE), 0, stream, a); break;
    default: hipLaunchKernelGGL((blend_backward_kernel<1, 0, 0>), dim3(slots * 4), dim3(WAVE), 0, stream, a); break;
  }
  return GSR_OK;
}

}  // namespace gsr
