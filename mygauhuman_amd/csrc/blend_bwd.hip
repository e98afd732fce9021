// blend_bwd.hip -- per-pixel backward of the alpha blend (replaces renderCUDA, CR/backward.cu:399-587).
//
// Same tile/quadrant decomposition as blend_fwd.hip (NW waves per 16x16 tile, each lane owns SLOTS = 4/NW pixels,
// one per 8x8 quadrant).  The list is replayed back to front from each pixel's n_contrib / final_T.
//
// Gradient accumulation, MI355X-first: the reference issues nine global atomicAdds per contributing
// (pixel, Gaussian) pair (CR/backward.cu:538,574-584).  Here, per Gaussian and per wave,
//   1. a lane first sums its SLOTS pixels in registers,
//   2. four DPP steps (quad_perm x2, row_ror 4/8) give every 16-lane row its row total -- no LDS, no shuffles,
//   3. lane (row r, k < 9) adds value k of its row into a per-tile LDS accumulator with ONE ds_add_f32
//      wave-instruction (36 active lanes),
//   4. after the batch, the tile flushes the LDS accumulators with float atomics into a packed 64-byte gradient row
//      per Gaussian (16 lanes per row => each memory-side atomic request carries a whole Gaussian), skipping zeros.
// That is one 64-B atomic request per (Gaussian, tile) instance instead of 9 x (pixels hit) scattered atomics.
// A ballot skips all of it when no lane of the wave is hit by the Gaussian.
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

constexpr int NACC = 9;        // mean2D.x, mean2D.y, conic.x, conic.y, conic.w, opacity, r, g, b
constexpr int ACC_STRIDE = 9;  // floats per Gaussian in the LDS accumulator (odd stride: conflict-free rows)

template <int NW>
__global__ __launch_bounds__(WAVE *NW) void blend_backward_kernel(const BlendBwdArgs a) {
  constexpr int SLOTS = 4 / NW;
  constexpr int BATCH = WAVE * NW;
  __shared__ float4 s0[BATCH];
  __shared__ float4 s1[BATCH];
  __shared__ float4 s2[BATCH];
  __shared__ uint32_t s_id[BATCH];
  __shared__ float s_acc[BATCH * ACC_STRIDE];
  __shared__ uint32_t s_max[NW];

  const uint32_t tile = xcd_remap_b(blockIdx.x, gridDim.x);
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const int wave = threadIdx.x / WAVE;
  const uint32_t lane = lane_id();
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);
  const size_t plane = (size_t)a.H * a.W;
  const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;

  float pxf[SLOTS], pyf[SLOTS], T[SLOTS], Tfin[SLOTS], bgdot[SLOTS];
  float dpix0[SLOTS], dpix1[SLOTS], dpix2[SLOTS], ddep[SLOTS], dalp[SLOTS];
  float arec0[SLOTS], arec1[SLOTS], arec2[SLOTS], adep[SLOTS], aalp[SLOTS];
  float lalpha[SLOTS], lc0[SLOTS], lc1[SLOTS], lc2[SLOTS], ldep[SLOTS];
  int lastc[SLOTS];
  uint32_t maxlast = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = wave * SLOTS + s;
    const int px = tx * TILE + (q & 1) * 8 + (int)(lane & 7);
    const int py = ty * TILE + (q >> 1) * 8 + (int)(lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const int p = py * a.W + px;
    pxf[s] = (float)px;
    pyf[s] = (float)py;
    Tfin[s] = inside ? a.final_T[p] : 0.f;
    T[s] = Tfin[s];
    lastc[s] = inside ? (int)a.n_contrib[p] : 0;
    dpix0[s] = inside ? a.dL_dpix[p] : 0.f;
    dpix1[s] = inside ? a.dL_dpix[plane + p] : 0.f;
    dpix2[s] = inside ? a.dL_dpix[2 * plane + p] : 0.f;
    ddep[s] = inside ? a.dL_ddepth[p] : 0.f;
    dalp[s] = inside ? a.dL_dalpha[p] : 0.f;
    bgdot[s] = a.bg[0] * dpix0[s] + a.bg[1] * dpix1[s] + a.bg[2] * dpix2[s];
    arec0[s] = arec1[s] = arec2[s] = adep[s] = aalp[s] = 0.f;
    lalpha[s] = lc0[s] = lc1[s] = lc2[s] = ldep[s] = 0.f;
    maxlast = max(maxlast, (uint32_t)lastc[s]);
  }
  // block-wide max of n_contrib: list entries at front positions >= maxlast contribute to no pixel of the tile
  {
    uint32_t m = maxlast;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, WAVE));
    if (lane == 0) s_max[wave] = m;
    for (int e = threadIdx.x; e < BATCH * ACC_STRIDE; e += BATCH) s_acc[e] = 0.f;
    __syncthreads();
    m = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) m = max(m, s_max[w]);
    maxlast = m;
  }
  const int skip = n - (int)maxlast;  // entries idx < skip (counted from the back) are behind every last contributor

  // per-lane constants for step 3 (which accumulator column this lane feeds)
  const int kcol = (int)(lane & 15);

  for (int base = skip; base < n; base += BATCH) {
    const int idx = base + (int)threadIdx.x;
    if (idx < n) {
      const uint32_t id = a.point_list[range.y - 1 - idx];
      const float4 *src = reinterpret_cast<const float4 *>(a.recs + id);
      s_id[threadIdx.x] = id;
      s0[threadIdx.x] = src[0];
      s1[threadIdx.x] = src[1];
      s2[threadIdx.x] = src[2];
    }
    __syncthreads();
    const int cnt = min(BATCH, n - base);
    for (int j = 0; j < cnt; j++) {
      const int fpos = n - 1 - (base + j);  // 0-based position from the front (= contributor after the decrement)
      const float4 g0 = s0[j];
      const float4 g1 = s1[j];
      const float4 g2 = s2[j];
      float acc[NACC];
#pragma unroll
      for (int k = 0; k < NACC; k++) acc[k] = 0.f;
      bool any = false;
#pragma unroll
      for (int s = 0; s < SLOTS; s++) {
        const float dx = g0.x - pxf[s], dy = g0.y - pyf[s];
        const float power = -0.5f * (g0.z * dx * dx + g1.x * dy * dy) - g0.w * dx * dy;
        const float G = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
        const float alpha = fminf(0.99f, g1.y * G);
        const bool hit = (fpos < lastc[s]) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        if (__ballot(hit) != 0ull) {
          any = true;
          const float rc = __builtin_amdgcn_rcpf(1.f - alpha);
          const float Tn = T[s] * rc;
          const float w = alpha * Tn;  // dchannel_dcolor
          const float one_m_la = 1.f - lalpha[s];
          const float r0 = lalpha[s] * lc0[s] + one_m_la * arec0[s];
          const float r1 = lalpha[s] * lc1[s] + one_m_la * arec1[s];
          const float r2 = lalpha[s] * lc2[s] + one_m_la * arec2[s];
          const float rd = lalpha[s] * ldep[s] + one_m_la * adep[s];
          const float ra = lalpha[s] + one_m_la * aalp[s];
          float dL_dopa = (g1.w - r0) * dpix0[s] + (g2.x - r1) * dpix1[s] + (g2.y - r2) * dpix2[s];
          dL_dopa += (g1.z - rd) * ddep[s];
          dL_dopa += (1.f - ra) * dalp[s];
          dL_dopa *= Tn;
          dL_dopa += (-Tfin[s] * rc) * bgdot[s];
          const float dL_dG = g1.y * dL_dopa;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * g0.z - gdy * g0.w;
          const float dG_ddely = -gdy * g1.x - gdx * g0.w;
          const float m = hit ? 1.f : 0.f;
          const float hG = m * dL_dG;
          acc[0] += hG * dG_ddelx * ddelx_dx;
          acc[1] += hG * dG_ddely * ddely_dy;
          acc[2] += -0.5f * gdx * dx * hG;
          acc[3] += -0.5f * gdx * dy * hG;
          acc[4] += -0.5f * gdy * dy * hG;
          acc[5] += m * G * dL_dopa;
          const float hw = m * w;
          acc[6] += hw * dpix0[s];
          acc[7] += hw * dpix1[s];
          acc[8] += hw * dpix2[s];
          // commit the replay state for the lanes that were hit
          T[s] = hit ? Tn : T[s];
          arec0[s] = hit ? r0 : arec0[s];
          arec1[s] = hit ? r1 : arec1[s];
          arec2[s] = hit ? r2 : arec2[s];
          adep[s] = hit ? rd : adep[s];
          aalp[s] = hit ? ra : aalp[s];
          lalpha[s] = hit ? alpha : lalpha[s];
          lc0[s] = hit ? g1.w : lc0[s];
          lc1[s] = hit ? g2.x : lc1[s];
          lc2[s] = hit ? g2.y : lc2[s];
          ldep[s] = hit ? g1.z : ldep[s];
        }
      }
      if (any) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < NACC; k++) acc[k] = row16_sum(acc[k]);
        float v = acc[0];
#pragma unroll
        for (int k = 1; k < NACC; k++) v = (kcol == k) ? acc[k] : v;
        if (kcol < NACC) atomicAdd(&s_acc[j * ACC_STRIDE + kcol], v);
      }
    }
    __syncthreads();
    // flush: 16 lanes per Gaussian row -> one 64-byte line of grad_rows per (Gaussian, tile) instance
    for (int e = threadIdx.x; e < cnt * 16; e += BATCH) {
      const int j = e >> 4, k = e & 15;
      if (k < NACC) {
        const float v = s_acc[j * ACC_STRIDE + k];
        if (v != 0.f) {
          atomicAdd(&a.grad_rows[(size_t)s_id[j] * GROW + k], v);
          s_acc[j * ACC_STRIDE + k] = 0.f;
        }
      }
    }
    __syncthreads();
  }
}

static int g_blend_bwd_nw = 2;
int set_blend_backward_waves(int nw) {
  if (nw != 1 && nw != 2 && nw != 4) return GSR_EINVAL;
  g_blend_bwd_nw = nw;
  return GSR_OK;
}

int launch_blend_backward(const BlendBwdArgs &a, hipStream_t stream) {
  const unsigned tiles = (unsigned)(a.grid_x * a.grid_y);
  if (tiles == 0) return GSR_OK;
  switch (g_blend_bwd_nw) {
    case 1: hipLaunchKernelGGL(blend_backward_kernel<1>, dim3(tiles), dim3(WAVE * 1), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(blend_backward_kernel<2>, dim3(tiles), dim3(WAVE * 2), 0, stream, a); break;
    default: hipLaunchKernelGGL(blend_backward_kernel<4>, dim3(tiles), dim3(WAVE * 4), 0, stream, a); break;
  }
  return GSR_OK;
}

}  // namespace gsr
