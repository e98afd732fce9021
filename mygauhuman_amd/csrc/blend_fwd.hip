// blend_fwd.hip -- per-tile front-to-back alpha blending (replaces renderCUDA, CR/forward.cu:261-383).
//
// Work decomposition (wave64-first, not a 16x16-thread CUDA block):
//   * a 16x16 tile is four 8x8 quadrants; a wave's 64 lanes cover one quadrant (lane = (y&7)*8 + (x&7)), so the
//     footprint a wave tests against a Gaussian is a compact 8x8 square and a whole-wave miss is common;
//   * NW waves work on a tile (template), each owning SLOTS = 4/NW quadrants in registers: NW=4 is one pixel per
//     lane, NW=1 is one wave per tile with four pixels per lane (no block barrier at all);
//   * the tile's depth-sorted instance list is staged through LDS in batches of 64*NW 48-byte SplatRec records
//     (three float4 planes; the per-Gaussian reads in the inner loop are single-address LDS broadcasts);
//   * wave64 ballots skip the blend arithmetic for a quadrant when no lane is hit, and retire a wave / the block
//     once every pixel is saturated (the reference uses __syncthreads_count, CR/forward.cu:314).
// Results follow the reference's rules exactly: skip power > 0, alpha = min(0.99, o*exp(power)), skip alpha < 1/255,
// stop (without blending) when T*(1-alpha) < 1e-4, alpha image = sum of weights, depth image = sum z*w.
#include "gsr_common.h"

namespace gsr {

// bijective XCD-aware remap: consecutive tiles (which share Gaussians) land on the same XCD / L2
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t n) {
  const uint32_t q = n / 8, r = n % 8, xcd = bid % 8, k = bid / 8;
  const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

template <int NW>
__global__ __launch_bounds__(WAVE *NW) void blend_forward_kernel(const BlendFwdArgs a) {
  constexpr int SLOTS = 4 / NW;
  constexpr int BATCH = WAVE * NW;
  __shared__ float4 s0[BATCH];  // x, y, conic_a, conic_b
  __shared__ float4 s1[BATCH];  // conic_c, opacity, depth, r
  __shared__ float4 s2[BATCH];  // g, b, -, -

  const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = tile % a.grid_x, ty = tile / a.grid_x;
  const int wave = threadIdx.x / WAVE;
  const uint32_t lane = lane_id();
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);

  float pxf[SLOTS], pyf[SLOTS], T[SLOTS], C0[SLOTS], C1[SLOTS], C2[SLOTS], Dp[SLOTS], Wt[SLOTS];
  uint32_t last[SLOTS];
  bool inside[SLOTS], done[SLOTS];
  int pixid[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = wave * SLOTS + s;
    const int px = tx * TILE + (q & 1) * 8 + (int)(lane & 7);
    const int py = ty * TILE + (q >> 1) * 8 + (int)(lane >> 3);
    pxf[s] = (float)px;
    pyf[s] = (float)py;
    inside[s] = px < a.W && py < a.H;
    done[s] = !inside[s];
    pixid[s] = py * a.W + px;
    T[s] = 1.0f;
    C0[s] = C1[s] = C2[s] = Dp[s] = Wt[s] = 0.f;
    last[s] = 0;
  }

  for (int base = 0; base < n; base += BATCH) {
    bool all_done = true;
#pragma unroll
    for (int s = 0; s < SLOTS; s++) all_done = all_done && done[s];
    const bool wave_done = __ballot(!all_done) == 0ull;
    if (NW == 1) {
      if (wave_done) break;
      __syncthreads();
    } else {
      if (__syncthreads_and(all_done ? 1 : 0)) break;
    }
    const int idx = base + (int)threadIdx.x;
    if (idx < n) {
      const uint32_t id = a.point_list[range.x + idx];
      const float4 *src = reinterpret_cast<const float4 *>(a.recs + id);
      s0[threadIdx.x] = src[0];
      s1[threadIdx.x] = src[1];
      s2[threadIdx.x] = src[2];
    }
    __syncthreads();
    if (wave_done) continue;
    const int cnt = min(BATCH, n - base);
    for (int j = 0; j < cnt; j++) {
      const float4 g0 = s0[j];
      const float4 g1 = s1[j];
#pragma unroll
      for (int s = 0; s < SLOTS; s++) {
        const float dx = g0.x - pxf[s], dy = g0.y - pyf[s];
        const float power = -0.5f * (g0.z * dx * dx + g1.x * dy * dy) - g0.w * dx * dy;
        const float alpha = fminf(0.99f, g1.y * __builtin_amdgcn_exp2f(power * 1.4426950408889634f));
        const bool hit = !done[s] && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        if (__ballot(hit) != 0ull) {
          const float test_T = T[s] * (1.0f - alpha);
          const bool stop = hit && test_T < 0.0001f;
          const bool blend = hit && !stop;
          done[s] = done[s] || stop;
          const float w = blend ? alpha * T[s] : 0.0f;
          const float4 g2 = s2[j];
          C0[s] += g1.w * w;
          C1[s] += g2.x * w;
          C2[s] += g2.y * w;
          Dp[s] += g1.z * w;
          Wt[s] += w;
          T[s] = blend ? test_T : T[s];
          last[s] = blend ? (uint32_t)(base + j + 1) : last[s];
        }
      }
    }
  }

  const size_t plane = (size_t)a.H * a.W;
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    if (inside[s]) {
      const int p = pixid[s];
      a.final_T[p] = T[s];
      a.n_contrib[p] = last[s];
      a.out_color[p] = C0[s] + T[s] * a.bg[0];
      a.out_color[plane + p] = C1[s] + T[s] * a.bg[1];
      a.out_color[2 * plane + p] = C2[s] + T[s] * a.bg[2];
      a.out_alpha[p] = Wt[s];  // CR/forward.cu:380
      a.out_depth[p] = Dp[s];
    }
  }
}

static int g_blend_fwd_nw = 2;
int set_blend_forward_waves(int nw) {
  if (nw != 1 && nw != 2 && nw != 4) return GSR_EINVAL;
  g_blend_fwd_nw = nw;
  return GSR_OK;
}

int launch_blend_forward(const BlendFwdArgs &a, hipStream_t stream) {
  const unsigned tiles = (unsigned)(a.grid_x * a.grid_y);
  if (tiles == 0) return GSR_OK;
  switch (g_blend_fwd_nw) {
    case 1: hipLaunchKernelGGL(blend_forward_kernel<1>, dim3(tiles), dim3(WAVE * 1), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(blend_forward_kernel<2>, dim3(tiles), dim3(WAVE * 2), 0, stream, a); break;
    default: hipLaunchKernelGGL(blend_forward_kernel<4>, dim3(tiles), dim3(WAVE * 4), 0, stream, a); break;
  }
  return GSR_OK;
}

}  // namespace gsr
