// mlp.hip -- the per-Gaussian skinning-weight offset network of render() (nets/mlp_delta_weight_lbs.py:5-32, run every frame when
// motion_offset_flag is set: gaussian_renderer/__init__.py:100-106) as ONE kernel on the matrix cores, forward pass.
//
//   emb[63]  = (x, sin(2^o x), cos(2^o x), o = 0..9)                                   (get_embedder(10), :34-77)
//   h1 = relu(W0 emb + b0)   h2 = relu(W1 h1 + b1)   h3 = relu(W2 h2 + b2)   h4 = relu(W3 [emb; h3] + b3)   out[24] = Wfc h4 + bfc
// In plain torch this is five skinny fp32 GEMMs over 200k points plus a dozen elementwise kernels on 100-MB activations: 1.4 ms of a
// forward whose rasterizer takes 0.25.  Here the activations never leave the registers:
//   * a wave owns 32 points; an activation tile lives as the C/D fragment of v_mfma_f32_32x32x2_f32 -- column (point) on the lane,
//     32 feature rows in 16 registers (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5));
//   * the next layer computes  Y[o][p] = sum_k Wt[o][k] X[k][p]  with the accumulator REGISTER i of X as the B operand itself: lanes
//     0..31 hold row r0(i), lanes 32..63 row r0(i) + 4 -- the two k of one 32x32x2 step -- so no conversion, no lane movement, no LDS
//     for the activations; the weight matrix is packed once per call into the matching A fragments (gsr_lbs_offset_mlp_pack);
//   * f32 in, f32 accumulate: the result is a k-ordered fmaf chain, torch's own accuracy (MI355X has no reduced-precision f32 path);
//   * a workgroup = four waves (one per SIMD: this instruction reaches its issue rate from one wave), 128 points; the packed weights
//     of the current layer sit in LDS (<= 96 KB), one ds_read_b128 per k-step feeds the four output tiles.
// Arithmetic: 137 kFLOP per point = 27 GFLOP at 200k points = 16,384 MFMA cycles per 128 x 128 layer per wave.
#include <math.h>
#include <string.h>

#include <atomic>

#include "gsr_common.h"

namespace gsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MLP_WG = 512;            // threads per workgroup of the forward / chain kernels: eight waves (two per SIMD: the second hides
                                       // the first's LDS reads and the barriers around the weight staging), 32 points each
constexpr int MLP_WG_POINTS = MLP_WG / 2;

constexpr int MLP_LAYERS = 5;
constexpr int MLP_E = 63, MLP_W = 128, MLP_OUT = 24;
// per layer: input tiles of 32 features, output tiles of 32 features
__host__ __device__ constexpr int mlp_tin(int l) { return l == 0 ? 2 : (l == 3 ? 6 : 4); }
__host__ __device__ constexpr int mlp_tout(int l) { return l == 4 ? 1 : 4; }
// packed A fragments of a layer: [input tile][accumulator register 0..15][lane 0..63][output tile]
__host__ __device__ constexpr int mlp_packed_floats(int l) { return mlp_tin(l) * 16 * 64 * mlp_tout(l); }
__host__ __device__ constexpr int mlp_packed_offset(int l) {
  int o = 0;
  for (int k = 0; k < l; k++) o += mlp_packed_floats(k);
  return o;
}
constexpr int MLP_PACKED_W = mlp_packed_offset(MLP_LAYERS);     // floats of all packed weights
constexpr int MLP_PACKED_B = 4 * MLP_W + 32;                      // biases, the last layer padded to 32
constexpr int MLP_PACKED = MLP_PACKED_W + MLP_PACKED_B;
constexpr int MLP_LDS_FLOATS = mlp_packed_floats(3);              // the largest layer: 6 x 16 x 64 x 4 floats = 96 KB
// backward (dh = W^T dZ) fragments, steps 0..3 = layers fc, 3, 2, 1: [o tile of dZ][register][lane][k tile of dh]; layer 0 has no
// input gradient (the positions are detached: gaussian_renderer/__init__.py:104) and layer 3 only its h part
__host__ __device__ constexpr int mlp_bwd_tin(int s) { return s == 0 ? 1 : 4; }
__host__ __device__ constexpr int mlp_bwd_floats(int s) { return mlp_bwd_tin(s) * 16 * 64 * 4; }
__host__ __device__ constexpr int mlp_bwd_offset(int s) {
  int o = MLP_PACKED;
  for (int k = 0; k < s; k++) o += mlp_bwd_floats(k);
  return o;
}
constexpr int MLP_PACKED_ALL = mlp_bwd_offset(4);
// bf16 fragments ("bf16x3": every weight as two bf16 terms, hi + lo), the same float counts and offsets as the f32 fragments:
// 16-byte units [input tile][k-step 0..1][output tile][hi | lo][lane], eight bf16 each
constexpr int MLP_PACKED_FWD16 = MLP_PACKED_ALL;                                  // bf16 forward fragments
constexpr int MLP_PACKED_BWD16 = MLP_PACKED_FWD16 + MLP_PACKED_W;                 // bf16 backward fragments (steps fc, 3, 2, 1)
constexpr int MLP_PACKED_TOTAL = MLP_PACKED_BWD16 + (MLP_PACKED_ALL - MLP_PACKED);

__host__ __device__ constexpr int mlp_row_of_reg(int i) { return (i & 3) + 8 * (i >> 2); }

__host__ __device__ __forceinline__ uint32_t bf16_rne(float x) {   // round to nearest even (finite inputs)
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

struct MlpWeights {   // the reference module's tensors (Conv1d weight [out][in][1] = row-major [out][in])
  const float *w[MLP_LAYERS];
  const float *b[MLP_LAYERS];
};

// one thread per packed float
__global__ __launch_bounds__(256) void mlp_pack_kernel(const MlpWeights src, float *packed) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= MLP_PACKED_TOTAL) return;
  if (e >= MLP_PACKED_BWD16) {   // backward fragments in bf16: W_l[o = the k of the step][column of input feature kf]
    const int r0 = e - MLP_PACKED_BWD16 + MLP_PACKED;
    int st = 0, base = MLP_PACKED;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (r0 >= mlp_bwd_offset(k)) {
        st = k;
        base = mlp_bwd_offset(k);
      }
    }
    const int l = 4 - st, r = r0 - base, unit = r / 4, sub = r % 4;
    const int lane = unit % 64, hl = (unit / 64) % 2, t_out = (unit / 128) % 4, ks = (unit / 512) % 2, t_in = unit / 1024;
    const int kf = 32 * t_out + (lane & 31), ncols = l == 3 ? MLP_E + MLP_W : MLP_W, col = l == 3 ? MLP_E + kf : kf;
    const int nrows = l == 4 ? MLP_OUT : MLP_W;
    uint32_t word = 0;
    for (int q = 0; q < 2; q++) {
      const int j = 2 * sub + q;
      const int o = 32 * t_in + 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
      const float w = o < nrows ? src.w[l][(size_t)o * ncols + col] : 0.f;
      const uint32_t hi = bf16_rne(w), lo = bf16_rne(w - __uint_as_float(hi << 16));
      word |= (hl ? lo : hi) << (16 * q);
    }
    reinterpret_cast<uint32_t *>(packed)[e] = word;
    return;
  }
  if (e >= MLP_PACKED_ALL) {   // two bf16 elements (2 sub, 2 sub + 1) of a 16-byte fragment unit
    const int r0 = e - MLP_PACKED_ALL;
    int l = 0, base = 0;
#pragma unroll
    for (int k = 0; k < MLP_LAYERS; k++) {
      if (r0 >= mlp_packed_offset(k)) {
        l = k;
        base = mlp_packed_offset(k);
      }
    }
    const int nto = mlp_tout(l), r = r0 - base, unit = r / 4, sub = r % 4;
    const int lane = unit % 64, hl = (unit / 64) % 2, t_out = (unit / 128) % nto, ks = (unit / 128 / nto) % 2, t_in = unit / 128 / nto / 2;
    const int o = 32 * t_out + (lane & 31), nrows = l == 4 ? MLP_OUT : MLP_W;
    uint32_t word = 0;
    for (int q = 0; q < 2; q++) {
      const int j = 2 * sub + q;
      const int k = 32 * t_in + 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
      int col, ncols;
      if (l == 0) {
        ncols = MLP_E, col = k < MLP_E ? k : -1;
      } else if (l == 3) {
        ncols = MLP_E + MLP_W, col = k < MLP_E ? k : (k == MLP_E ? -1 : MLP_E + (k - 64));
      } else {
        ncols = MLP_W, col = k;
      }
      const float w = (col >= 0 && o < nrows) ? src.w[l][(size_t)o * ncols + col] : 0.f;
      const uint32_t hi = bf16_rne(w), lo = bf16_rne(w - __uint_as_float(hi << 16));
      word |= (hl ? lo : hi) << (16 * q);
    }
    reinterpret_cast<uint32_t *>(packed)[e] = word;
    return;
  }
  if (e >= MLP_PACKED) {   // backward fragments: W_l[o][column of input feature k]
    int st = 0, base = MLP_PACKED;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (e >= mlp_bwd_offset(k)) {
        st = k;
        base = mlp_bwd_offset(k);
      }
    }
    const int l = 4 - st, r = e - base;
    const int t_out = r % 4, lane = (r / 4) % 64, i = (r / 4 / 64) % 16, t_in = r / 4 / 64 / 16;
    const int o = 32 * t_in + mlp_row_of_reg(i) + 4 * (lane >> 5), kf = 32 * t_out + (lane & 31);
    const int ncols = l == 3 ? MLP_E + MLP_W : MLP_W, col = l == 3 ? MLP_E + kf : kf, nrows = l == 4 ? MLP_OUT : MLP_W;
    packed[e] = o < nrows ? src.w[l][(size_t)o * ncols + col] : 0.f;
    return;
  }
  if (e >= MLP_PACKED_W) {
    const int k = e - MLP_PACKED_W, l = k / MLP_W, o = k % MLP_W;
    packed[e] = l < 4 ? src.b[l][o] : (o < MLP_OUT ? src.b[4][o] : 0.f);
    return;
  }
  int l = 0, base = 0;
#pragma unroll
  for (int k = 0; k < MLP_LAYERS; k++) {
    if (e >= mlp_packed_offset(k)) {
      l = k;
      base = mlp_packed_offset(k);
    }
  }
  const int nto = mlp_tout(l), r = e - base;
  const int t_out = r % nto, lane = (r / nto) % 64, i = (r / nto / 64) % 16, t_in = r / nto / 64 / 16;
  const int o = 32 * t_out + (lane & 31);
  const int k = 32 * t_in + mlp_row_of_reg(i) + 4 * (lane >> 5);
  // column of the reference's matrix for input feature k of this layer (-1: a padding feature)
  int col, ncols, nrows = MLP_W;
  if (l == 0) {
    ncols = MLP_E, col = k < MLP_E ? k : -1;
  } else if (l == 3) {
    ncols = MLP_E + MLP_W, col = k < MLP_E ? k : (k == MLP_E ? -1 : MLP_E + (k - 64));
  } else {
    ncols = MLP_W, col = k;
  }
  if (l == 4) nrows = MLP_OUT;
  packed[e] = (col >= 0 && o < nrows) ? src.w[l][(size_t)o * ncols + col] : 0.f;
}

// embedding feature k (0..63; 63 = padding) of a point.  FAST (the bf16 kernels): sin / cos on the hardware instruction, which takes
// REVOLUTIONS: the argument x 2^o is exact, its product with 1 / (2 pi) is formed in two terms (error ~1e-8 revolutions before the
// fraction is taken), v_sin_f32 / v_cos_f32 add ~1e-6 -- a tenth of what that path's bf16 terms drop -- at a tenth of libm's ~120
// instructions per value (32 values per lane were a quarter of the forward's time).  The f32 kernels keep sinf / cosf.
template <bool FAST>
__device__ __forceinline__ float mlp_embed(int k, float x, float y, float z) {
  if (k >= MLP_E) return 0.f;
  if (k < 3) return k == 0 ? x : (k == 1 ? y : z);
  const int t = k - 3, oct = t / 6, r = t % 6, c = r % 3;
  const float ang = (c == 0 ? x : (c == 1 ? y : z)) * (float)(1 << oct);
  if constexpr (FAST) {
    constexpr float INV2PI_HI = 0.15915494f, INV2PI_LO = (float)(0.15915494309189535 - (double)0.15915494f);
    const float p_hi = ang * INV2PI_HI;
    const float p_lo = __builtin_fmaf(ang, INV2PI_HI, -p_hi) + ang * INV2PI_LO;
    const float rev = (p_hi - floorf(p_hi)) + p_lo;
    return r < 3 ? __builtin_amdgcn_sinf(rev) : __builtin_amdgcn_cosf(rev);
  } else {
    return r < 3 ? sinf(ang) : cosf(ang);
  }
}

// the embedding as two activation tiles (features 0..31, 32..63); a template recursion: left as a loop the compiler does not unroll
// the 32 inlined sin / cos evaluations and puts the tile array into scratch
template <bool FAST, int K = 0>
__device__ __forceinline__ void mlp_embed_tiles(f32x16 (&emb)[2], int half, float x, float y, float z) {
  if constexpr (K < 32) {
    emb[K / 16][K % 16] = mlp_embed<FAST>(32 * (K / 16) + mlp_row_of_reg(K % 16) + 4 * half, x, y, z);
    mlp_embed_tiles<FAST, K + 1>(emb, half, x, y, z);
  }
}

// out[to] += A-fragments(s_w) x in[ti]: NTI input tiles (32 features each) -> NTO output tiles, 16 k-steps per input tile
template <int NTI, int NTO>
__device__ __forceinline__ void mlp_mm(const float *s_w, const f32x16 (&in)[NTI], f32x16 (&out)[NTO], uint32_t lane) {
  static_assert(NTO == 4 || NTO == 1, "four output tiles (one 16-byte fragment read per k-step) or one");
#pragma unroll
  for (int ti = 0; ti < NTI; ti++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if constexpr (NTO == 4) {
        const float4 a = *reinterpret_cast<const float4 *>(&s_w[((ti * 16 + i) * 64 + lane) * 4]);
        out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, in[ti][i], out[0], 0, 0, 0);
        out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, in[ti][i], out[1], 0, 0, 0);
        out[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, in[ti][i], out[2], 0, 0, 0);
        out[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, in[ti][i], out[3], 0, 0, 0);
      } else {
        const float a = s_w[(ti * 16 + i) * 64 + lane];
        out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[ti][i], out[0], 0, 0, 0);
      }
    }
  }
}

template <int L, int NIN>
__device__ __forceinline__ void mlp_layer(const float *s_w, const float *bias, const f32x16 (&in)[NIN], f32x16 (&out)[mlp_tout(L)],
                                          uint32_t lane) {
  constexpr int NTO = mlp_tout(L);
  static_assert(NIN == mlp_tin(L), "input tiles of the layer");
  const uint32_t half = lane >> 5;
#pragma unroll
  for (int t = 0; t < NTO; t++) {
#pragma unroll
    for (int i = 0; i < 16; i++) out[t][i] = bias[32 * t + mlp_row_of_reg(i) + 4 * half];
  }
  mlp_mm<NIN, NTO>(s_w, in, out, lane);
}

__device__ __forceinline__ void mlp_relu(f32x16 (&t)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int i = 0; i < 16; i++) t[k][i] = fmaxf(t[k][i], 0.f);
  }
}

// all threads of the workgroup: a packed fragment block -> LDS (16-byte copies); barriers on both sides by the caller
template <int OFFSET, int FLOATS>
__device__ __forceinline__ void mlp_stage_block(const float *packed, float *s_w) {
  const float4 *src = reinterpret_cast<const float4 *>(packed + OFFSET);
  float4 *dst = reinterpret_cast<float4 *>(s_w);
  for (int q = threadIdx.x; q < FLOATS / 4; q += MLP_WG) dst[q] = src[q];
}
// the same copy in two halves: the loads of the NEXT fragment block are issued before the current layer's MFMAs and land in LDS after
// them (between the two barriers that used to enclose an exposed global round trip per layer: a quarter of the forward's time)
constexpr int MLP_PRE = 3;   // x 16 floats per thread: the largest block is 24,576 floats / 512 threads = 48 floats
// (kept as three 16-float vectors, like the activation tiles: a float4[12] array is left in scratch by the compiler)
template <int OFFSET, int FLOATS, int NP>
__device__ __forceinline__ void mlp_prefetch(const float *packed, f32x16 (&pre)[NP]) {
  static_assert(FLOATS % (4 * MLP_WG) == 0 && FLOATS / (4 * MLP_WG) <= 4 * NP, "whole 16-byte pieces per thread");
  const float4 *src = reinterpret_cast<const float4 *>(packed + OFFSET);
#pragma unroll
  for (int j = 0; j < FLOATS / (4 * MLP_WG); j++) {
    const float4 t = src[(int)threadIdx.x + j * MLP_WG];
    pre[j / 4][4 * (j % 4) + 0] = t.x;
    pre[j / 4][4 * (j % 4) + 1] = t.y;
    pre[j / 4][4 * (j % 4) + 2] = t.z;
    pre[j / 4][4 * (j % 4) + 3] = t.w;
  }
}
template <int FLOATS, int NP>
__device__ __forceinline__ void mlp_commit(const f32x16 (&pre)[NP], float *s_w) {
  float4 *dst = reinterpret_cast<float4 *>(s_w);
#pragma unroll
  for (int j = 0; j < FLOATS / (4 * MLP_WG); j++)
    dst[(int)threadIdx.x + j * MLP_WG] = make_float4(pre[j / 4][4 * (j % 4)], pre[j / 4][4 * (j % 4) + 1], pre[j / 4][4 * (j % 4) + 2],
                                                     pre[j / 4][4 * (j % 4) + 3]);
}

template <int L>
__device__ __forceinline__ void mlp_stage(const float *packed, float *s_w) {
  mlp_stage_block<mlp_packed_offset(L), mlp_packed_floats(L)>(packed, s_w);
}

__global__ __launch_bounds__(MLP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_forward_kernel(int P, const float *xyz, const float *packed,
                                                                                                   float *out) {
  extern __shared__ __attribute__((aligned(16))) float s_mlp[];   // MLP_LDS_FLOATS weights + MLP_PACKED_B biases
  float *s_w = s_mlp, *s_b = s_mlp + MLP_LDS_FLOATS;
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE, half = lane >> 5;
  const int p = (int)(blockIdx.x * (uint32_t)MLP_WG_POINTS + wave * 32u + (lane & 31u));
  float x = 0.f, y = 0.f, z = 0.f;
  if (p < P) {
    x = xyz[(size_t)p * 3 + 0];
    y = xyz[(size_t)p * 3 + 1];
    z = xyz[(size_t)p * 3 + 2];
  }
  for (int q = threadIdx.x; q < MLP_PACKED_B; q += MLP_WG) s_b[q] = packed[MLP_PACKED_W + q];
  mlp_stage<0>(packed, s_w);
  f32x16 emb[2];
  mlp_embed_tiles<false>(emb, (int)half, x, y, z);
  __syncthreads();
  f32x16 a[4], b[4];
  mlp_layer<0, 2>(s_w, s_b, emb, a, lane);
  mlp_relu(a);
  __syncthreads();
  mlp_stage<1>(packed, s_w);
  __syncthreads();
  mlp_layer<1, 4>(s_w, s_b + MLP_W, a, b, lane);
  mlp_relu(b);
  __syncthreads();
  mlp_stage<2>(packed, s_w);
  __syncthreads();
  mlp_layer<2, 4>(s_w, s_b + 2 * MLP_W, b, a, lane);
  mlp_relu(a);
  __syncthreads();
  mlp_stage<3>(packed, s_w);
  __syncthreads();
  {
    f32x16 cat[6] = {emb[0], emb[1], a[0], a[1], a[2], a[3]};   // cat((features, net), dim=1): nets/mlp_delta_weight_lbs.py:28-29
    mlp_layer<3, 6>(s_w, s_b + 3 * MLP_W, cat, b, lane);
  }
  mlp_relu(b);
  __syncthreads();
  mlp_stage<4>(packed, s_w);
  __syncthreads();
  f32x16 o[1];
  mlp_layer<4, 4>(s_w, s_b + 4 * MLP_W, b, o, lane);
  if (p < P) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = mlp_row_of_reg(i) + 4 * (int)half;
      if (row < MLP_OUT) out[(size_t)p * MLP_OUT + row] = o[0][i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// EXPERIMENT "bf16x3": the same forward on v_mfma_f32_32x32x16_bf16 (8 passes for K = 16 against 16 passes for K = 2 of the f32
// instruction: 16 x the rate) with f32-like accuracy from splitting BOTH operands in two bf16 terms, x = hi + lo, and three products
// per step (hi hi + hi lo + lo hi; the dropped lo lo is 2^-16 of the product).  An activation tile converts in place: registers
// 8 s .. 8 s + 7 of the accumulator ARE the B fragment of k-step s (their rows 16 s + 8 (j >> 2) + 4 half + (j & 3), j = 0..7; the
// weights are packed in that k order).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mlp_split8(const f32x16 &t, int s, uint4 &hi, uint4 &lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float x0 = t[8 * s + 2 * q], x1 = t[8 * s + 2 * q + 1];
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[q]) : "v"(x0), "v"(x1));
    const float r0 = x0 - __uint_as_float(h[q] << 16), r1 = x1 - __uint_as_float(h[q] & 0xFFFF0000u);
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l[q]) : "v"(r0), "v"(r1));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
template <int NT>
__device__ __forceinline__ void mlp_split_tiles(const f32x16 (&t)[NT], uint4 (&fh)[NT][2], uint4 (&fl)[NT][2]) {
#pragma unroll
  for (int k = 0; k < NT; k++) {
    mlp_split8(t[k], 0, fh[k][0], fl[k][0]);
    mlp_split8(t[k], 1, fh[k][1], fl[k][1]);
  }
}

template <int NTI, int NTO>
__device__ __forceinline__ void mlp_mm_bf16(const uint4 *s_w, const uint4 (&fh)[NTI][2], const uint4 (&fl)[NTI][2], f32x16 (&out)[NTO],
                                            uint32_t lane) {
#pragma unroll
  for (int ti = 0; ti < NTI; ti++) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, fh[ti][ks]), bl = __builtin_bit_cast(bf16x8, fl[ti][ks]);
#pragma unroll
      for (int to = 0; to < NTO; to++) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, s_w[(((ti * 2 + ks) * NTO + to) * 2 + 0) * 64 + lane]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, s_w[(((ti * 2 + ks) * NTO + to) * 2 + 1) * 64 + lane]);
        out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, out[to], 0, 0, 0);   // (the small terms first)
        out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, out[to], 0, 0, 0);
        out[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, out[to], 0, 0, 0);
      }
    }
  }
}

template <int L, int NIN>
__device__ __forceinline__ void mlp_layer_bf16(const uint4 *s_w, const float *bias, const uint4 (&fh)[NIN][2], const uint4 (&fl)[NIN][2],
                                               f32x16 (&out)[mlp_tout(L)], uint32_t lane) {
  constexpr int NTO = mlp_tout(L);
  static_assert(NIN == mlp_tin(L), "input tiles of the layer");
  const uint32_t half = lane >> 5;
#pragma unroll
  for (int t = 0; t < NTO; t++) {
#pragma unroll
    for (int i = 0; i < 16; i++) out[t][i] = bias[32 * t + mlp_row_of_reg(i) + 4 * half];
  }
  mlp_mm_bf16<NIN, NTO>(s_w, fh, fl, out, lane);
}

template <int L>
__device__ __forceinline__ void mlp_stage_bf16(const float *packed, float *s_w) {
  mlp_stage_block<MLP_PACKED_ALL + mlp_packed_offset(L), mlp_packed_floats(L)>(packed, s_w);
}

__global__ __launch_bounds__(MLP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_forward_bf16x3_kernel(int P, const float *xyz,
                                                                                                          const float *packed, float *out) {
  extern __shared__ __attribute__((aligned(16))) float s_mlp[];
  float *s_w = s_mlp, *s_b = s_mlp + MLP_LDS_FLOATS;
  const uint4 *s_f = reinterpret_cast<const uint4 *>(s_w);
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE, half = lane >> 5;
  const int p = (int)(blockIdx.x * (uint32_t)MLP_WG_POINTS + wave * 32u + (lane & 31u));
  float x = 0.f, y = 0.f, z = 0.f;
  if (p < P) {
    x = xyz[(size_t)p * 3 + 0];
    y = xyz[(size_t)p * 3 + 1];
    z = xyz[(size_t)p * 3 + 2];
  }
  for (int q = threadIdx.x; q < MLP_PACKED_B; q += MLP_WG) s_b[q] = packed[MLP_PACKED_W + q];
  mlp_stage_bf16<0>(packed, s_w);
  f32x16 pre[MLP_PRE];
  uint4 eh[2][2], el[2][2];
  {
    f32x16 emb[2];
    mlp_embed_tiles<true>(emb, (int)half, x, y, z);
    mlp_split_tiles<2>(emb, eh, el);
  }
  __syncthreads();
  f32x16 a[4];
  uint4 fh[4][2], fl[4][2];
  mlp_prefetch<MLP_PACKED_FWD16 + mlp_packed_offset(1), mlp_packed_floats(1)>(packed, pre);
  mlp_layer_bf16<0, 2>(s_f, s_b, eh, el, a, lane);
  mlp_relu(a);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_packed_floats(1)>(pre, s_w);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_FWD16 + mlp_packed_offset(2), mlp_packed_floats(2)>(packed, pre);
  mlp_layer_bf16<1, 4>(s_f, s_b + MLP_W, fh, fl, a, lane);
  mlp_relu(a);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_packed_floats(2)>(pre, s_w);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_FWD16 + mlp_packed_offset(3), mlp_packed_floats(3)>(packed, pre);
  mlp_layer_bf16<2, 4>(s_f, s_b + 2 * MLP_W, fh, fl, a, lane);
  mlp_relu(a);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_packed_floats(3)>(pre, s_w);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_FWD16 + mlp_packed_offset(4), mlp_packed_floats(4)>(packed, pre);
  {
    uint4 ch[6][2], cl[6][2];
#pragma unroll
    for (int k = 0; k < 6; k++) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        ch[k][q] = k < 2 ? eh[k][q] : fh[k - 2][q];
        cl[k][q] = k < 2 ? el[k][q] : fl[k - 2][q];
      }
    }
    mlp_layer_bf16<3, 6>(s_f, s_b + 3 * MLP_W, ch, cl, a, lane);
  }
  mlp_relu(a);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_packed_floats(4)>(pre, s_w);
  __syncthreads();
  f32x16 o[1];
  mlp_layer_bf16<4, 4>(s_f, s_b + 4 * MLP_W, fh, fl, o, lane);
  if (p < P) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = mlp_row_of_reg(i) + 4 * (int)half;
      if (row < MLP_OUT) out[(size_t)p * MLP_OUT + row] = o[0][i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward.  Two kernels:
//   chain : per 32 points of a wave, the forward again (no activation was kept) -- every layer's output is written feature-major
//           ([feature][point]: the lanes of a tile are consecutive points, so the stores coalesce) for the weight gradients and its
//           ReLU mask kept as 64 bits per lane -- then dh = W^T dZ layer by layer with the SAME register chaining (the accumulator
//           tile of dZ is the B operand; the A fragments are the transposed weights packed by mlp_pack_kernel), dZ written likewise;
//   wgrad : dW_l[o][k] = sum_p dZ_l[o][p] X_{l-1}[k][p] -- both operands are now contiguous along p, which is the contraction index
//           of this product: tiles of 32 points go through LDS (row stride 33: conflict-free operand reads) into A[i = o][kk = p]
//           and B[kk = p][j = k] fragments; a workgroup owns a chunk of points of ONE layer and adds its partial dW / db with atomics.
// Workspace rows (each Pp = P rounded up to a workgroup's 256 points, floats): emb 64 | h1 128 | h2 128 | h3 128 | h4 128 | dZ0..dZ3 4 x 128 | dOut^T 32.
constexpr int MLP_WS_EMB = 0, MLP_WS_H1 = 64, MLP_WS_H4 = MLP_WS_H1 + 3 * MLP_W, MLP_WS_DZ0 = MLP_WS_H4 + MLP_W,
              MLP_WS_DOUT = MLP_WS_DZ0 + 4 * MLP_W, MLP_WS_ROWS = MLP_WS_DOUT + 32;

template <int NT>
__device__ __forceinline__ void mlp_store_tiles(float *ws, size_t Pp, int row0, const f32x16 (&t)[NT], int pcol, uint32_t half) {
#pragma unroll
  for (int k = 0; k < NT; k++) {
#pragma unroll
    for (int i = 0; i < 16; i++) ws[(size_t)(row0 + 32 * k + mlp_row_of_reg(i) + 4 * (int)half) * Pp + pcol] = t[k][i];
  }
}
__device__ __forceinline__ void mlp_relu_mask(f32x16 (&t)[4], uint32_t (&m)[2]) {   // relu in place, bit (16 k + i) % 32 of m[k / 2]
  m[0] = m[1] = 0u;
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const bool pos = t[k][i] > 0.f;
      m[k / 2] |= pos ? (1u << ((16 * k + i) % 32)) : 0u;
      t[k][i] = pos ? t[k][i] : 0.f;
    }
  }
}
__device__ __forceinline__ void mlp_apply_mask(f32x16 (&t)[4], const uint32_t (&m)[2]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int i = 0; i < 16; i++) t[k][i] = ((m[k / 2] >> ((16 * k + i) % 32)) & 1u) ? t[k][i] : 0.f;
  }
}
__device__ __forceinline__ void mlp_zero(f32x16 (&t)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int i = 0; i < 16; i++) t[k][i] = 0.f;
  }
}

__global__ __launch_bounds__(MLP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_backward_chain_kernel(
    int P, int Pp_, const float *xyz, const float *packed, const float *dout, float *ws) {
  extern __shared__ __attribute__((aligned(16))) float s_mlp[];
  float *s_w = s_mlp, *s_b = s_mlp + MLP_LDS_FLOATS;
  const size_t Pp = (size_t)Pp_;
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE, half = lane >> 5;
  const int p = (int)(blockIdx.x * (uint32_t)MLP_WG_POINTS + wave * 32u + (lane & 31u));   // < Pp always
  float x = 0.f, y = 0.f, z = 0.f;
  if (p < P) {
    x = xyz[(size_t)p * 3 + 0];
    y = xyz[(size_t)p * 3 + 1];
    z = xyz[(size_t)p * 3 + 2];
  }
  for (int q = threadIdx.x; q < MLP_PACKED_B; q += MLP_WG) s_b[q] = packed[MLP_PACKED_W + q];
  mlp_stage<0>(packed, s_w);
  f32x16 emb[2];
  mlp_embed_tiles<false>(emb, (int)half, x, y, z);
  mlp_store_tiles<2>(ws, Pp, MLP_WS_EMB, emb, p, half);
  __syncthreads();
  // ---- the forward again: outputs to the workspace, masks to registers
  f32x16 a[4], b[4];
  uint32_t m1[2], m2[2], m3[2], m4[2];
  mlp_layer<0, 2>(s_w, s_b, emb, a, lane);
  mlp_relu_mask(a, m1);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_H1, a, p, half);
  __syncthreads();
  mlp_stage<1>(packed, s_w);
  __syncthreads();
  mlp_layer<1, 4>(s_w, s_b + MLP_W, a, b, lane);
  mlp_relu_mask(b, m2);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_H1 + MLP_W, b, p, half);
  __syncthreads();
  mlp_stage<2>(packed, s_w);
  __syncthreads();
  mlp_layer<2, 4>(s_w, s_b + 2 * MLP_W, b, a, lane);
  mlp_relu_mask(a, m3);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_H1 + 2 * MLP_W, a, p, half);
  __syncthreads();
  mlp_stage<3>(packed, s_w);
  __syncthreads();
  {
    f32x16 cat[6] = {emb[0], emb[1], a[0], a[1], a[2], a[3]};
    mlp_layer<3, 6>(s_w, s_b + 3 * MLP_W, cat, b, lane);
  }
  mlp_relu_mask(b, m4);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_H4, b, p, half);
  // ---- backward: dOut (rows = the 24 outputs, padded to 32) -> dh4 -> dZ3 -> dh3 -> dZ2 -> dh2 -> dZ1 -> dh1 -> dZ0
  f32x16 d[1];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int row = mlp_row_of_reg(i) + 4 * (int)half;
    d[0][i] = (row < MLP_OUT && p < P) ? dout[(size_t)p * MLP_OUT + row] : 0.f;
  }
  mlp_store_tiles<1>(ws, Pp, MLP_WS_DOUT, d, p, half);
  __syncthreads();
  mlp_stage_block<mlp_bwd_offset(0), mlp_bwd_floats(0)>(packed, s_w);
  __syncthreads();
  mlp_zero(a);
  mlp_mm<1, 4>(s_w, d, a, lane);
  mlp_apply_mask(a, m4);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_DZ0 + 3 * MLP_W, a, p, half);
  __syncthreads();
  mlp_stage_block<mlp_bwd_offset(1), mlp_bwd_floats(1)>(packed, s_w);
  __syncthreads();
  mlp_zero(b);
  mlp_mm<4, 4>(s_w, a, b, lane);
  mlp_apply_mask(b, m3);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_DZ0 + 2 * MLP_W, b, p, half);
  __syncthreads();
  mlp_stage_block<mlp_bwd_offset(2), mlp_bwd_floats(2)>(packed, s_w);
  __syncthreads();
  mlp_zero(a);
  mlp_mm<4, 4>(s_w, b, a, lane);
  mlp_apply_mask(a, m2);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_DZ0 + MLP_W, a, p, half);
  __syncthreads();
  mlp_stage_block<mlp_bwd_offset(3), mlp_bwd_floats(3)>(packed, s_w);
  __syncthreads();
  mlp_zero(b);
  mlp_mm<4, 4>(s_w, a, b, lane);
  mlp_apply_mask(b, m1);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_DZ0, b, p, half);
}

// the chain kernel on the bf16 instruction (both operands split in two bf16 terms, see mlp_forward_bf16x3_kernel): what the module runs
template <int ST>
__device__ __forceinline__ void mlp_stage_bwd16(const float *packed, float *s_w) {
  mlp_stage_block<MLP_PACKED_BWD16 + (mlp_bwd_offset(ST) - MLP_PACKED), mlp_bwd_floats(ST)>(packed, s_w);
}

__global__ __launch_bounds__(MLP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_backward_chain_bf16x3_kernel(
    int P, int Pp_, const float *xyz, const float *packed, const float *dout, float *ws) {
  extern __shared__ __attribute__((aligned(16))) float s_mlp[];
  float *s_w = s_mlp, *s_b = s_mlp + MLP_LDS_FLOATS;
  const uint4 *s_f = reinterpret_cast<const uint4 *>(s_w);
  const size_t Pp = (size_t)Pp_;
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE, half = lane >> 5;
  const int p = (int)(blockIdx.x * (uint32_t)MLP_WG_POINTS + wave * 32u + (lane & 31u));   // < Pp always
  float x = 0.f, y = 0.f, z = 0.f;
  if (p < P) {
    x = xyz[(size_t)p * 3 + 0];
    y = xyz[(size_t)p * 3 + 1];
    z = xyz[(size_t)p * 3 + 2];
  }
  for (int q = threadIdx.x; q < MLP_PACKED_B; q += MLP_WG) s_b[q] = packed[MLP_PACKED_W + q];
  mlp_stage_bf16<0>(packed, s_w);
  f32x16 pre[2];   // (two vectors = 64 KB per workgroup in flight: a third spills in this kernel; layer 3's last 32 KB are staged directly)
  uint4 eh[2][2], el[2][2];
  {
    f32x16 emb[2];
    mlp_embed_tiles<true>(emb, (int)half, x, y, z);
    mlp_store_tiles<2>(ws, Pp, MLP_WS_EMB, emb, p, half);
    mlp_split_tiles<2>(emb, eh, el);
  }
  __syncthreads();
  // ---- the forward again: outputs to the workspace, masks to registers
  f32x16 a[4];
  uint4 fh[4][2], fl[4][2];
  uint32_t m1[2], m2[2], m3[2], m4[2];
  mlp_prefetch<MLP_PACKED_FWD16 + mlp_packed_offset(1), mlp_packed_floats(1)>(packed, pre);
  mlp_layer_bf16<0, 2>(s_f, s_b, eh, el, a, lane);
  mlp_relu_mask(a, m1);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_H1, a, p, half);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_packed_floats(1)>(pre, s_w);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_FWD16 + mlp_packed_offset(2), mlp_packed_floats(2)>(packed, pre);
  mlp_layer_bf16<1, 4>(s_f, s_b + MLP_W, fh, fl, a, lane);
  mlp_relu_mask(a, m2);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_H1 + MLP_W, a, p, half);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_packed_floats(2)>(pre, s_w);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_FWD16 + mlp_packed_offset(3), 16384>(packed, pre);
  mlp_layer_bf16<2, 4>(s_f, s_b + 2 * MLP_W, fh, fl, a, lane);
  mlp_relu_mask(a, m3);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_H1 + 2 * MLP_W, a, p, half);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<16384>(pre, s_w);
  mlp_stage_block<MLP_PACKED_FWD16 + mlp_packed_offset(3) + 16384, mlp_packed_floats(3) - 16384>(packed, s_w + 16384);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_BWD16 + (mlp_bwd_offset(0) - MLP_PACKED), mlp_bwd_floats(0)>(packed, pre);
  {
    uint4 ch[6][2], cl[6][2];
#pragma unroll
    for (int k = 0; k < 6; k++) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        ch[k][q] = k < 2 ? eh[k][q] : fh[k - 2][q];
        cl[k][q] = k < 2 ? el[k][q] : fl[k - 2][q];
      }
    }
    mlp_layer_bf16<3, 6>(s_f, s_b + 3 * MLP_W, ch, cl, a, lane);
  }
  mlp_relu_mask(a, m4);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_H4, a, p, half);
  // ---- backward: dOut -> dh4 -> dZ3 -> dh3 -> dZ2 -> dh2 -> dZ1 -> dh1 -> dZ0
  uint4 dh[1][2], dl[1][2];
  {
    f32x16 d[1];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = mlp_row_of_reg(i) + 4 * (int)half;
      d[0][i] = (row < MLP_OUT && p < P) ? dout[(size_t)p * MLP_OUT + row] : 0.f;
    }
    mlp_store_tiles<1>(ws, Pp, MLP_WS_DOUT, d, p, half);
    mlp_split_tiles<1>(d, dh, dl);
  }
  __syncthreads();
  mlp_commit<mlp_bwd_floats(0)>(pre, s_w);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_BWD16 + (mlp_bwd_offset(1) - MLP_PACKED), mlp_bwd_floats(1)>(packed, pre);
  mlp_zero(a);
  mlp_mm_bf16<1, 4>(s_f, dh, dl, a, lane);
  mlp_apply_mask(a, m4);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_DZ0 + 3 * MLP_W, a, p, half);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_bwd_floats(1)>(pre, s_w);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_BWD16 + (mlp_bwd_offset(2) - MLP_PACKED), mlp_bwd_floats(2)>(packed, pre);
  mlp_zero(a);
  mlp_mm_bf16<4, 4>(s_f, fh, fl, a, lane);
  mlp_apply_mask(a, m3);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_DZ0 + 2 * MLP_W, a, p, half);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_bwd_floats(2)>(pre, s_w);
  __syncthreads();
  mlp_prefetch<MLP_PACKED_BWD16 + (mlp_bwd_offset(3) - MLP_PACKED), mlp_bwd_floats(3)>(packed, pre);
  mlp_zero(a);
  mlp_mm_bf16<4, 4>(s_f, fh, fl, a, lane);
  mlp_apply_mask(a, m2);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_DZ0 + MLP_W, a, p, half);
  mlp_split_tiles<4>(a, fh, fl);
  __syncthreads();
  mlp_commit<mlp_bwd_floats(3)>(pre, s_w);
  __syncthreads();
  mlp_zero(a);
  mlp_mm_bf16<4, 4>(s_f, fh, fl, a, lane);
  mlp_apply_mask(a, m1);
  mlp_store_tiles<4>(ws, Pp, MLP_WS_DZ0, a, p, half);
}

struct MlpWgradLayer {
  int a_row, ta;            // dZ rows in the workspace, tiles of 32
  int b_row0, tb0, b_row1;  // X rows: tb0 tiles from b_row0, the rest from b_row1
  int tb;
  float *dW, *db;
  int nrows, ncols, kind;   // kind 0: column = k (k < ncols); 3: layer 3's [emb(63) | pad | h(128)] columns
};
struct MlpWgradArgs {
  MlpWgradLayer L[MLP_LAYERS];
  const float *ws;
  int Pp, chunk;
};
constexpr int WG_STRIDE = 33;   // floats per staged row of 32 points

template <int TA, int TB>
__device__ __forceinline__ void mlp_wgrad_body(const MlpWgradLayer &L, const float *ws, size_t Pp, int p0, int p1, float *s_A, float *s_B) {
  constexpr int NACC = TA == 4 ? TB : 1;   // TA == 4: wave w owns dZ tile w and every X tile; TA == 1 (fc): X tile w
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE, half = lane >> 5, r = lane & 31u;
  f32x16 acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[k][i] = 0.f;
  }
  float dbsum = 0.f;
  const int my_ta = TA == 4 ? (int)wave : 0;
  // a stage = [rows][32 points], eight threads per row with 16 bytes each; the NEXT stage's global loads are issued before this
  // stage's MFMAs and land in LDS after them (the loads' latency used to sit exposed between two barriers)
  constexpr int NQ = ((TA + TB) * 32 * 8 + 255) / 256;   // 16-byte pieces per thread per stage
  float4 pre[NQ];
  auto fetch = [&](int pb) {
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      const int q = (int)threadIdx.x + j * 256;
      if (q < (TA + TB) * 32 * 8) {
        const int row = q / 8, c4 = (q % 8) * 4, rb = row - TA * 32;
        const int src_row = row < TA * 32 ? L.a_row + row : (rb < L.tb0 * 32 ? L.b_row0 + rb : L.b_row1 + (rb - L.tb0 * 32));
        pre[j] = *reinterpret_cast<const float4 *>(&ws[(size_t)src_row * Pp + pb + c4]);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      const int q = (int)threadIdx.x + j * 256;
      if (q < (TA + TB) * 32 * 8) {
        const int row = q / 8, c4 = (q % 8) * 4, rb = row - TA * 32;
        float *dst = (row < TA * 32 ? s_A + row * WG_STRIDE : s_B + rb * WG_STRIDE) + c4;
        dst[0] = pre[j].x, dst[1] = pre[j].y, dst[2] = pre[j].z, dst[3] = pre[j].w;
      }
    }
  };
  fetch(p0);
  for (int pb = p0; pb < p1; pb += 32) {
    commit();
    __syncthreads();
    if (pb + 32 < p1) fetch(pb + 32);
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const float af = s_A[(32 * my_ta + (int)r) * WG_STRIDE + 2 * s + (int)half];
      if constexpr (TA == 4) {
#pragma unroll
        for (int tb = 0; tb < TB; tb++) {
          const float bf = s_B[(32 * tb + (int)r) * WG_STRIDE + 2 * s + (int)half];
          acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[tb], 0, 0, 0);
        }
      } else {
        const float bf = s_B[(32 * (int)wave + (int)r) * WG_STRIDE + 2 * s + (int)half];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[0], 0, 0, 0);
      }
    }
    if ((int)threadIdx.x < TA * 32) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < 32; c++) t += s_A[(int)threadIdx.x * WG_STRIDE + c];
      dbsum += t;
    }
    __syncthreads();
  }
  // D[i = o][j = k]: column on the lane, rows in the registers
#pragma unroll
  for (int k = 0; k < NACC; k++) {
    const int tb = TA == 4 ? k : (int)wave;
    const int kf = 32 * tb + (int)r;
    int col = kf;
    if (L.kind == 3) col = kf < MLP_E ? kf : (kf == MLP_E ? -1 : MLP_E + (kf - 64));
    if (col >= L.ncols) col = -1;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int o = 32 * my_ta + mlp_row_of_reg(i) + 4 * (int)half;
      if (col >= 0 && o < L.nrows) atomicAdd(&L.dW[(size_t)o * L.ncols + col], acc[k][i]);
    }
  }
  if ((int)threadIdx.x < TA * 32 && (int)threadIdx.x < L.nrows) atomicAdd(&L.db[threadIdx.x], dbsum);
}

// the same products on the bf16 instruction: the staged rows are converted to two bf16 planes (hi, lo) on their way into LDS; a row
// of 32 points = 64 bytes + 16 of padding (stride 80: the 16-byte fragment reads of a lane group fall on 16 distinct bank quads);
// lane (r, h) of k-step s reads points 16 s + 8 h .. + 7 of row r -- one ds_read_b128 -- for A[i = o][kk = p] and B[kk = p][j = k] alike
constexpr int WG16_STRIDE = 20;   // 32-bit words per staged row of 32 bf16

template <int TA, int TB>
__device__ __forceinline__ void mlp_wgrad_body_bf16(const MlpWgradLayer &L, const float *ws, size_t Pp, int p0, int p1, uint32_t *s_Ah,
                                                    uint32_t *s_Al, uint32_t *s_Bh, uint32_t *s_Bl) {
  constexpr int NACC = TA == 4 ? TB : 1;
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE, half = lane >> 5, r = lane & 31u;
  f32x16 acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[k][i] = 0.f;
  }
  const int my_ta = TA == 4 ? (int)wave : 0;
  constexpr int NQ = ((TA + TB) * 32 * 8 + 255) / 256;   // 16-byte pieces per thread per stage
  constexpr int NQA = (TA * 32 * 8 + 255) / 256;          // ... of which the first NQA belong to dZ rows (row = tid / 8 + 32 j)
  float4 pre[NQ];
  float dbpart[NQA];
#pragma unroll
  for (int j = 0; j < NQA; j++) dbpart[j] = 0.f;
  auto fetch = [&](int pb) {
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      const int q = (int)threadIdx.x + j * 256;
      if (q < (TA + TB) * 32 * 8) {
        const int row = q / 8, c4 = (q % 8) * 4, rb = row - TA * 32;
        const int src_row = row < TA * 32 ? L.a_row + row : (rb < L.tb0 * 32 ? L.b_row0 + rb : L.b_row1 + (rb - L.tb0 * 32));
        pre[j] = *reinterpret_cast<const float4 *>(&ws[(size_t)src_row * Pp + pb + c4]);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      const int q = (int)threadIdx.x + j * 256;
      if (q < (TA + TB) * 32 * 8) {
        const int row = q / 8, c4 = (q % 8) * 4, rb = row - TA * 32;
        const bool isA = row < TA * 32;
        const float4 v = pre[j];
        uint32_t h0, h1, l0, l1;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h0) : "v"(v.x), "v"(v.y));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h1) : "v"(v.z), "v"(v.w));
        const float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xFFFF0000u);
        const float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xFFFF0000u);
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l0) : "v"(r0), "v"(r1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l1) : "v"(r2), "v"(r3));
        const int w = (isA ? row : rb) * WG16_STRIDE + c4 / 2;
        *reinterpret_cast<uint2 *>((isA ? s_Ah : s_Bh) + w) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>((isA ? s_Al : s_Bl) + w) = make_uint2(l0, l1);
        if (j < NQA && isA) dbpart[j < NQA ? j : 0] += (v.x + v.y) + (v.z + v.w);
      }
    }
  };
  fetch(p0);
  for (int pb = p0; pb < p1; pb += 32) {
    commit();
    __syncthreads();
    if (pb + 32 < p1) fetch(pb + 32);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int fo = 8 * ks + 4 * (int)half;   // word offset of the lane's eight points inside a row
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_Ah + (32 * my_ta + (int)r) * WG16_STRIDE + fo));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_Al + (32 * my_ta + (int)r) * WG16_STRIDE + fo));
#pragma unroll
      for (int k = 0; k < NACC; k++) {
        const int tb = TA == 4 ? k : (int)wave;
        const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_Bh + (32 * tb + (int)r) * WG16_STRIDE + fo));
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_Bl + (32 * tb + (int)r) * WG16_STRIDE + fo));
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[k], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NACC; k++) {
    const int tb = TA == 4 ? k : (int)wave;
    const int kf = 32 * tb + (int)r;
    int col = kf;
    if (L.kind == 3) col = kf < MLP_E ? kf : (kf == MLP_E ? -1 : MLP_E + (kf - 64));
    if (col >= L.ncols) col = -1;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int o = 32 * my_ta + mlp_row_of_reg(i) + 4 * (int)half;
      if (col >= 0 && o < L.nrows) atomicAdd(&L.dW[(size_t)o * L.ncols + col], acc[k][i]);
    }
  }
  // db[row]: the eight threads of a row (tid % 8) meet by shuffles; thread tid / 8 + 32 j owns dZ row tid / 8 + 32 j
#pragma unroll
  for (int j = 0; j < NQA; j++) {
    float t = dbpart[j];
    t += __shfl_xor(t, 1, WAVE);
    t += __shfl_xor(t, 2, WAVE);
    t += __shfl_xor(t, 4, WAVE);
    const int row = (int)threadIdx.x / 8 + 32 * j;
    if ((threadIdx.x & 7u) == 0u && row < TA * 32 && row < L.nrows) atomicAdd(&L.db[row], t);
  }
}

__global__ __launch_bounds__(256) void mlp_wgrad_bf16x3_kernel(const MlpWgradArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_Ah[128 * WG16_STRIDE], s_Al[128 * WG16_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t s_Bh[192 * WG16_STRIDE], s_Bl[192 * WG16_STRIDE];
  const MlpWgradLayer &L = a.L[blockIdx.y];
  const int p0 = (int)blockIdx.x * a.chunk, p1 = min(a.Pp, p0 + a.chunk);
  if (p0 >= p1) return;
  switch (blockIdx.y) {
    case 0: mlp_wgrad_body_bf16<4, 2>(L, a.ws, (size_t)a.Pp, p0, p1, s_Ah, s_Al, s_Bh, s_Bl); break;
    case 3: mlp_wgrad_body_bf16<4, 6>(L, a.ws, (size_t)a.Pp, p0, p1, s_Ah, s_Al, s_Bh, s_Bl); break;
    case 4: mlp_wgrad_body_bf16<1, 4>(L, a.ws, (size_t)a.Pp, p0, p1, s_Ah, s_Al, s_Bh, s_Bl); break;
    default: mlp_wgrad_body_bf16<4, 4>(L, a.ws, (size_t)a.Pp, p0, p1, s_Ah, s_Al, s_Bh, s_Bl); break;
  }
}

__global__ __launch_bounds__(256) void mlp_wgrad_kernel(const MlpWgradArgs a) {
  __shared__ float s_A[128 * WG_STRIDE];
  __shared__ float s_B[192 * WG_STRIDE];
  const MlpWgradLayer &L = a.L[blockIdx.y];
  const int p0 = (int)blockIdx.x * a.chunk, p1 = min(a.Pp, p0 + a.chunk);
  if (p0 >= p1) return;
  switch (blockIdx.y) {
    case 0: mlp_wgrad_body<4, 2>(L, a.ws, (size_t)a.Pp, p0, p1, s_A, s_B); break;
    case 3: mlp_wgrad_body<4, 6>(L, a.ws, (size_t)a.Pp, p0, p1, s_A, s_B); break;
    case 4: mlp_wgrad_body<1, 4>(L, a.ws, (size_t)a.Pp, p0, p1, s_A, s_B); break;
    default: mlp_wgrad_body<4, 4>(L, a.ws, (size_t)a.Pp, p0, p1, s_A, s_B); break;
  }
}

}  // namespace gsr

static std::atomic<int> g_mlp_precision{1};   // 0: f32 MFMA, 1: bf16 MFMA with both operands split in two terms (default)

extern "C" {

size_t gsr_lbs_offset_mlp_packed_floats(void) { return (size_t)gsr::MLP_PACKED_TOTAL; }

size_t gsr_lbs_offset_mlp_backward_workspace_floats(int P) {
  const size_t Pp = ((size_t)(P > 0 ? P : 0) + gsr::MLP_WG_POINTS - 1) / gsr::MLP_WG_POINTS * gsr::MLP_WG_POINTS;
  return (size_t)gsr::MLP_WS_ROWS * Pp;
}

int gsr_lbs_offset_mlp_backward(int P, const float *xyz, const float *packed, const float *dL_dout, float *workspace,
                                float *const *dL_dweights, float *const *dL_dbiases, gsr_stream_t stream_) {
  using namespace gsr;
  if (P < 0 || (P > 0 && (!xyz || !packed || !dL_dout || !workspace || !dL_dweights || !dL_dbiases))) {
    set_error("gsr_lbs_offset_mlp_backward: bad size or null pointer");
    return GSR_EINVAL;
  }
  if (reinterpret_cast<size_t>(packed) % 16 != 0 || reinterpret_cast<size_t>(workspace) % 16 != 0) {
    set_error("gsr_lbs_offset_mlp_backward: the packed weights and the workspace must be 16-byte aligned");
    return GSR_EINVAL;
  }
  for (int l = 0; l < MLP_LAYERS; l++) {
    if (P > 0 && (!dL_dweights[l] || !dL_dbiases[l])) {
      set_error("gsr_lbs_offset_mlp_backward: layer %d: null gradient array", l);
      return GSR_EINVAL;
    }
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int Pp = (P + MLP_WG_POINTS - 1) / MLP_WG_POINTS * MLP_WG_POINTS;
  constexpr size_t lds = (size_t)(MLP_LDS_FLOATS + MLP_PACKED_B) * sizeof(float);
  GSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds));
  GSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_backward_chain_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds));
  if (g_mlp_precision.load() == 1)
    hipLaunchKernelGGL(mlp_backward_chain_bf16x3_kernel, dim3((unsigned)(Pp / MLP_WG_POINTS)), dim3(MLP_WG), lds, stream, P, Pp, xyz, packed,
                       dL_dout, workspace);
  else
    hipLaunchKernelGGL(mlp_backward_chain_kernel, dim3((unsigned)(Pp / MLP_WG_POINTS)), dim3(MLP_WG), lds, stream, P, Pp, xyz, packed, dL_dout,
                       workspace);
  GSR_HIP(hipGetLastError());
  MlpWgradArgs a;
  memset(&a, 0, sizeof(a));
  a.ws = workspace, a.Pp = Pp, a.chunk = 1024;
  const int xrow[5] = {MLP_WS_EMB, MLP_WS_H1, MLP_WS_H1 + MLP_W, MLP_WS_EMB, MLP_WS_H4};
  for (int l = 0; l < MLP_LAYERS; l++) {
    MlpWgradLayer &L = a.L[l];
    L.a_row = l < 4 ? MLP_WS_DZ0 + l * MLP_W : MLP_WS_DOUT;
    L.ta = l < 4 ? 4 : 1;
    L.b_row0 = xrow[l];
    L.tb = mlp_tin(l);
    L.tb0 = l == 3 ? 2 : L.tb;
    L.b_row1 = MLP_WS_H1 + 2 * MLP_W;   // layer 3: h3 behind the embedding
    L.dW = dL_dweights[l], L.db = dL_dbiases[l];
    L.nrows = l < 4 ? MLP_W : MLP_OUT;
    L.ncols = l == 0 ? MLP_E : (l == 3 ? MLP_E + MLP_W : MLP_W);
    L.kind = l == 3 ? 3 : 0;
  }
  if (g_mlp_precision.load() == 1)
    hipLaunchKernelGGL(mlp_wgrad_bf16x3_kernel, dim3((unsigned)((Pp + a.chunk - 1) / a.chunk), MLP_LAYERS), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3((unsigned)((Pp + a.chunk - 1) / a.chunk), MLP_LAYERS), dim3(256), 0, stream, a);
  return check_hip(hipGetLastError(), "mlp_wgrad_kernel", __FILE__, __LINE__);
}


int gsr_lbs_offset_mlp_pack(const float *const *weights, const float *const *biases, float *packed, gsr_stream_t stream_) {
  using namespace gsr;
  if (!weights || !biases || !packed) {
    set_error("gsr_lbs_offset_mlp_pack: null argument");
    return GSR_EINVAL;
  }
  MlpWeights w;
  for (int l = 0; l < MLP_LAYERS; l++) {
    if (!weights[l] || !biases[l]) {
      set_error("gsr_lbs_offset_mlp_pack: layer %d: null weight or bias", l);
      return GSR_EINVAL;
    }
    w.w[l] = weights[l];
    w.b[l] = biases[l];
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((MLP_PACKED_TOTAL + 255) / 256), dim3(256), 0, stream, w, packed);
  return check_hip(hipGetLastError(), "mlp_pack_kernel", __FILE__, __LINE__);
}

int gsr_lbs_offset_mlp_set_precision(int mode) {
  if (mode != 0 && mode != 1) {
    gsr::set_error("gsr_lbs_offset_mlp_set_precision: 0 (f32 matrix instruction) or 1 (bf16 instruction, operands split in two terms)");
    return GSR_EINVAL;
  }
  g_mlp_precision.store(mode);
  return GSR_OK;
}

int gsr_debug_lbs_offset_mlp_forward_bf16x3(int P, const float *xyz, const float *packed, float *out, gsr_stream_t stream_) {
  using namespace gsr;
  if (P < 0 || (P > 0 && (!xyz || !packed || !out)) || reinterpret_cast<size_t>(packed) % 16 != 0) {
    set_error("gsr_debug_lbs_offset_mlp_forward_bf16x3: bad size, null pointer or misaligned fragments");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  constexpr size_t lds = (size_t)(MLP_LDS_FLOATS + MLP_PACKED_B) * sizeof(float);
  GSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_forward_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(mlp_forward_bf16x3_kernel, dim3((unsigned)((P + MLP_WG_POINTS - 1) / MLP_WG_POINTS)), dim3(MLP_WG), lds, stream, P, xyz,
                     packed, out);
  return check_hip(hipGetLastError(), "mlp_forward_bf16x3_kernel", __FILE__, __LINE__);
}

int gsr_lbs_offset_mlp_forward(int P, const float *xyz, const float *packed, float *out, gsr_stream_t stream_) {
  using namespace gsr;
  if (P < 0 || (P > 0 && (!xyz || !packed || !out))) {
    set_error("gsr_lbs_offset_mlp_forward: bad size or null pointer");
    return GSR_EINVAL;
  }
  if (reinterpret_cast<size_t>(packed) % 16 != 0) {
    set_error("gsr_lbs_offset_mlp_forward: the packed weights must be 16-byte aligned");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  constexpr size_t lds = (size_t)(MLP_LDS_FLOATS + MLP_PACKED_B) * sizeof(float);
  static_assert(lds <= 160 * 1024, "one layer's packed weights fit the LDS of a CU");
  // (dynamic LDS above 64 KB needs the attribute; set per call: it is per device and costs nothing)
  GSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (g_mlp_precision.load() == 1) return gsr_debug_lbs_offset_mlp_forward_bf16x3(P, xyz, packed, out, stream_);
  hipLaunchKernelGGL(mlp_forward_kernel, dim3((unsigned)((P + MLP_WG_POINTS - 1) / MLP_WG_POINTS)), dim3(MLP_WG), lds, stream, P, xyz, packed,
                     out);
  return check_hip(hipGetLastError(), "mlp_forward_kernel", __FILE__, __LINE__);
}

}  // extern "C"
