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

#include "gsr_common.h"

namespace gsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

__host__ __device__ constexpr int mlp_row_of_reg(int i) { return (i & 3) + 8 * (i >> 2); }

struct MlpWeights {   // the reference module's tensors (Conv1d weight [out][in][1] = row-major [out][in])
  const float *w[MLP_LAYERS];
  const float *b[MLP_LAYERS];
};

// one thread per packed float
__global__ __launch_bounds__(256) void mlp_pack_kernel(const MlpWeights src, float *packed) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= MLP_PACKED) return;
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

// embedding feature k (0..63; 63 = padding) of a point
__device__ __forceinline__ float mlp_embed(int k, float x, float y, float z) {
  if (k >= MLP_E) return 0.f;
  if (k < 3) return k == 0 ? x : (k == 1 ? y : z);
  const int t = k - 3, oct = t / 6, r = t % 6, c = r % 3;
  const float ang = (c == 0 ? x : (c == 1 ? y : z)) * (float)(1 << oct);
  return r < 3 ? sinf(ang) : cosf(ang);
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
#pragma unroll
  for (int ti = 0; ti < NIN; ti++) {
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

__device__ __forceinline__ void mlp_relu(f32x16 (&t)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int i = 0; i < 16; i++) t[k][i] = fmaxf(t[k][i], 0.f);
  }
}

// all threads of the workgroup: the packed weights of layer L -> LDS (16-byte copies); barriers on both sides by the caller
template <int L>
__device__ __forceinline__ void mlp_stage(const float *packed, float *s_w) {
  const float4 *src = reinterpret_cast<const float4 *>(packed + mlp_packed_offset(L));
  float4 *dst = reinterpret_cast<float4 *>(s_w);
  constexpr int N4 = mlp_packed_floats(L) / 4;
  for (int q = threadIdx.x; q < N4; q += 256) dst[q] = src[q];
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mlp_forward_kernel(int P, const float *xyz, const float *packed,
                                                                                                   float *out) {
  extern __shared__ __attribute__((aligned(16))) float s_mlp[];   // MLP_LDS_FLOATS weights + MLP_PACKED_B biases
  float *s_w = s_mlp, *s_b = s_mlp + MLP_LDS_FLOATS;
  const uint32_t lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE, half = lane >> 5;
  const int p = (int)(blockIdx.x * 128u + wave * 32u + (lane & 31u));
  float x = 0.f, y = 0.f, z = 0.f;
  if (p < P) {
    x = xyz[(size_t)p * 3 + 0];
    y = xyz[(size_t)p * 3 + 1];
    z = xyz[(size_t)p * 3 + 2];
  }
  for (int q = threadIdx.x; q < MLP_PACKED_B; q += 256) s_b[q] = packed[MLP_PACKED_W + q];
  mlp_stage<0>(packed, s_w);
  // the embedding as two activation tiles (features 0..31, 32..63)
  f32x16 emb[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
#pragma unroll
    for (int i = 0; i < 16; i++) emb[t][i] = mlp_embed(32 * t + mlp_row_of_reg(i) + 4 * (int)half, x, y, z);
  }
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

}  // namespace gsr

extern "C" {

size_t gsr_lbs_offset_mlp_packed_floats(void) { return (size_t)gsr::MLP_PACKED; }

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
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((MLP_PACKED + 255) / 256), dim3(256), 0, stream, w, packed);
  return check_hip(hipGetLastError(), "mlp_pack_kernel", __FILE__, __LINE__);
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
  hipLaunchKernelGGL(mlp_forward_kernel, dim3((unsigned)((P + 127) / 128)), dim3(256), lds, stream, P, xyz, packed, out);
  return check_hip(hipGetLastError(), "mlp_forward_kernel", __FILE__, __LINE__);
}

}  // extern "C"
