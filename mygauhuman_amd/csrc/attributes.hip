// attributes.hip -- per-frame, per-Gaussian attributes that gaussian_renderer.render() derives between the LBS deform and
// the rasterizer (gaussian_renderer/__init__.py:128-198), forward and backward, one thread per Gaussian, ONE kernel each
// way instead of ~150 elementwise / batched-3x3-matmul launches:
//   cov3D[6]      = strip(T (R diag(mod*s))(R diag(mod*s))^T T^T)       scene/gaussian_model.py:35-42, utils/general_utils.py:64-117
//   colors[3]     = max(SH(active degree, dir) + 0.5, 0)                  utils/sh_utils.py:57-117, renderer :193-196
//   features[18]  = [normal | world_normal | albedo | occlusion | roughness | axis], the six extra colour sets of the
//                   reference's feature passes (:203-272) packed in the layout the fused blend kernel reads:
//     world_normal = n/|n| * .5 + .5                                      :160-161,171   (n = LBS-rotated canonical normal)
//     normal       = flipY(n/|n| . view3x3) * .5 + .5                     :165-170, transform.py:9-17
//     axis         = the same view transform of T a/|T a|, a = minimum_axis flipped towards the camera
//                    (scene/gaussian_model.py:186-190, utils/general_utils.py:144-157; the reference's row-0 quirk kept)
//     roughness    = mean of the three roughness channels, replicated      :258
// The backward recomputes the cheap intermediates from the inputs (nothing is saved besides the inputs themselves).
// Non-differentiable selections (argsort of the scales, flip sign, colour clamp) get zero gradient, as in autograd.
#include "gsr_common.h"
#include "sh_math.h"

namespace gsr {

struct AttrArgs {
  int P, D, M;
  const float *means, *transforms, *world_normals, *scales;
  float mod;
  const float *rot_cov, *rot_axis, *albedo, *roughness, *occlusion, *shs, *campos, *view;
  const float *shs_rest;  // non-null: shs holds only the DC coefficient [P][1][3], this the other 15 [P][15][3] (the model's two
                          // parameter tensors, read in place instead of through torch.cat)
  // forward outputs
  float *cov3D, *colors, *features;
  // backward inputs / outputs
  const float *g_cov3D, *g_colors, *g_features;
  float *d_means, *d_transforms, *d_world_normals, *d_scales, *d_rot_cov, *d_rot_axis, *d_albedo, *d_roughness, *d_occlusion,
      *d_shs, *d_shs_rest;
  const float *acc_means;  // backward, optional: a position gradient that already exists (the rasterizer's dL_dmeans3D) ADDED to d_means
};

constexpr int ATTR_BLOCK = 256;
constexpr int ASH_M = 16, ASH_ROW = 48, ASH_LDS_ROW = 52;
constexpr int NFEAT = 18;

__device__ __forceinline__ void quat_to_rot(const float q[4], float R[3][3]) {  // utils/general_utils.py:78-100 (q normalised)
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1.f - 2.f * (y * y + z * z);
  R[0][1] = 2.f * (x * y - w * z);
  R[0][2] = 2.f * (x * z + w * y);
  R[1][0] = 2.f * (x * y + w * z);
  R[1][1] = 1.f - 2.f * (x * x + z * z);
  R[1][2] = 2.f * (y * z - w * x);
  R[2][0] = 2.f * (x * z - w * y);
  R[2][1] = 2.f * (y * z + w * x);
  R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// gradient of quat_to_rot: dq[k] = sum_ij dR[i][j] dR_ij/dq_k
__device__ __forceinline__ void quat_to_rot_bwd(const float q[4], const float dR[3][3], float dq[4]) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  dq[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
  dq[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - w * dR[1][2] + z * dR[2][0] + w * dR[2][1] -
                 2.f * x * dR[2][2]);
  dq[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + w * dR[0][2] + x * dR[1][0] + z * dR[1][2] - w * dR[2][0] + z * dR[2][1] -
                 2.f * y * dR[2][2]);
  dq[3] = 2.f * (-2.f * z * dR[0][0] - w * dR[0][1] + x * dR[0][2] + w * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] +
                 x * dR[2][0] + y * dR[2][1]);
}

template <int N>
__device__ __forceinline__ float normalize_n(const float *v, float *out) {  // out = v/|v|, returns |v|
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < N; k++) s += v[k] * v[k];
  const float len = sqrtf(s);
#pragma unroll
  for (int k = 0; k < N; k++) out[k] = v[k] / len;
  return len;
}

template <int N>
__device__ __forceinline__ void normalize_bwd(const float *n, float len, const float *g, float *dv) {  // dv = (g - n (n.g))/|v|
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < N; k++) dot += n[k] * g[k];
  const float inv = 1.0f / len;
#pragma unroll
  for (int k = 0; k < N; k++) dv[k] = (g[k] - n[k] * dot) * inv;
}

// stable ascending argsort of three values (torch.argsort leaves the order of equal scales unspecified)
__device__ __forceinline__ void argsort3(const float s[3], int idx[3]) {
  const int r0 = (s[1] < s[0]) + (s[2] < s[0]);
  const int r1 = (s[0] <= s[1]) + (s[2] < s[1]);
  // idx[m] = the element whose rank is m (written without dynamic indexing: that would put the array in scratch)
#pragma unroll
  for (int m = 0; m < 3; m++) idx[m] = r0 == m ? 0 : (r1 == m ? 1 : 2);
}

// v (row vector) times the upper-left 3x3 of the row-major 4x4 `m`, y flipped, mapped to [0,1]  (:165-170)
__device__ __forceinline__ void to_view_colour(const float v[3], const float *m, float out[3]) {
  const float a = v[0] * m[0] + v[1] * m[4] + v[2] * m[8];
  const float b = v[0] * m[1] + v[1] * m[5] + v[2] * m[9];
  const float c = v[0] * m[2] + v[1] * m[6] + v[2] * m[10];
  out[0] = a * 0.5f + 0.5f;
  out[1] = -b * 0.5f + 0.5f;
  out[2] = c * 0.5f + 0.5f;
}
__device__ __forceinline__ void to_view_colour_bwd(const float g[3], const float *m, float dv[3]) {
  const float ga = 0.5f * g[0], gb = -0.5f * g[1], gc = 0.5f * g[2];
  dv[0] += ga * m[0] + gb * m[1] + gc * m[2];
  dv[1] += ga * m[4] + gb * m[5] + gc * m[6];
  dv[2] += ga * m[8] + gb * m[9] + gc * m[10];
}

struct AxisChain {  // intermediates of the minimum-axis chain, shared by forward and backward
  float q[4], qlen, R[3][3];
  int idx[3];
  float sign, ac[3], aclen, wa_raw[3], wa[3], walen;
};

__device__ __forceinline__ void axis_forward(const AttrArgs &a, int i, const float T[9], const float dirn[3], AxisChain &c) {
  float qa[4], s[3];
#pragma unroll
  for (int k = 0; k < 4; k++) qa[k] = a.rot_axis[(size_t)i * 4 + k];
#pragma unroll
  for (int k = 0; k < 3; k++) s[k] = a.scales[(size_t)i * 3 + k];
  c.qlen = normalize_n<4>(qa, c.q);
  quat_to_rot(c.q, c.R);
  argsort3(s, c.idx);
  float ar[3];
#pragma unroll
  for (int k = 0; k < 3; k++) ar[k] = c.idx[k] == 0 ? c.R[0][0] : (c.idx[k] == 1 ? c.R[0][1] : c.R[0][2]);
  const float d = -(ar[0] * dirn[0] + ar[1] * dirn[1] + ar[2] * dirn[2]);
  c.sign = d >= 0.f ? 1.0f : -1.0f;  // flip_align_view, utils/general_utils.py:151-157
  float af[3] = {ar[0] * c.sign, ar[1] * c.sign, ar[2] * c.sign};
  c.aclen = normalize_n<3>(af, c.ac);
#pragma unroll
  for (int r = 0; r < 3; r++) c.wa_raw[r] = T[r * 3 + 0] * c.ac[0] + T[r * 3 + 1] * c.ac[1] + T[r * 3 + 2] * c.ac[2];
  c.walen = normalize_n<3>(c.wa_raw, c.wa);
}

__device__ __forceinline__ void cov_forward(const AttrArgs &a, int i, const float T[9], float q[4], float &qlen, float L[3][3],
                                            float S[3][3], float TS[3][3]) {
  float qr[4], R[3][3];
#pragma unroll
  for (int k = 0; k < 4; k++) qr[k] = a.rot_cov[(size_t)i * 4 + k];
  qlen = normalize_n<4>(qr, q);
  quat_to_rot(q, R);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) L[r][k] = R[r][k] * (a.mod * a.scales[(size_t)i * 3 + k]);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) S[r][k] = L[r][0] * L[k][0] + L[r][1] * L[k][1] + L[r][2] * L[k][2];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) TS[r][k] = T[r * 3 + 0] * S[0][k] + T[r * 3 + 1] * S[1][k] + T[r * 3 + 2] * S[2][k];
}

__device__ __forceinline__ void attributes_forward_one(const AttrArgs &a, int i, const float *sh) {
  float T[9], mean[3], dirn[3];
#pragma unroll
  for (int k = 0; k < 9; k++) T[k] = a.transforms[(size_t)i * 9 + k];
#pragma unroll
  for (int k = 0; k < 3; k++) mean[k] = a.means[(size_t)i * 3 + k];
  const float dir[3] = {mean[0] - a.campos[0], mean[1] - a.campos[1], mean[2] - a.campos[2]};
  normalize_n<3>(dir, dirn);

  // ---- covariance
  {
    float q[4], qlen, L[3][3], S[3][3], TS[3][3];
    cov_forward(a, i, T, q, qlen, L, S, TS);
    float *o = a.cov3D + (size_t)i * 6;
#define GSR_C(r, k) (TS[r][0] * T[(k) * 3 + 0] + TS[r][1] * T[(k) * 3 + 1] + TS[r][2] * T[(k) * 3 + 2])
    o[0] = GSR_C(0, 0);
    o[1] = GSR_C(0, 1);
    o[2] = GSR_C(0, 2);
    o[3] = GSR_C(1, 1);
    o[4] = GSR_C(1, 2);
    o[5] = GSR_C(2, 2);
#undef GSR_C
  }
  // ---- view-dependent colour
  if (sh) {
    uint32_t clamp_bits;
    const float3 rgb = sh_to_rgb(a.D, make_float3(mean[0], mean[1], mean[2]), a.campos, sh, clamp_bits);
    a.colors[(size_t)i * 3 + 0] = rgb.x;
    a.colors[(size_t)i * 3 + 1] = rgb.y;
    a.colors[(size_t)i * 3 + 2] = rgb.z;
  }
  // ---- feature colours
  float f[NFEAT];
  float wn_raw[3], wn[3];
#pragma unroll
  for (int k = 0; k < 3; k++) wn_raw[k] = a.world_normals[(size_t)i * 3 + k];
  normalize_n<3>(wn_raw, wn);
  to_view_colour(wn, a.view, &f[0]);
#pragma unroll
  for (int k = 0; k < 3; k++) f[3 + k] = wn[k] * 0.5f + 0.5f;
#pragma unroll
  for (int k = 0; k < 3; k++) f[6 + k] = a.albedo[(size_t)i * 3 + k];
#pragma unroll
  for (int k = 0; k < 3; k++) f[9 + k] = a.occlusion[(size_t)i * 3 + k];
  const float rough = (a.roughness[(size_t)i * 3] + a.roughness[(size_t)i * 3 + 1] + a.roughness[(size_t)i * 3 + 2]) / 3.0f;
  f[12] = f[13] = f[14] = rough;
  AxisChain c;
  axis_forward(a, i, T, dirn, c);
  to_view_colour(c.wa, a.view, &f[15]);
  float2 *dst = reinterpret_cast<float2 *>(a.features + (size_t)i * NFEAT);
#pragma unroll
  for (int k = 0; k < NFEAT / 2; k++) dst[k] = make_float2(f[2 * k], f[2 * k + 1]);
}

__device__ __forceinline__ void attributes_backward_one(const AttrArgs &a, int i, const float *sh_in, float *dsh_out) {
  float T[9], dT[9], mean[3], dirn[3], dmean[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 9; k++) T[k] = a.transforms[(size_t)i * 9 + k];
#pragma unroll
  for (int k = 0; k < 3; k++) mean[k] = a.means[(size_t)i * 3 + k];
  const float dir[3] = {mean[0] - a.campos[0], mean[1] - a.campos[1], mean[2] - a.campos[2]};
  normalize_n<3>(dir, dirn);
  float dscale[3] = {0.f, 0.f, 0.f};

  // ---- covariance: Sigma' = T S T^T, S = L L^T, L = R diag(mod*s)
  {
    float q[4], qlen, L[3][3], S[3][3], TS[3][3];
    cov_forward(a, i, T, q, qlen, L, S, TS);
    float G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // symmetrised incoming gradient G + G^T
    if (a.g_cov3D) {
      const float *g = a.g_cov3D + (size_t)i * 6;
      G[0][0] = 2.f * g[0];
      G[1][1] = 2.f * g[3];
      G[2][2] = 2.f * g[5];
      G[0][1] = G[1][0] = g[1];
      G[0][2] = G[2][0] = g[2];
      G[1][2] = G[2][1] = g[4];
    }
    // dT = (G + G^T) T S
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) dT[r * 3 + k] = G[r][0] * TS[0][k] + G[r][1] * TS[1][k] + G[r][2] * TS[2][k];
    // dS (symmetrised) = T^T (G + G^T) T ;  dL = dS_sym L
    float GT[3][3], dS[3][3], dL[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) GT[r][k] = G[r][0] * T[0 * 3 + k] + G[r][1] * T[1 * 3 + k] + G[r][2] * T[2 * 3 + k];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) dS[r][k] = T[0 * 3 + r] * GT[0][k] + T[1 * 3 + r] * GT[1][k] + T[2 * 3 + r] * GT[2][k];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) dL[r][k] = dS[r][0] * L[0][k] + dS[r][1] * L[1][k] + dS[r][2] * L[2][k];
    float R[3][3], dR[3][3];
    quat_to_rot(q, R);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float sk = a.mod * a.scales[(size_t)i * 3 + k];
      dscale[k] = a.mod * (dL[0][k] * R[0][k] + dL[1][k] * R[1][k] + dL[2][k] * R[2][k]);
#pragma unroll
      for (int r = 0; r < 3; r++) dR[r][k] = dL[r][k] * sk;
    }
    float dq[4], dqr[4];
    quat_to_rot_bwd(q, dR, dq);
    normalize_bwd<4>(q, qlen, dq, dqr);
#pragma unroll
    for (int k = 0; k < 4; k++) a.d_rot_cov[(size_t)i * 4 + k] = dqr[k];
  }

  // ---- view-dependent colour
  if (sh_in) {
    uint32_t clamp_bits;
    const float3 pos = make_float3(mean[0], mean[1], mean[2]);
    sh_to_rgb(a.D, pos, a.campos, sh_in, clamp_bits);
    float3 gc = make_float3(0.f, 0.f, 0.f);
    if (a.g_colors) gc = make_float3(a.g_colors[(size_t)i * 3], a.g_colors[(size_t)i * 3 + 1], a.g_colors[(size_t)i * 3 + 2]);
    sh_backward(a.D, pos, a.campos, sh_in, clamp_bits, gc, dmean, dsh_out);
    for (int k = (a.D + 1) * (a.D + 1) * 3; k < a.M * 3; k++) dsh_out[k] = 0.f;  // inactive bands
  }

  // ---- feature colours
  float g[NFEAT];
  if (a.g_features) {
    const float2 *src = reinterpret_cast<const float2 *>(a.g_features + (size_t)i * NFEAT);
#pragma unroll
    for (int k = 0; k < NFEAT / 2; k++) {
      const float2 v = src[k];
      g[2 * k] = v.x;
      g[2 * k + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NFEAT; k++) g[k] = 0.f;
  }
  {
    float wn_raw[3], wn[3], dwn[3], dwn_raw[3];
#pragma unroll
    for (int k = 0; k < 3; k++) wn_raw[k] = a.world_normals[(size_t)i * 3 + k];
    const float len = normalize_n<3>(wn_raw, wn);
#pragma unroll
    for (int k = 0; k < 3; k++) dwn[k] = 0.5f * g[3 + k];
    to_view_colour_bwd(&g[0], a.view, dwn);
    normalize_bwd<3>(wn, len, dwn, dwn_raw);
#pragma unroll
    for (int k = 0; k < 3; k++) a.d_world_normals[(size_t)i * 3 + k] = dwn_raw[k];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) a.d_occlusion[(size_t)i * 3 + k] = g[9 + k];
  {
    const float gr = (g[12] + g[13] + g[14]) / 3.0f;
    if (a.d_roughness == a.d_albedo) {  // (kernel-uniform) albedo and roughness are ONE tensor (get_roughness reads _albedo): one sum
#pragma unroll
      for (int k = 0; k < 3; k++) a.d_albedo[(size_t)i * 3 + k] = g[6 + k] + gr;
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) a.d_albedo[(size_t)i * 3 + k] = g[6 + k];
#pragma unroll
      for (int k = 0; k < 3; k++) a.d_roughness[(size_t)i * 3 + k] = gr;
    }
  }
  {
    AxisChain c;
    axis_forward(a, i, T, dirn, c);
    float dwa[3] = {0.f, 0.f, 0.f}, dwa_raw[3];
    to_view_colour_bwd(&g[15], a.view, dwa);
    normalize_bwd<3>(c.wa, c.walen, dwa, dwa_raw);
    float dac[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) dT[r * 3 + k] += dwa_raw[r] * c.ac[k];
#pragma unroll
    for (int k = 0; k < 3; k++) dac[k] = T[0 * 3 + k] * dwa_raw[0] + T[1 * 3 + k] * dwa_raw[1] + T[2 * 3 + k] * dwa_raw[2];
    float daf[3];
    normalize_bwd<3>(c.ac, c.aclen, dac, daf);
    float dR[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float v = daf[k] * c.sign;
      dR[0][0] += c.idx[k] == 0 ? v : 0.f;
      dR[0][1] += c.idx[k] == 1 ? v : 0.f;
      dR[0][2] += c.idx[k] == 2 ? v : 0.f;
    }
    float dq[4], dqr[4];
    quat_to_rot_bwd(c.q, dR, dq);
    normalize_bwd<4>(c.q, c.qlen, dq, dqr);
#pragma unroll
    for (int k = 0; k < 4; k++) a.d_rot_axis[(size_t)i * 4 + k] = dqr[k];
  }
#pragma unroll
  for (int k = 0; k < 9; k++) a.d_transforms[(size_t)i * 9 + k] = dT[k];
#pragma unroll
  for (int k = 0; k < 3; k++) a.d_scales[(size_t)i * 3 + k] = dscale[k];
#pragma unroll
  for (int k = 0; k < 3; k++) a.d_means[(size_t)i * 3 + k] = dmean[k] + (a.acc_means ? a.acc_means[(size_t)i * 3 + k] : 0.f);
}

// STAGE: the workgroup's SH block (256 rows x 192 B, contiguous) goes through LDS with coalesced 16-byte accesses; the
// backward works on its LDS row in place (coefficients in, gradients out) and the block leaves coalesced.
// (amdgpu_waves_per_eu(3): the staged SH rows (52 KB per workgroup) admit three waves per SIMD; the backward's 172 VGPRs allowed two:
// 168 with two spilled = three, 48.1 -> 45.3 us in the render() frame)
template <bool STAGE, bool BWD>
__global__ __launch_bounds__(ATTR_BLOCK) __attribute__((amdgpu_waves_per_eu(3))) void attributes_kernel(const AttrArgs a) {
  __shared__ __attribute__((aligned(16))) float s_sh[STAGE ? ATTR_BLOCK * ASH_LDS_ROW : 4];
  const int i = blockIdx.x * ATTR_BLOCK + threadIdx.x;
  if (STAGE) {
    const int first = blockIdx.x * ATTR_BLOCK;
    const int nrows = min(ATTR_BLOCK, a.P - first);
    constexpr int REST = ASH_ROW - 3;  // floats of the 15 higher coefficients
    if (a.shs_rest) {
      // two source arrays: rows of 3 floats and rows of 45 floats, each block contiguous; 16-byte loads where the block allows
      const float *dc = a.shs + (size_t)first * 3;
      for (int e = threadIdx.x; e < nrows * 3; e += ATTR_BLOCK) s_sh[(e / 3) * ASH_LDS_ROW + e % 3] = dc[e];
      const float *rest = a.shs_rest + (size_t)first * REST;
      const int n4 = nrows * REST / 4;
      const float4 *rest4 = reinterpret_cast<const float4 *>(rest);
      for (int q = threadIdx.x; q < n4; q += ATTR_BLOCK) {
        const float4 v = rest4[q];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int e = 4 * q + j;
          s_sh[(e / REST) * ASH_LDS_ROW + 3 + e % REST] = vv[j];
        }
      }
      for (int e = 4 * n4 + (int)threadIdx.x; e < nrows * REST; e += ATTR_BLOCK) s_sh[(e / REST) * ASH_LDS_ROW + 3 + e % REST] = rest[e];
    } else {
      const float4 *slab = reinterpret_cast<const float4 *>(a.shs + (size_t)first * ASH_ROW);
      for (int q = threadIdx.x; q < nrows * (ASH_ROW / 4); q += ATTR_BLOCK) {
        const int row = q / (ASH_ROW / 4), k4 = q % (ASH_ROW / 4);
        *reinterpret_cast<float4 *>(&s_sh[row * ASH_LDS_ROW + 4 * k4]) = slab[q];
      }
    }
    __syncthreads();
    float *row = &s_sh[threadIdx.x * ASH_LDS_ROW];
    if (i < a.P) {
      if (BWD)
        attributes_backward_one(a, i, row, row);
      else
        attributes_forward_one(a, i, row);
    }
    if (BWD && a.d_shs) {  // (d_shs == null: the caller does not want the SH gradient -- view-parallel compact exchange)
      __syncthreads();
      if (a.shs_rest) {
        float *dc = a.d_shs + (size_t)first * 3;
        for (int e = threadIdx.x; e < nrows * 3; e += ATTR_BLOCK) dc[e] = s_sh[(e / 3) * ASH_LDS_ROW + e % 3];
        float *rest = a.d_shs_rest + (size_t)first * REST;
        const int n4 = nrows * REST / 4;
        float4 *rest4 = reinterpret_cast<float4 *>(rest);
        for (int q = threadIdx.x; q < n4; q += ATTR_BLOCK) {
          float vv[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int e = 4 * q + j;
            vv[j] = s_sh[(e / REST) * ASH_LDS_ROW + 3 + e % REST];
          }
          rest4[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        }
        for (int e = 4 * n4 + (int)threadIdx.x; e < nrows * REST; e += ATTR_BLOCK) rest[e] = s_sh[(e / REST) * ASH_LDS_ROW + 3 + e % REST];
      } else {
        float4 *out = reinterpret_cast<float4 *>(a.d_shs + (size_t)first * ASH_ROW);
        for (int q = threadIdx.x; q < nrows * (ASH_ROW / 4); q += ATTR_BLOCK) {
          const int r = q / (ASH_ROW / 4), k4 = q % (ASH_ROW / 4);
          out[q] = *reinterpret_cast<const float4 *>(&s_sh[r * ASH_LDS_ROW + 4 * k4]);
        }
      }
    }
  } else if (i < a.P) {
    const float *sh = a.shs ? a.shs + (size_t)i * a.M * 3 : nullptr;
    if (BWD)
      attributes_backward_one(a, i, sh, a.shs ? a.d_shs + (size_t)i * a.M * 3 : nullptr);
    else
      attributes_forward_one(a, i, sh);
  }
}

static int check_common(const char *who, int P, int D, int M, const float *means, const float *transforms, const float *wn,
                        const float *scales, const float *rot_cov, const float *rot_axis, const float *albedo,
                        const float *roughness, const float *occlusion, const float *shs, const float *campos, const float *view) {
  if (P < 0 || D < 0 || D > 3) {
    set_error("%s: P must be >= 0 and the SH degree in 0..3", who);
    return GSR_EINVAL;
  }
  if (P > 0 && (!means || !transforms || !wn || !scales || !rot_cov || !rot_axis || !albedo || !roughness || !occlusion ||
                !campos || !view)) {
    set_error("%s: null input array", who);
    return GSR_EINVAL;
  }
  if (shs && M < (D + 1) * (D + 1)) {
    set_error("%s: %d SH coefficients per Gaussian are too few for degree %d", who, M, D);
    return GSR_EINVAL;
  }
  return GSR_OK;
}

}  // namespace gsr

extern "C" {

static int split_ok(const char *who, int M, const void *dc, const void *rest, const void *d_dc, const void *d_rest, bool bwd) {
  if (!rest) return GSR_OK;
  const bool no_grad = bwd && !d_dc && !d_rest;  // both null: the SH gradient is not wanted
  if (!dc || M != gsr::ASH_M || reinterpret_cast<size_t>(rest) % 16 != 0 ||
      (bwd && !no_grad && (!d_dc || !d_rest || reinterpret_cast<size_t>(d_rest) % 16 != 0))) {
    gsr::set_error("%s: split SH arrays need M = 16, the DC array and 16-byte aligned [P][15][3] arrays", who);
    return GSR_EINVAL;
  }
  return GSR_OK;
}

int gsr_frame_attributes_forward_split(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                       const float *world_normals, const float *scales, float scale_modifier, const float *rot_cov,
                                       const float *rot_axis, const float *albedo, const float *roughness, const float *occlusion,
                                       const float *shs, const float *shs_rest, const float *campos, const float *viewmatrix,
                                       float *cov3D, float *colors, float *features, gsr_stream_t stream_);

int gsr_frame_attributes_forward(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                 const float *world_normals, const float *scales, float scale_modifier, const float *rot_cov,
                                 const float *rot_axis, const float *albedo, const float *roughness, const float *occlusion,
                                 const float *shs, const float *campos, const float *viewmatrix, float *cov3D, float *colors,
                                 float *features, gsr_stream_t stream_) {
  return gsr_frame_attributes_forward_split(P, sh_degree, M, means3D, transforms, world_normals, scales, scale_modifier, rot_cov,
                                            rot_axis, albedo, roughness, occlusion, shs, nullptr, campos, viewmatrix, cov3D, colors,
                                            features, stream_);
}

int gsr_frame_attributes_forward_split(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                       const float *world_normals, const float *scales, float scale_modifier, const float *rot_cov,
                                       const float *rot_axis, const float *albedo, const float *roughness, const float *occlusion,
                                       const float *shs, const float *shs_rest, const float *campos, const float *viewmatrix,
                                       float *cov3D, float *colors, float *features, gsr_stream_t stream_) {
  using namespace gsr;
  if (split_ok("gsr_frame_attributes_forward_split", M, shs, shs_rest, nullptr, nullptr, false) != GSR_OK) return GSR_EINVAL;
  int rc = check_common("gsr_frame_attributes_forward", P, sh_degree, M, means3D, transforms, world_normals, scales, rot_cov,
                        rot_axis, albedo, roughness, occlusion, shs, campos, viewmatrix);
  if (rc != GSR_OK) return rc;
  if (P > 0 && (!cov3D || !features || (shs && !colors))) {
    set_error("gsr_frame_attributes_forward: null output array");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  AttrArgs a = {};
  a.P = P, a.D = sh_degree, a.M = M;
  a.means = means3D, a.transforms = transforms, a.world_normals = world_normals, a.scales = scales, a.mod = scale_modifier;
  a.rot_cov = rot_cov, a.rot_axis = rot_axis, a.albedo = albedo, a.roughness = roughness, a.occlusion = occlusion;
  a.shs = shs, a.shs_rest = shs_rest, a.campos = campos, a.view = viewmatrix;
  a.cov3D = cov3D, a.colors = colors, a.features = features;
  const bool stage = shs && M == ASH_M && (shs_rest || reinterpret_cast<size_t>(shs) % 16 == 0);
  const dim3 grid((P + ATTR_BLOCK - 1) / ATTR_BLOCK), block(ATTR_BLOCK);
  if (stage)
    hipLaunchKernelGGL((attributes_kernel<true, false>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((attributes_kernel<false, false>), grid, block, 0, stream, a);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_frame_attributes_backward_split(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                        const float *world_normals, const float *scales, float scale_modifier, const float *rot_cov,
                                        const float *rot_axis, const float *albedo, const float *roughness, const float *occlusion,
                                        const float *shs, const float *shs_rest, const float *campos, const float *viewmatrix,
                                        const float *dL_dcov3D, const float *dL_dcolors, const float *dL_dfeatures,
                                        float *dL_dmeans3D, float *dL_dtransforms, float *dL_dworld_normals, float *dL_dscales,
                                        float *dL_drot_cov, float *dL_drot_axis, float *dL_dalbedo, float *dL_droughness,
                                        float *dL_docclusion, float *dL_dshs, float *dL_dshs_rest, gsr_stream_t stream_);

int gsr_frame_attributes_backward(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                  const float *world_normals, const float *scales, float scale_modifier, const float *rot_cov,
                                  const float *rot_axis, const float *albedo, const float *roughness, const float *occlusion,
                                  const float *shs, const float *campos, const float *viewmatrix, const float *dL_dcov3D,
                                  const float *dL_dcolors, const float *dL_dfeatures, float *dL_dmeans3D, float *dL_dtransforms,
                                  float *dL_dworld_normals, float *dL_dscales, float *dL_drot_cov, float *dL_drot_axis,
                                  float *dL_dalbedo, float *dL_droughness, float *dL_docclusion, float *dL_dshs,
                                  gsr_stream_t stream_) {
  return gsr_frame_attributes_backward_split(P, sh_degree, M, means3D, transforms, world_normals, scales, scale_modifier, rot_cov,
                                             rot_axis, albedo, roughness, occlusion, shs, nullptr, campos, viewmatrix, dL_dcov3D,
                                             dL_dcolors, dL_dfeatures, dL_dmeans3D, dL_dtransforms, dL_dworld_normals, dL_dscales,
                                             dL_drot_cov, dL_drot_axis, dL_dalbedo, dL_droughness, dL_docclusion, dL_dshs, nullptr,
                                             stream_);
}

int gsr_frame_attributes_backward_acc(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                        const float *world_normals, const float *scales, float scale_modifier, const float *rot_cov,
                                        const float *rot_axis, const float *albedo, const float *roughness, const float *occlusion,
                                        const float *shs, const float *shs_rest, const float *campos, const float *viewmatrix,
                                        const float *dL_dcov3D, const float *dL_dcolors, const float *dL_dfeatures,
                                        float *dL_dmeans3D, float *dL_dtransforms, float *dL_dworld_normals, float *dL_dscales,
                                        float *dL_drot_cov, float *dL_drot_axis, float *dL_dalbedo, float *dL_droughness,
                                        float *dL_docclusion, float *dL_dshs, float *dL_dshs_rest, const float *acc_dmeans3D,
                                        gsr_stream_t stream_) {
  using namespace gsr;
  if (split_ok("gsr_frame_attributes_backward_split", M, shs, shs_rest, dL_dshs, dL_dshs_rest, true) != GSR_OK) return GSR_EINVAL;
  int rc = check_common("gsr_frame_attributes_backward", P, sh_degree, M, means3D, transforms, world_normals, scales, rot_cov,
                        rot_axis, albedo, roughness, occlusion, shs, campos, viewmatrix);
  if (rc != GSR_OK) return rc;
  if (P > 0 && (!dL_dmeans3D || !dL_dtransforms || !dL_dworld_normals || !dL_dscales || !dL_drot_cov || !dL_drot_axis ||
                !dL_dalbedo || !dL_droughness || !dL_docclusion)) {
    set_error("gsr_frame_attributes_backward: null output array");
    return GSR_EINVAL;
  }
  // dL_dshs == null (and dL_dshs_rest == null): the SH gradient is not produced -- only in the staged 16-coefficient layouts
  if (P > 0 && shs && !dL_dshs && !(M == ASH_M && (shs_rest || reinterpret_cast<size_t>(shs) % 16 == 0))) {
    set_error("gsr_frame_attributes_backward: dL_dshs may only be omitted for M = 16 (16-byte aligned or split SH arrays)");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  AttrArgs a = {};
  a.P = P, a.D = sh_degree, a.M = M;
  a.means = means3D, a.transforms = transforms, a.world_normals = world_normals, a.scales = scales, a.mod = scale_modifier;
  a.rot_cov = rot_cov, a.rot_axis = rot_axis, a.albedo = albedo, a.roughness = roughness, a.occlusion = occlusion;
  a.shs = shs, a.shs_rest = shs_rest, a.campos = campos, a.view = viewmatrix;
  a.g_cov3D = dL_dcov3D, a.g_colors = dL_dcolors, a.g_features = dL_dfeatures;
  a.d_means = dL_dmeans3D, a.d_transforms = dL_dtransforms, a.d_world_normals = dL_dworld_normals, a.d_scales = dL_dscales;
  a.d_rot_cov = dL_drot_cov, a.d_rot_axis = dL_drot_axis, a.d_albedo = dL_dalbedo, a.d_roughness = dL_droughness;
  a.d_occlusion = dL_docclusion, a.d_shs = dL_dshs, a.d_shs_rest = dL_dshs_rest;
  a.acc_means = acc_dmeans3D;
  const bool stage = shs && M == ASH_M &&
                     (shs_rest || (reinterpret_cast<size_t>(shs) % 16 == 0 && reinterpret_cast<size_t>(dL_dshs) % 16 == 0));  // (null is aligned)
  const dim3 grid((P + ATTR_BLOCK - 1) / ATTR_BLOCK), block(ATTR_BLOCK);
  if (stage)
    hipLaunchKernelGGL((attributes_kernel<true, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((attributes_kernel<false, true>), grid, block, 0, stream, a);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_frame_attributes_backward_split(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                        const float *world_normals, const float *scales, float scale_modifier, const float *rot_cov,
                                        const float *rot_axis, const float *albedo, const float *roughness, const float *occlusion,
                                        const float *shs, const float *shs_rest, const float *campos, const float *viewmatrix,
                                        const float *dL_dcov3D, const float *dL_dcolors, const float *dL_dfeatures,
                                        float *dL_dmeans3D, float *dL_dtransforms, float *dL_dworld_normals, float *dL_dscales,
                                        float *dL_drot_cov, float *dL_drot_axis, float *dL_dalbedo, float *dL_droughness,
                                        float *dL_docclusion, float *dL_dshs, float *dL_dshs_rest, gsr_stream_t stream_) {
  return gsr_frame_attributes_backward_acc(P, sh_degree, M, means3D, transforms, world_normals, scales, scale_modifier, rot_cov, rot_axis,
                                           albedo, roughness, occlusion, shs, shs_rest, campos, viewmatrix, dL_dcov3D, dL_dcolors,
                                           dL_dfeatures, dL_dmeans3D, dL_dtransforms, dL_dworld_normals, dL_dscales, dL_drot_cov,
                                           dL_drot_axis, dL_dalbedo, dL_droughness, dL_docclusion, dL_dshs, dL_dshs_rest, nullptr, stream_);
}

}  // extern "C"
