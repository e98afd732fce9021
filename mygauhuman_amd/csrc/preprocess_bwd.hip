// preprocess_bwd.hip -- per-Gaussian backward: conic -> cov2D -> (cov3D, mean), projection, SH, scale/rotation.
// One fused kernel replaces computeCov2DCUDA (CR/backward.cu:144-274) + preprocessCUDA (:346-396) and also
// unpacks the blend kernel's packed gradient rows into the binding's dL_dmean2D / dL_dconic / dL_dopacity /
// dL_dcolor tensors (DGR/rasterize_points.cu:159-167,206).  GLM products are written out in GLM's column-major
// evaluation order, m[c][r] = column c, row r.
#include "gsr_common.h"
#include "sh_math.h"

namespace gsr {

// CR/backward.cu:278-341 (quaternion used as given, no normalisation Jacobian)
__device__ __forceinline__ void cov3d_backward(const float3 sc, float mod, const float4 q, const float *d, float *dL_dscale,
                                               float *dL_drot) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  float R[3][3];
  R[0][0] = 1.f - 2.f * (y * y + z * z);
  R[0][1] = 2.f * (x * y - r * z);
  R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z);
  R[1][1] = 1.f - 2.f * (x * x + z * z);
  R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y);
  R[2][1] = 2.f * (y * z + r * x);
  R[2][2] = 1.f - 2.f * (x * x + y * y);
  const float s[3] = {mod * sc.x, mod * sc.y, mod * sc.z};
  float M[3][3];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int rr = 0; rr < 3; rr++) M[c][rr] = s[rr] * R[c][rr];
  const float dS[3][3] = {{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}};
  float dMt[3][3];  // dMt[c][r] = dL_dM[r][c], dL_dM = (2 M) dL_dSigma
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
      dMt[rr][c] = (2.0f * M[0][rr]) * dS[c][0] + (2.0f * M[1][rr]) * dS[c][1] + (2.0f * M[2][rr]) * dS[c][2];
#pragma unroll
  for (int k = 0; k < 3; k++) dL_dscale[k] = R[0][k] * dMt[k][0] + R[1][k] * dMt[k][1] + R[2][k] * dMt[k][2];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int rr = 0; rr < 3; rr++) dMt[k][rr] *= s[k];
  dL_drot[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
  dL_drot[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) -
               4 * x * (dMt[2][2] + dMt[1][1]);
  dL_drot[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) -
               4 * y * (dMt[2][2] + dMt[0][0]);
  dL_drot[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) -
               4 * z * (dMt[1][1] + dMt[0][0]);
}

// what the staged kernel needs to run the SH part itself, half a row at a time
struct DeferredSh {
  bool live;       // false: culled Gaussian, its SH gradient is zero
  float3 mean;
  float dRGB[3];   // clamp-masked colour gradient
  float dm[3];     // dL_dmean3D without the SH term (not stored yet)
};

// per-Gaussian body; sh_in / dsh_out point at this Gaussian's [M][3] blocks in global memory.  DEFER_SH: the SH part and the
// store of dL_dmean3D are left to the caller (out).
template <bool DEFER_SH>
__device__ __forceinline__ void preprocess_backward_one(const PreprocessBwdArgs &a, int i, const float *sh_in, float *dsh_out,
                                                        DeferredSh *out) {
  if (DEFER_SH) out->live = false;
  if (!(a.radii[i] > 0)) {
    // culled Gaussian: the reference leaves the zero-initialised outputs untouched (CR/backward.cu:156,367);
    // writing the zeros here lets the host hand in uninitialised tensors (no separate fill kernels)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      a.dL_dmean2D[3 * (size_t)i + k] = 0.f;
      a.dL_dcolor[3 * (size_t)i + k] = 0.f;
      a.dL_dmean3D[3 * (size_t)i + k] = 0.f;
      if (a.scales) a.dL_dscale[3 * (size_t)i + k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      a.dL_dconic[4 * (size_t)i + k] = 0.f;
      if (a.scales) a.dL_drot[4 * (size_t)i + k] = 0.f;
    }
    a.dL_dopacity[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)i + k] = 0.f;
    if (a.shs && !DEFER_SH)
      for (int k = 0; k < a.M * 3; k++) dsh_out[k] = 0.f;
    return;  // (the extra-channel gradients of the whole workgroup are copied cooperatively by the kernel)
  }
  float4 *row = reinterpret_cast<float4 *>(a.grad_rows + (size_t)i * a.grow);
  const float4 m0 = row[0], m1 = row[1], m2 = row[2];
  if (a.clear_rows) {  // columns 0..11 (with feature channels the workgroup has cleared the columns behind them, after its copy)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    row[0] = z;
    row[1] = z;
    row[2] = z;
  }
  // moments -> blend gradients (CR/backward.cu:567-584), conic / opacity from the forward's splat record
  const float4 rec0 = reinterpret_cast<const float4 *>(a.recs + i)[0];
  const float4 rec1 = reinterpret_cast<const float4 *>(a.recs + i)[1];
  const float cA = rec0.z, cB = rec0.w, cC = rec1.x, op = rec1.y;
  float4 g0, g1, g2;
  g0.x = -op * (cA * m0.x + cB * m0.y) * (0.5f * a.W);  // dL_dmean2D.x
  g0.y = -op * (cC * m0.y + cB * m0.x) * (0.5f * a.H);  // dL_dmean2D.y
  g0.z = -0.5f * op * m0.z;                             // dL_dconic.x
  g0.w = -0.5f * op * m0.w;                             // dL_dconic.y
  g1.x = -0.5f * op * m1.x;                             // dL_dconic.w
  g1.y = m1.y;                                          // dL_dopacity
  g1.z = m1.z;                                          // dL_dcolor
  g1.w = m1.w;
  g2.x = m2.x;
  // unpack the blend gradients into the binding's tensors
  a.dL_dmean2D[3 * (size_t)i + 0] = g0.x;
  a.dL_dmean2D[3 * (size_t)i + 1] = g0.y;
  a.dL_dmean2D[3 * (size_t)i + 2] = 0.f;
  a.dL_dconic[4 * (size_t)i + 0] = g0.z;
  a.dL_dconic[4 * (size_t)i + 1] = g0.w;
  a.dL_dconic[4 * (size_t)i + 2] = 0.f;
  a.dL_dconic[4 * (size_t)i + 3] = g1.x;
  a.dL_dopacity[i] = g1.y;
  a.dL_dcolor[3 * (size_t)i + 0] = g1.z;
  a.dL_dcolor[3 * (size_t)i + 1] = g1.w;
  a.dL_dcolor[3 * (size_t)i + 2] = g2.x;
  const float dcx = g0.z, dcy = g0.w, dcz = g1.x;

  const float *vm = a.view, *proj = a.proj;
  const float3 mean = make_float3(a.means3D[3 * (size_t)i], a.means3D[3 * (size_t)i + 1], a.means3D[3 * (size_t)i + 2]);
  float c6[6];
#pragma unroll
  for (int k = 0; k < 6; k++) c6[k] = a.cov3D[6 * (size_t)i + k];

  // ---- computeCov2DCUDA ----
  float3 t = make_float3(vm[0] * mean.x + vm[4] * mean.y + vm[8] * mean.z + vm[12],
                         vm[1] * mean.x + vm[5] * mean.y + vm[9] * mean.z + vm[13],
                         vm[2] * mean.x + vm[6] * mean.y + vm[10] * mean.z + vm[14]);
  const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
  const float txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  const float gxm = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  const float gym = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  const float h_x = a.focal_x, h_y = a.focal_y;
  const float J00 = h_x / t.z, J02 = -(h_x * t.x) / (t.z * t.z), J11 = h_y / t.z, J12 = -(h_y * t.y) / (t.z * t.z);
  const float Wm[3][3] = {{vm[0], vm[4], vm[8]}, {vm[1], vm[5], vm[9]}, {vm[2], vm[6], vm[10]}};
  float T[2][3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    T[0][r] = Wm[0][r] * J00 + Wm[2][r] * J02;
    T[1][r] = Wm[1][r] * J11 + Wm[2][r] * J12;
  }
  const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  float A[3][2];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    A[c][0] = T[0][0] * V[c][0] + T[0][1] * V[c][1] + T[0][2] * V[c][2];
    A[c][1] = T[1][0] * V[c][0] + T[1][1] * V[c][1] + T[1][2] * V[c][2];
  }
  const float ca = (A[0][0] * T[0][0] + A[1][0] * T[0][1] + A[2][0] * T[0][2]) + 0.3f;
  const float cb = A[0][1] * T[0][0] + A[1][1] * T[0][1] + A[2][1] * T[0][2];
  const float cc = (A[0][1] * T[1][0] + A[1][1] * T[1][1] + A[2][1] * T[1][2]) + 0.3f;
  const float denom = ca * cc - cb * cb;
  float dL_da = 0, dL_db = 0, dL_dc = 0;
  const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  float dcov[6];
  if (denom2inv != 0) {
    dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
    dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
    dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
    dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
    dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
    dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
    dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
    dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
    dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
  } else {
#pragma unroll
    for (int k = 0; k < 6; k++) dcov[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)i + k] = dcov[k];
  float dT[2][3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float tv0 = T[0][0] * V[k][0] + T[0][1] * V[k][1] + T[0][2] * V[k][2];
    const float tv1 = T[1][0] * V[k][0] + T[1][1] * V[k][1] + T[1][2] * V[k][2];
    dT[0][k] = 2 * tv0 * dL_da + tv1 * dL_db;
    dT[1][k] = 2 * tv1 * dL_dc + tv0 * dL_db;
  }
  const float dJ00 = Wm[0][0] * dT[0][0] + Wm[0][1] * dT[0][1] + Wm[0][2] * dT[0][2];
  const float dJ02 = Wm[2][0] * dT[0][0] + Wm[2][1] * dT[0][1] + Wm[2][2] * dT[0][2];
  const float dJ11 = Wm[1][0] * dT[1][0] + Wm[1][1] * dT[1][1] + Wm[1][2] * dT[1][2];
  const float dJ12 = Wm[2][0] * dT[1][0] + Wm[2][1] * dT[1][1] + Wm[2][2] * dT[1][2];
  const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
  const float dtx = gxm * -h_x * tz2 * dJ02;
  const float dty = gym * -h_y * tz2 * dJ12;
  const float dtz = -h_x * tz2 * dJ00 - h_y * tz2 * dJ11 + (2 * h_x * t.x) * tz3 * dJ02 + (2 * h_y * t.y) * tz3 * dJ12;
  float dm[3];
  dm[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
  dm[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
  dm[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;

  // ---- preprocessCUDA: projection term ----
  const float mhw = proj[3] * mean.x + proj[7] * mean.y + proj[11] * mean.z + proj[15];
  const float m_w = 1.0f / (mhw + 0.0000001f);
  const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
  const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
  const float g2x = g0.x, g2y = g0.y;
  dm[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
  dm[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
  dm[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;

  if (DEFER_SH) {
    const uint32_t cb = a.clamped[i];
    out->live = true;
    out->mean = mean;
    out->dRGB[0] = (cb & 1u) ? 0.f : g1.z;
    out->dRGB[1] = (cb & 2u) ? 0.f : g1.w;
    out->dRGB[2] = (cb & 4u) ? 0.f : g2.x;
    out->dm[0] = dm[0];
    out->dm[1] = dm[1];
    out->dm[2] = dm[2];
  } else {
    if (a.shs) {
      sh_backward(a.D, mean, a.campos, sh_in, a.clamped[i], make_float3(g1.z, g1.w, g2.x), dm, dsh_out);
      for (int k = (a.D + 1) * (a.D + 1) * 3; k < a.M * 3; k++) dsh_out[k] = 0.f;  // inactive bands
    }
    a.dL_dmean3D[3 * (size_t)i + 0] = dm[0];
    a.dL_dmean3D[3 * (size_t)i + 1] = dm[1];
    a.dL_dmean3D[3 * (size_t)i + 2] = dm[2];
  }

  if (a.scales) {
    float ds[3], dq[4];
    const float3 sc = make_float3(a.scales[3 * (size_t)i], a.scales[3 * (size_t)i + 1], a.scales[3 * (size_t)i + 2]);
    const float4 q = make_float4(a.rotations[4 * (size_t)i], a.rotations[4 * (size_t)i + 1], a.rotations[4 * (size_t)i + 2],
                                 a.rotations[4 * (size_t)i + 3]);
    cov3d_backward(sc, a.scale_modifier, q, dcov, ds, dq);
#pragma unroll
    for (int k = 0; k < 3; k++) a.dL_dscale[3 * (size_t)i + k] = ds[k];
#pragma unroll
    for (int k = 0; k < 4; k++) a.dL_drot[4 * (size_t)i + k] = dq[k];
  }
}

constexpr int BSH_M = 16, BSH_ROW = BSH_M * 3;
constexpr int BSH_HALF = BSH_ROW / 2;   // floats staged at a time (8 coefficients)
constexpr int BSH_LDS_ROW = 28;         // padded LDS row, see geometry.hip
constexpr int BSH_CPT = BSH_HALF / 4;   // 16-byte chunks per thread per half
constexpr int BWD_BLOCK = 128;          // Gaussians per workgroup: small, so that the 1563 workgroups of 200k Gaussians spread evenly (6-7 per CU)

// STAGE_SH: the workgroup's SH block (BWD_BLOCK x 192 B, contiguous) goes through LDS in two halves of 8 coefficients with coalesced
// 16-byte loads (the second half waits in registers); each thread works on its LDS row in place (coefficients in, gradients
// out) and the half goes out coalesced.  14 KB of LDS per workgroup instead of 52 (see geometry.hip).
template <bool STAGE_SH>
__global__ __launch_bounds__(BWD_BLOCK) void preprocess_backward_kernel(const PreprocessBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float s_sh[STAGE_SH ? BWD_BLOCK * BSH_LDS_ROW : 4];
  const int i = blockIdx.x * BWD_BLOCK + threadIdx.x;
  if (a.CE > 0) {
    // fused multi-feature blend: the colour gradients of the extra channels pass straight through, columns 9.. of the
    // gradient rows -> dL_dextra[P][CE]; the workgroup's rows are contiguous on both sides, so the copy is cooperative
    // (coalesced stores); rows of culled Gaussians were never touched by an atomic and are still zero
    const int first = blockIdx.x * BWD_BLOCK;
    const int nrows = min(BWD_BLOCK, a.P - first);
    const float *src = a.grad_rows + (size_t)first * a.grow + 9;
    float *dst = a.dL_dextra + (size_t)first * a.CE;
    for (int e = threadIdx.x; e < nrows * a.CE; e += BWD_BLOCK) {
      const int r = e / a.CE, c = e - r * a.CE;
      dst[e] = src[(size_t)r * a.grow + c];
    }
    if (a.clear_rows) {
      // the channel columns are consumed: clear them (columns 12.. of every row; 0..11 are cleared by the row's own thread below)
      __syncthreads();
      float *rows = a.grad_rows + (size_t)first * a.grow;
      for (int e = threadIdx.x; e < nrows * (a.grow - 12); e += BWD_BLOCK) {
        const int r = e / (a.grow - 12), c = e - r * (a.grow - 12);
        rows[(size_t)r * a.grow + 12 + c] = 0.f;
      }
    }
  }
  if (STAGE_SH) {
    const int first = blockIdx.x * BWD_BLOCK;
    const int nrows = min(BWD_BLOCK, a.P - first);
    float4 second[BSH_CPT];  // fp16 storage: 3 x 8 halves in [0..2]
    if (a.sh_half) {  // fp16 storage (see geometry.hip); the gradient that leaves below is fp32 either way
      const uint4 *slab = reinterpret_cast<const uint4 *>(reinterpret_cast<const _Float16 *>(a.shs) + (size_t)first * BSH_ROW);
#pragma unroll
      for (int j = 0; j < BSH_CPT / 2; j++) {
        const int e = (int)threadIdx.x + j * BWD_BLOCK, row = e / (BSH_CPT / 2), c8 = e % (BSH_CPT / 2);
        if (row < nrows) {
          const uint4 v = slab[row * (BSH_ROW / 8) + c8];
          second[j] = __builtin_bit_cast(float4, slab[row * (BSH_ROW / 8) + BSH_CPT / 2 + c8]);
          const _Float16 *hv = reinterpret_cast<const _Float16 *>(&v);
          float *dst = &s_sh[row * BSH_LDS_ROW + 8 * c8];
#pragma unroll
          for (int e2 = 0; e2 < 8; e2++) dst[e2] = (float)hv[e2];
        }
      }
    } else {
      const float4 *slab = reinterpret_cast<const float4 *>(a.shs + (size_t)first * BSH_ROW);
#pragma unroll
      for (int j = 0; j < BSH_CPT; j++) {
        const int e = (int)threadIdx.x + j * BWD_BLOCK, row = e / BSH_CPT, c4 = e % BSH_CPT;
        if (row < nrows) {
          *reinterpret_cast<float4 *>(&s_sh[row * BSH_LDS_ROW + 4 * c4]) = slab[row * (BSH_ROW / 4) + c4];
          second[j] = slab[row * (BSH_ROW / 4) + BSH_CPT + c4];
        }
      }
    }
    DeferredSh d;
    d.live = false;
    if (i < a.P) preprocess_backward_one<true>(a, i, nullptr, nullptr, &d);
    float3 d0 = make_float3(1.f, 0.f, 0.f);
    ShDir dir = {};
    float dd[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d.live) dir = sh_dir(d.mean, a.campos, d0);
    float *mine = &s_sh[threadIdx.x * BSH_LDS_ROW];
    float4 *out = reinterpret_cast<float4 *>(a.dL_dsh + (size_t)first * BSH_ROW);
    __syncthreads();  // first half staged
    if (d.live) {
      sh_backward_range<0, BSH_M / 2>(a.D, dir, mine, d.dRGB, dd);
    } else if (i < a.P) {
#pragma unroll
      for (int k = 0; k < BSH_HALF; k++) mine[k] = 0.f;
    }
    __syncthreads();
    // the first half's gradients leave, the second half's coefficients take their place (same thread, same LDS chunk)
#pragma unroll
    for (int j = 0; j < BSH_CPT; j++) {
      const int e = (int)threadIdx.x + j * BWD_BLOCK, row = e / BSH_CPT, c4 = e % BSH_CPT;
      if (row < nrows) {
        float4 *cell = reinterpret_cast<float4 *>(&s_sh[row * BSH_LDS_ROW + 4 * c4]);
        out[row * (BSH_ROW / 4) + c4] = *cell;
        if (!a.sh_half) *cell = second[j];
      }
    }
    if (a.sh_half) {
      __syncthreads();  // the fp16 chunks cover the rows differently (8 floats each): wait until every gradient chunk has left
#pragma unroll
      for (int j = 0; j < BSH_CPT / 2; j++) {
        const int e = (int)threadIdx.x + j * BWD_BLOCK, row = e / (BSH_CPT / 2), c8 = e % (BSH_CPT / 2);
        if (row < nrows) {
          const _Float16 *hv = reinterpret_cast<const _Float16 *>(&second[j]);
          float *dst = &s_sh[row * BSH_LDS_ROW + 8 * c8];
#pragma unroll
          for (int e2 = 0; e2 < 8; e2++) dst[e2] = (float)hv[e2];
        }
      }
    }
    __syncthreads();
    if (d.live) {
      sh_backward_range<BSH_M / 2, BSH_M>(a.D, dir, mine, d.dRGB, dd);
      sh_backward_finish(d0, d.dRGB, dd, d.dm);
      a.dL_dmean3D[3 * (size_t)i + 0] = d.dm[0];
      a.dL_dmean3D[3 * (size_t)i + 1] = d.dm[1];
      a.dL_dmean3D[3 * (size_t)i + 2] = d.dm[2];
    } else if (i < a.P) {
#pragma unroll
      for (int k = 0; k < BSH_HALF; k++) mine[k] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BSH_CPT; j++) {
      const int e = (int)threadIdx.x + j * BWD_BLOCK, row = e / BSH_CPT, c4 = e % BSH_CPT;
      if (row < nrows) out[row * (BSH_ROW / 4) + BSH_CPT + c4] = *reinterpret_cast<const float4 *>(&s_sh[row * BSH_LDS_ROW + 4 * c4]);
    }
  } else {
    if (i < a.P)
      preprocess_backward_one<false>(a, i, a.shs ? a.shs + (size_t)i * a.M * 3 : nullptr,
                                     a.shs ? a.dL_dsh + (size_t)i * a.M * 3 : nullptr, nullptr);
  }
}

int launch_preprocess_backward(const PreprocessBwdArgs &a, hipStream_t stream) {
  if (a.P <= 0) return GSR_OK;
  const bool stage = a.shs && a.M == BSH_M && (reinterpret_cast<size_t>(a.shs) % 16 == 0) &&
                     (reinterpret_cast<size_t>(a.dL_dsh) % 16 == 0);
  if (a.sh_half && a.shs && !stage) {
    set_error("fp16 SH storage needs 16 coefficients per Gaussian and 16-byte aligned arrays");
    return GSR_EINVAL;
  }
  if (stage)
    hipLaunchKernelGGL(preprocess_backward_kernel<true>, dim3((a.P + BWD_BLOCK - 1) / BWD_BLOCK), dim3(BWD_BLOCK), 0, stream, a);
  else
    hipLaunchKernelGGL(preprocess_backward_kernel<false>, dim3((a.P + BWD_BLOCK - 1) / BWD_BLOCK), dim3(BWD_BLOCK), 0, stream, a);
  return GSR_OK;
}

}  // namespace gsr
