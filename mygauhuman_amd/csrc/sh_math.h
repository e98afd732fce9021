// sh_math.h -- real spherical harmonics up to degree 3: colour evaluation and its backward, shared by the rasterizer's
// preprocess kernels (geometry.hip, preprocess_bwd.hip) and the per-frame attribute kernels (attributes.hip).
// Formulas and coefficient order follow CR/forward.cu:20-71 / CR/backward.cu:20-139 (identical to utils/sh_utils.py:57-117).
#pragma once
#include "gsr_common.h"

namespace gsr {

constexpr float kSH0 = 0.28209479177387814f;  // CR/auxiliary.h:22-39
constexpr float kSH1 = 0.4886025119029199f;
constexpr float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                           0.5462742152960396f};
constexpr float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                           -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

// SH -> RGB (CR/forward.cu:20-71); sh points at this Gaussian's [M][3] block
__device__ __forceinline__ float3 sh_to_rgb(int deg, const float3 pos, const float *campos, const float *sh,
                                            uint32_t &clamp_bits) {
  const float dx0 = pos.x - campos[0], dy0 = pos.y - campos[1], dz0 = pos.z - campos[2];
  const float len = sqrtf(dx0 * dx0 + dy0 * dy0 + dz0 * dz0);
  const float x = dx0 / len, y = dy0 / len, z = dz0 / len;
  float out[3];
  clamp_bits = 0;
#pragma unroll
  for (int ch = 0; ch < 3; ch++) {
#define S(k) sh[(k) * 3 + ch]
    float res = kSH0 * S(0);
    if (deg > 0) {
      res = res - kSH1 * y * S(1) + kSH1 * z * S(2) - kSH1 * x * S(3);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + kSH2[0] * xy * S(4) + kSH2[1] * yz * S(5) + kSH2[2] * (2.0f * zz - xx - yy) * S(6) +
              kSH2[3] * xz * S(7) + kSH2[4] * (xx - yy) * S(8);
        if (deg > 2) {
          res = res + kSH3[0] * y * (3.0f * xx - yy) * S(9) + kSH3[1] * xy * z * S(10) +
                kSH3[2] * y * (4.0f * zz - xx - yy) * S(11) + kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                kSH3[4] * x * (4.0f * zz - xx - yy) * S(13) + kSH3[5] * z * (xx - yy) * S(14) +
                kSH3[6] * x * (xx - 3.0f * yy) * S(15);
        }
      }
    }
#undef S
    res += 0.5f;
    if (res < 0) clamp_bits |= 1u << ch;
    out[ch] = fmaxf(res, 0.0f);
  }
  return make_float3(out[0], out[1], out[2]);
}

// CR/backward.cu:20-139
__device__ __forceinline__ void sh_backward(int deg, const float3 pos, const float *campos, const float *sh,
                                            uint32_t clamp_bits, const float3 dL_dcolor, float *dL_dmean, float *dL_dsh) {
  const float d0x = pos.x - campos[0], d0y = pos.y - campos[1], d0z = pos.z - campos[2];
  const float len = sqrtf(d0x * d0x + d0y * d0y + d0z * d0z);
  const float x = d0x / len, y = d0y / len, z = d0z / len;
  const float dRGB[3] = {(clamp_bits & 1u) ? 0.f : dL_dcolor.x, (clamp_bits & 2u) ? 0.f : dL_dcolor.y,
                         (clamp_bits & 4u) ? 0.f : dL_dcolor.z};
  float ddx[3] = {0, 0, 0}, ddy[3] = {0, 0, 0}, ddz[3] = {0, 0, 0};
#define S(k, ch) sh[(k) * 3 + (ch)]
#define OUT(k, w)                     \
  {                                   \
    const float _w = (w);             \
    dL_dsh[(k) * 3 + 0] = _w * dRGB[0]; \
    dL_dsh[(k) * 3 + 1] = _w * dRGB[1]; \
    dL_dsh[(k) * 3 + 2] = _w * dRGB[2]; \
  }
  // NB: within each degree block the coefficients are READ (into ddx/ddy/ddz) before their gradients are WRITTEN, so
  // dL_dsh may alias sh (the kernel stages both through one LDS row).
  if (deg > 0) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      ddx[ch] = -kSH1 * S(3, ch);
      ddy[ch] = -kSH1 * S(1, ch);
      ddz[ch] = kSH1 * S(2, ch);
    }
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        ddx[ch] += kSH2[0] * y * S(4, ch) + kSH2[2] * 2.f * -x * S(6, ch) + kSH2[3] * z * S(7, ch) + kSH2[4] * 2.f * x * S(8, ch);
        ddy[ch] += kSH2[0] * x * S(4, ch) + kSH2[1] * z * S(5, ch) + kSH2[2] * 2.f * -y * S(6, ch) + kSH2[4] * 2.f * -y * S(8, ch);
        ddz[ch] += kSH2[1] * y * S(5, ch) + kSH2[2] * 2.f * 2.f * z * S(6, ch) + kSH2[3] * x * S(7, ch);
      }
      if (deg > 2) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          ddx[ch] += (kSH3[0] * S(9, ch) * 3.f * 2.f * xy + kSH3[1] * S(10, ch) * yz + kSH3[2] * S(11, ch) * -2.f * xy +
                      kSH3[3] * S(12, ch) * -3.f * 2.f * xz + kSH3[4] * S(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                      kSH3[5] * S(14, ch) * 2.f * xz + kSH3[6] * S(15, ch) * 3.f * (xx - yy));
          ddy[ch] += (kSH3[0] * S(9, ch) * 3.f * (xx - yy) + kSH3[1] * S(10, ch) * xz +
                      kSH3[2] * S(11, ch) * (-3.f * yy + 4.f * zz - xx) + kSH3[3] * S(12, ch) * -3.f * 2.f * yz +
                      kSH3[4] * S(13, ch) * -2.f * xy + kSH3[5] * S(14, ch) * -2.f * yz + kSH3[6] * S(15, ch) * -3.f * 2.f * xy);
          ddz[ch] += (kSH3[1] * S(10, ch) * xy + kSH3[2] * S(11, ch) * 4.f * 2.f * yz +
                      kSH3[3] * S(12, ch) * 3.f * (2.f * zz - xx - yy) + kSH3[4] * S(13, ch) * 4.f * 2.f * xz +
                      kSH3[5] * S(14, ch) * (xx - yy));
        }
      }
    }
  }
  OUT(0, kSH0);
  if (deg > 0) {
    OUT(1, -kSH1 * y);
    OUT(2, kSH1 * z);
    OUT(3, -kSH1 * x);
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      OUT(4, kSH2[0] * xy);
      OUT(5, kSH2[1] * yz);
      OUT(6, kSH2[2] * (2.f * zz - xx - yy));
      OUT(7, kSH2[3] * xz);
      OUT(8, kSH2[4] * (xx - yy));
      if (deg > 2) {
        OUT(9, kSH3[0] * y * (3.f * xx - yy));
        OUT(10, kSH3[1] * xy * z);
        OUT(11, kSH3[2] * y * (4.f * zz - xx - yy));
        OUT(12, kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
        OUT(13, kSH3[4] * x * (4.f * zz - xx - yy));
        OUT(14, kSH3[5] * z * (xx - yy));
        OUT(15, kSH3[6] * x * (xx - 3.f * yy));
      }
    }
  }
#undef S
#undef OUT
  const float dd0 = ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2];
  const float dd1 = ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2];
  const float dd2 = ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2];
  // dnormvdv, CR/auxiliary.h:107-117
  const float sum2 = d0x * d0x + d0y * d0y + d0z * d0z;
  const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  dL_dmean[0] += ((+sum2 - d0x * d0x) * dd0 - d0y * d0x * dd1 - d0z * d0x * dd2) * inv32;
  dL_dmean[1] += (-d0x * d0y * dd0 + (sum2 - d0y * d0y) * dd1 - d0z * d0y * dd2) * inv32;
  dL_dmean[2] += (-d0x * d0z * dd0 - d0y * d0z * dd1 + (sum2 - d0z * d0z) * dd2) * inv32;
}

// ---- the same math coefficient by coefficient, for kernels that stage HALF of a Gaussian's coefficients at a time -----------
// (the preprocess kernels: 24 instead of 48 floats per Gaussian in LDS -> twice the resident workgroups, see geometry.hip).
// sh_weight<K>: the basis weight w_K(x, y, z) with exactly the association of the expressions above (colour = sum_k w_k sh_k
// + 0.5, added in the order k = 0..15, is then bit-identical to sh_to_rgb); sh_weight_grad<K>: d w_K / d(x, y, z).
struct ShDir {
  float x, y, z, xx, yy, zz, xy, yz, xz;
};
__device__ __forceinline__ ShDir sh_dir(const float3 pos, const float *campos, float3 &d0) {
  d0 = make_float3(pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]);
  const float len = sqrtf(d0.x * d0.x + d0.y * d0.y + d0.z * d0.z);
  ShDir d;
  d.x = d0.x / len;
  d.y = d0.y / len;
  d.z = d0.z / len;
  d.xx = d.x * d.x;
  d.yy = d.y * d.y;
  d.zz = d.z * d.z;
  d.xy = d.x * d.y;
  d.yz = d.y * d.z;
  d.xz = d.x * d.z;
  return d;
}
template <int K>
__device__ __forceinline__ float sh_weight(const ShDir &d) {
  if constexpr (K == 0) return kSH0;
  if constexpr (K == 1) return -kSH1 * d.y;
  if constexpr (K == 2) return kSH1 * d.z;
  if constexpr (K == 3) return -kSH1 * d.x;
  if constexpr (K == 4) return kSH2[0] * d.xy;
  if constexpr (K == 5) return kSH2[1] * d.yz;
  if constexpr (K == 6) return kSH2[2] * (2.0f * d.zz - d.xx - d.yy);
  if constexpr (K == 7) return kSH2[3] * d.xz;
  if constexpr (K == 8) return kSH2[4] * (d.xx - d.yy);
  if constexpr (K == 9) return kSH3[0] * d.y * (3.0f * d.xx - d.yy);
  if constexpr (K == 10) return kSH3[1] * d.xy * d.z;
  if constexpr (K == 11) return kSH3[2] * d.y * (4.0f * d.zz - d.xx - d.yy);
  if constexpr (K == 12) return kSH3[3] * d.z * (2.0f * d.zz - 3.0f * d.xx - 3.0f * d.yy);
  if constexpr (K == 13) return kSH3[4] * d.x * (4.0f * d.zz - d.xx - d.yy);
  if constexpr (K == 14) return kSH3[5] * d.z * (d.xx - d.yy);
  if constexpr (K == 15) return kSH3[6] * d.x * (d.xx - 3.0f * d.yy);
  return 0.f;
}
template <int K>
__device__ __forceinline__ float3 sh_weight_grad(const ShDir &d) {
  if constexpr (K == 1) return make_float3(0.f, -kSH1, 0.f);
  if constexpr (K == 2) return make_float3(0.f, 0.f, kSH1);
  if constexpr (K == 3) return make_float3(-kSH1, 0.f, 0.f);
  if constexpr (K == 4) return make_float3(kSH2[0] * d.y, kSH2[0] * d.x, 0.f);
  if constexpr (K == 5) return make_float3(0.f, kSH2[1] * d.z, kSH2[1] * d.y);
  if constexpr (K == 6) return make_float3(kSH2[2] * 2.f * -d.x, kSH2[2] * 2.f * -d.y, kSH2[2] * 2.f * 2.f * d.z);
  if constexpr (K == 7) return make_float3(kSH2[3] * d.z, 0.f, kSH2[3] * d.x);
  if constexpr (K == 8) return make_float3(kSH2[4] * 2.f * d.x, kSH2[4] * 2.f * -d.y, 0.f);
  if constexpr (K == 9) return make_float3(kSH3[0] * 3.f * 2.f * d.xy, kSH3[0] * 3.f * (d.xx - d.yy), 0.f);
  if constexpr (K == 10) return make_float3(kSH3[1] * d.yz, kSH3[1] * d.xz, kSH3[1] * d.xy);
  if constexpr (K == 11)
    return make_float3(kSH3[2] * -2.f * d.xy, kSH3[2] * (-3.f * d.yy + 4.f * d.zz - d.xx), kSH3[2] * 4.f * 2.f * d.yz);
  if constexpr (K == 12)
    return make_float3(kSH3[3] * -3.f * 2.f * d.xz, kSH3[3] * -3.f * 2.f * d.yz, kSH3[3] * 3.f * (2.f * d.zz - d.xx - d.yy));
  if constexpr (K == 13)
    return make_float3(kSH3[4] * (-3.f * d.xx + 4.f * d.zz - d.yy), kSH3[4] * -2.f * d.xy, kSH3[4] * 4.f * 2.f * d.xz);
  if constexpr (K == 14) return make_float3(kSH3[5] * 2.f * d.xz, kSH3[5] * -2.f * d.yz, kSH3[5] * (d.xx - d.yy));
  if constexpr (K == 15) return make_float3(kSH3[6] * 3.f * (d.xx - d.yy), kSH3[6] * -3.f * 2.f * d.xy, 0.f);
  return make_float3(0.f, 0.f, 0.f);
}
constexpr int sh_band_of(int k) { return k == 0 ? 0 : k < 4 ? 1 : k < 9 ? 2 : 3; }

// colour accumulation over the coefficients [LO, HI): part[(k - LO) * 3 + ch] holds coefficient k; res carries over between calls
template <int LO, int HI, int K = LO>
__device__ __forceinline__ void sh_accumulate(int deg, const ShDir &d, const float *part, float (&res)[3]) {
  if constexpr (K < HI) {
    if (deg >= sh_band_of(K)) {
      const float w = sh_weight<K>(d);
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        const float t = w * part[(K - LO) * 3 + ch];
        res[ch] = K == 0 ? t : res[ch] + t;
      }
    }
    sh_accumulate<LO, HI, K + 1>(deg, d, part, res);
  }
}
__device__ __forceinline__ float3 sh_finish(float (&res)[3], uint32_t &clamp_bits) {
  clamp_bits = 0;
#pragma unroll
  for (int ch = 0; ch < 3; ch++) {
    res[ch] += 0.5f;
    if (res[ch] < 0) clamp_bits |= 1u << ch;
    res[ch] = fmaxf(res[ch], 0.0f);
  }
  return make_float3(res[0], res[1], res[2]);
}

// backward over the coefficients [LO, HI), in place: part[] holds the coefficients on entry and their gradients on exit
// (zeros for the bands above deg); dd accumulates d colour / d direction per channel ([0..2] x, [3..5] y, [6..8] z)
template <int LO, int HI, int K = LO>
__device__ __forceinline__ void sh_backward_range(int deg, const ShDir &d, float *part, const float (&dRGB)[3], float (&dd)[9]) {
  if constexpr (K < HI) {
    float *c = part + (K - LO) * 3;
    if (deg >= sh_band_of(K)) {
      const float w = sh_weight<K>(d);
      if constexpr (K > 0) {
        const float3 g = sh_weight_grad<K>(d);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          const float s = c[ch];
          dd[ch] += g.x * s;
          dd[3 + ch] += g.y * s;
          dd[6 + ch] += g.z * s;
        }
      }
#pragma unroll
      for (int ch = 0; ch < 3; ch++) c[ch] = w * dRGB[ch];
    } else {
      c[0] = c[1] = c[2] = 0.f;
    }
    sh_backward_range<LO, HI, K + 1>(deg, d, part, dRGB, dd);
  }
}
// direction gradient -> mean gradient (dnormvdv, CR/auxiliary.h:107-117)
__device__ __forceinline__ void sh_backward_finish(const float3 d0, const float (&dRGB)[3], const float (&dd)[9], float *dL_dmean) {
  const float dd0 = dd[0] * dRGB[0] + dd[1] * dRGB[1] + dd[2] * dRGB[2];
  const float dd1 = dd[3] * dRGB[0] + dd[4] * dRGB[1] + dd[5] * dRGB[2];
  const float dd2 = dd[6] * dRGB[0] + dd[7] * dRGB[1] + dd[8] * dRGB[2];
  const float sum2 = d0.x * d0.x + d0.y * d0.y + d0.z * d0.z;
  const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  dL_dmean[0] += ((+sum2 - d0.x * d0.x) * dd0 - d0.y * d0.x * dd1 - d0.z * d0.x * dd2) * inv32;
  dL_dmean[1] += (-d0.x * d0.y * dd0 + (sum2 - d0.y * d0.y) * dd1 - d0.z * d0.y * dd2) * inv32;
  dL_dmean[2] += (-d0.x * d0.z * dd0 - d0.y * d0.z * dd1 + (sum2 - d0.z * d0.z) * dd2) * inv32;
}

// the basis weights themselves: colour = sum_k w[k] * sh[k] (+0.5), i.e. dL_dsh[k] = w[k] * dL_dRGB (CR/backward.cu:40-116)
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float *w) {
  w[0] = kSH0;
  if (deg > 0) {
    w[1] = -kSH1 * y;
    w[2] = kSH1 * z;
    w[3] = -kSH1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      w[4] = kSH2[0] * xy;
      w[5] = kSH2[1] * yz;
      w[6] = kSH2[2] * (2.f * zz - xx - yy);
      w[7] = kSH2[3] * xz;
      w[8] = kSH2[4] * (xx - yy);
      if (deg > 2) {
        w[9] = kSH3[0] * y * (3.f * xx - yy);
        w[10] = kSH3[1] * xy * z;
        w[11] = kSH3[2] * y * (4.f * zz - xx - yy);
        w[12] = kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
        w[13] = kSH3[4] * x * (4.f * zz - xx - yy);
        w[14] = kSH3[5] * z * (xx - yy);
        w[15] = kSH3[6] * x * (xx - 3.f * yy);
      }
    }
  }
}

}  // namespace gsr
