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
