// ssim.hip -- fused SSIM (11x11 Gaussian window, sigma 1.5, zero padding) forward and backward.
//
// Reference: utils/loss_utils.py:25-66 -- five grouped conv2d calls (mu1, mu2, E[x^2], E[y^2], E[xy]) plus ~15 elementwise
// kernels forward, and autograd's transposed convolutions backward, on the image crop of train.py:268-287.  Here:
//   forward : ONE kernel per 16x16 tile -- the 26x26 halo of both images goes through LDS once, the window is applied
//             separably (11 taps along x into LDS, 11 along y in registers) to the five quantities at once, then the SSIM
//             map and the three per-pixel derivative maps the backward needs:
//               A = df/dmu1 - 2 mu1 df/ds1 - mu2 df/ds12,   B = df/ds1,   C = df/ds12
//             with f = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2));
//   backward: ONE kernel -- dL/dimg1(p) = conv(g A)(p) + 2 img1(p) conv(g B)(p) + img2(p) conv(g C)(p), g = dL/dmap, the same
//             separable window over the three maps (the window is symmetric, zero padding on both sides).
#include "gsr_common.h"

namespace gsr {

constexpr int SS_T = 16, SS_R = 5, SS_IN = SS_T + 2 * SS_R;  // tile, window radius, staged extent (26)

struct SsimWindow {
  float w[11];
};
// gaussian(11, 1.5) normalised (utils/loss_utils.py:25-27), evaluated in double on the host like the reference's Python floats
static SsimWindow make_window() {
  SsimWindow s;
  double g[11], sum = 0.0;
  for (int i = 0; i < 11; i++) {
    g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
    sum += g[i];
  }
  // the reference builds a float32 tensor of the unnormalised values, then divides by their float32 sum
  float gf[11], sf = 0.f;
  for (int i = 0; i < 11; i++) gf[i] = (float)g[i];
  for (int i = 0; i < 11; i++) sf += gf[i];
  for (int i = 0; i < 11; i++) s.w[i] = gf[i] / sf;
  (void)sum;
  return s;
}

__global__ __launch_bounds__(SS_T *SS_T) void ssim_forward_kernel(int H, int W, const float *img1, const float *img2, SsimWindow win,
                                                                  float *map, float *dA, float *dB, float *dC) {
  __shared__ float s1[SS_IN][SS_IN + 1], s2[SS_IN][SS_IN + 1];
  __shared__ float h[5][SS_IN][SS_T + 1];  // horizontally filtered rows: x, y, xx, yy, xy
  const size_t plane = (size_t)H * W;
  const float *p1 = img1 + blockIdx.z * plane, *p2 = img2 + blockIdx.z * plane;
  const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
  const int t = threadIdx.y * SS_T + threadIdx.x;
  for (int e = t; e < SS_IN * SS_IN; e += SS_T * SS_T) {
    const int ly = e / SS_IN, lx = e % SS_IN;
    const int gx = x0 + lx - SS_R, gy = y0 + ly - SS_R;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
    s1[ly][lx] = in ? p1[(size_t)gy * W + gx] : 0.f;
    s2[ly][lx] = in ? p2[(size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int e = t; e < SS_IN * SS_T; e += SS_T * SS_T) {
    const int ly = e / SS_T, lx = e % SS_T;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float u = s1[ly][lx + k], v = s2[ly][lx + k], wk = win.w[k];
      a += wk * u;
      b += wk * v;
      aa += wk * (u * u);
      bb += wk * (v * v);
      ab += wk * (u * v);
    }
    h[0][ly][lx] = a;
    h[1][ly][lx] = b;
    h[2][ly][lx] = aa;
    h[3][ly][lx] = bb;
    h[4][ly][lx] = ab;
  }
  __syncthreads();
  const int px = x0 + threadIdx.x, py = y0 + threadIdx.y;
  if (px >= W || py >= H) return;
  float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; k++) {
    const float wk = win.w[k];
    mu1 += wk * h[0][threadIdx.y + k][threadIdx.x];
    mu2 += wk * h[1][threadIdx.y + k][threadIdx.x];
    e11 += wk * h[2][threadIdx.y + k][threadIdx.x];
    e22 += wk * h[3][threadIdx.y + k][threadIdx.x];
    e12 += wk * h[4][threadIdx.y + k][threadIdx.x];
  }
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
  const float sg1 = e11 - mu1_sq, sg2 = e22 - mu2_sq, sg12 = e12 - mu12;
  const float a = 2.f * mu12 + C1, b = 2.f * sg12 + C2, c = mu1_sq + mu2_sq + C1, d = sg1 + sg2 + C2;
  const float f = (a * b) / (c * d);
  const size_t o = blockIdx.z * plane + (size_t)py * W + px;
  if (map) map[o] = f;
  if (dA) {
    const float df_dmu1 = (2.f * mu2 * b) / (c * d) - f * (2.f * mu1) / c;
    const float df_ds1 = -f / d;
    const float df_ds12 = (2.f * a) / (c * d);
    dA[o] = df_dmu1 - 2.f * mu1 * df_ds1 - mu2 * df_ds12;
    dB[o] = df_ds1;
    dC[o] = df_ds12;
  }
}

__global__ __launch_bounds__(SS_T *SS_T) void ssim_backward_kernel(int H, int W, const float *img1, const float *img2, SsimWindow win,
                                                                   const float *dL_dmap, float g_scalar, const float *dA,
                                                                   const float *dB, const float *dC, float *dL_dimg1) {
  __shared__ float s[3][SS_IN][SS_IN + 1];
  __shared__ float h[3][SS_IN][SS_T + 1];
  const size_t plane = (size_t)H * W, base = blockIdx.z * plane;
  const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
  const int t = threadIdx.y * SS_T + threadIdx.x;
  for (int e = t; e < SS_IN * SS_IN; e += SS_T * SS_T) {
    const int ly = e / SS_IN, lx = e % SS_IN;
    const int gx = x0 + lx - SS_R, gy = y0 + ly - SS_R;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
    const size_t o = base + (size_t)gy * W + gx;
    const float g = in ? (dL_dmap ? dL_dmap[o] : g_scalar) : 0.f;
    s[0][ly][lx] = in ? g * dA[o] : 0.f;
    s[1][ly][lx] = in ? g * dB[o] : 0.f;
    s[2][ly][lx] = in ? g * dC[o] : 0.f;
  }
  __syncthreads();
  for (int e = t; e < SS_IN * SS_T; e += SS_T * SS_T) {
    const int ly = e / SS_T, lx = e % SS_T;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float wk = win.w[k];
      a += wk * s[0][ly][lx + k];
      b += wk * s[1][ly][lx + k];
      c += wk * s[2][ly][lx + k];
    }
    h[0][ly][lx] = a;
    h[1][ly][lx] = b;
    h[2][ly][lx] = c;
  }
  __syncthreads();
  const int px = x0 + threadIdx.x, py = y0 + threadIdx.y;
  if (px >= W || py >= H) return;
  float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
  for (int k = 0; k < 11; k++) {
    const float wk = win.w[k];
    a += wk * h[0][threadIdx.y + k][threadIdx.x];
    b += wk * h[1][threadIdx.y + k][threadIdx.x];
    c += wk * h[2][threadIdx.y + k][threadIdx.x];
  }
  const size_t o = base + (size_t)py * W + px;
  dL_dimg1[o] = a + 2.f * img1[o] * b + img2[o] * c;
}

}  // namespace gsr

extern "C" {

int gsr_ssim_forward(int planes, int height, int width, const float *img1, const float *img2, float *ssim_map, float *dA,
                     float *dB, float *dC, gsr_stream_t stream_) {
  using namespace gsr;
  if (planes < 0 || height <= 0 || width <= 0 || (planes > 0 && (!img1 || !img2)) || ((dA || dB || dC) && !(dA && dB && dC)) ||
      planes > 65535) {
    set_error("gsr_ssim_forward: bad arguments");
    return GSR_EINVAL;
  }
  if (planes == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const dim3 grid((width + SS_T - 1) / SS_T, (height + SS_T - 1) / SS_T, planes), block(SS_T, SS_T);
  hipLaunchKernelGGL(ssim_forward_kernel, grid, block, 0, stream, height, width, img1, img2, make_window(), ssim_map, dA, dB, dC);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_ssim_backward(int planes, int height, int width, const float *img1, const float *img2, const float *dL_dmap,
                      float dL_dmap_scalar, const float *dA, const float *dB, const float *dC, float *dL_dimg1,
                      gsr_stream_t stream_) {
  using namespace gsr;
  if (planes < 0 || height <= 0 || width <= 0 || (planes > 0 && (!img1 || !img2 || !dA || !dB || !dC || !dL_dimg1)) ||
      planes > 65535) {
    set_error("gsr_ssim_backward: bad arguments");
    return GSR_EINVAL;
  }
  if (planes == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const dim3 grid((width + SS_T - 1) / SS_T, (height + SS_T - 1) / SS_T, planes), block(SS_T, SS_T);
  hipLaunchKernelGGL(ssim_backward_kernel, grid, block, 0, stream, height, width, img1, img2, make_window(), dL_dmap,
                     dL_dmap_scalar, dA, dB, dC, dL_dimg1);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

}  // extern "C"
