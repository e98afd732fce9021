// activations.hip -- the parameter activations render() applies every frame, in ONE kernel each way.
//
// The reference reads the Gaussian parameters through property getters (scene/gaussian_model.py:157-199: get_opacity =
// sigmoid(_opacity), get_albedo / get_roughness = sigmoid(_albedo) (sic: both read _albedo, :193-199), get_scaling =
// exp(_scaling), get_rotation = F.normalize(_rotation), get_normal = _normal / |_normal|) and render() adds
// opacity.repeat(1, 3) as the occlusion placeholder (gaussian_renderer/__init__.py:141): a dozen elementwise / reduction
// launches forward and about twenty in autograd's backward, ~110 us of a 1.47 ms frame at 200k Gaussians.  Here: one thread per
// Gaussian, 12 floats in, 17 out; the backward takes the gradients of the six outputs and writes the five raw gradients.
#include "gsr_common.h"

namespace gsr {

struct ActArgs {
  int P;
  const float *opacity_raw, *albedo_raw, *scaling_raw, *rotation_raw, *normal_raw;
  // forward outputs (backward: the saved forward outputs where they shorten the chain rule)
  float *opacity, *albedo, *scaling, *rotation, *normal, *occlusion;
  // backward: incoming / outgoing gradients
  const float *g_opacity, *g_albedo, *g_scaling, *g_rotation, *g_normal, *g_occlusion;
  float *d_opacity_raw, *d_albedo_raw, *d_scaling_raw, *d_rotation_raw, *d_normal_raw;
  const float *acc_rotation_raw;  // backward, optional: a gradient of the RAW rotation that already exists, added to d_rotation_raw
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void activations_forward_kernel(const ActArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.P) return;
  const float o = sigmoidf(a.opacity_raw[i]);
  a.opacity[i] = o;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    a.occlusion[3 * (size_t)i + k] = o;
    a.albedo[3 * (size_t)i + k] = sigmoidf(a.albedo_raw[3 * (size_t)i + k]);
    a.scaling[3 * (size_t)i + k] = expf(a.scaling_raw[3 * (size_t)i + k]);
  }
  float q[4], n[3], s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    q[k] = a.rotation_raw[4 * (size_t)i + k];
    s += q[k] * q[k];
  }
  const float ql = fmaxf(sqrtf(s), 1e-12f);  // F.normalize: x / max(|x|, eps)
#pragma unroll
  for (int k = 0; k < 4; k++) a.rotation[4 * (size_t)i + k] = q[k] / ql;
  s = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    n[k] = a.normal_raw[3 * (size_t)i + k];
    s += n[k] * n[k];
  }
  const float nl = sqrtf(s);  // get_normal divides by the plain norm (:176-178)
#pragma unroll
  for (int k = 0; k < 3; k++) a.normal[3 * (size_t)i + k] = n[k] / nl;
}

// x -> x / |x|: dx = (g - n (n . g)) / |x|   (below F.normalize's eps the output is x / eps: dx = g / eps)
template <int N>
__device__ __forceinline__ void unit_backward(const float *x, const float *g, float eps, float *dx) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < N; k++) s += x[k] * x[k];
  const float len = sqrtf(s);
  if (len < eps) {
#pragma unroll
    for (int k = 0; k < N; k++) dx[k] = g[k] / eps;
    return;
  }
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < N; k++) dot += (x[k] / len) * g[k];
#pragma unroll
  for (int k = 0; k < N; k++) dx[k] = (g[k] - (x[k] / len) * dot) / len;
}

__global__ __launch_bounds__(256) void activations_backward_kernel(const ActArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.P) return;
  // a null incoming gradient = that output did not reach the loss
  const float o = a.opacity[i];
  float go = a.g_opacity ? a.g_opacity[i] : 0.f;
  if (a.g_occlusion) go += a.g_occlusion[3 * (size_t)i] + a.g_occlusion[3 * (size_t)i + 1] + a.g_occlusion[3 * (size_t)i + 2];
  a.d_opacity_raw[i] = go * o * (1.f - o);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const size_t j = 3 * (size_t)i + k;
    const float al = a.albedo[j];
    a.d_albedo_raw[j] = a.g_albedo ? a.g_albedo[j] * al * (1.f - al) : 0.f;
    a.d_scaling_raw[j] = a.g_scaling ? a.g_scaling[j] * a.scaling[j] : 0.f;
  }
  float x[4], g[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    x[k] = a.rotation_raw[4 * (size_t)i + k];
    g[k] = a.g_rotation ? a.g_rotation[4 * (size_t)i + k] : 0.f;
  }
  unit_backward<4>(x, g, 1e-12f, d);
#pragma unroll
  for (int k = 0; k < 4; k++) a.d_rotation_raw[4 * (size_t)i + k] = d[k] + (a.acc_rotation_raw ? a.acc_rotation_raw[4 * (size_t)i + k] : 0.f);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    x[k] = a.normal_raw[3 * (size_t)i + k];
    g[k] = a.g_normal ? a.g_normal[3 * (size_t)i + k] : 0.f;
  }
  unit_backward<3>(x, g, 0.f, d);
#pragma unroll
  for (int k = 0; k < 3; k++) a.d_normal_raw[3 * (size_t)i + k] = d[k];
}

}  // namespace gsr

extern "C" int gsr_model_activations_forward(int P, const float *opacity_raw, const float *albedo_raw, const float *scaling_raw,
                                             const float *rotation_raw, const float *normal_raw, float *opacity, float *albedo,
                                             float *scaling, float *rotation, float *normal, float *occlusion,
                                             gsr_stream_t stream_) {
  using namespace gsr;
  if (P < 0 || (P > 0 && (!opacity_raw || !albedo_raw || !scaling_raw || !rotation_raw || !normal_raw || !opacity || !albedo ||
                          !scaling || !rotation || !normal || !occlusion))) {
    set_error("gsr_model_activations_forward: bad size or null pointer");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  ActArgs a = {};
  a.P = P;
  a.opacity_raw = opacity_raw, a.albedo_raw = albedo_raw, a.scaling_raw = scaling_raw, a.rotation_raw = rotation_raw;
  a.normal_raw = normal_raw;
  a.opacity = opacity, a.albedo = albedo, a.scaling = scaling, a.rotation = rotation, a.normal = normal, a.occlusion = occlusion;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(activations_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, a);
  return check_hip(hipGetLastError(), "activations_forward_kernel", __FILE__, __LINE__);
}

extern "C" int gsr_model_activations_backward_acc(int P, const float *rotation_raw, const float *normal_raw, const float *opacity,
                                              const float *albedo, const float *scaling, const float *g_opacity,
                                              const float *g_albedo, const float *g_scaling, const float *g_rotation,
                                              const float *g_normal, const float *g_occlusion, float *d_opacity_raw,
                                              float *d_albedo_raw, float *d_scaling_raw, float *d_rotation_raw,
                                              float *d_normal_raw, const float *acc_drotation_raw, gsr_stream_t stream_) {
  using namespace gsr;
  if (P < 0 || (P > 0 && (!rotation_raw || !normal_raw || !opacity || !albedo || !scaling || !d_opacity_raw || !d_albedo_raw ||
                          !d_scaling_raw || !d_rotation_raw || !d_normal_raw))) {
    set_error("gsr_model_activations_backward: bad size or null pointer");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  ActArgs a = {};
  a.P = P;
  a.rotation_raw = rotation_raw, a.normal_raw = normal_raw;
  a.opacity = const_cast<float *>(opacity), a.albedo = const_cast<float *>(albedo), a.scaling = const_cast<float *>(scaling);
  a.g_opacity = g_opacity, a.g_albedo = g_albedo, a.g_scaling = g_scaling, a.g_rotation = g_rotation, a.g_normal = g_normal;
  a.g_occlusion = g_occlusion;
  a.d_opacity_raw = d_opacity_raw, a.d_albedo_raw = d_albedo_raw, a.d_scaling_raw = d_scaling_raw;
  a.d_rotation_raw = d_rotation_raw, a.d_normal_raw = d_normal_raw;
  a.acc_rotation_raw = acc_drotation_raw;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(activations_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, a);
  return check_hip(hipGetLastError(), "activations_backward_kernel", __FILE__, __LINE__);
}

extern "C" int gsr_model_activations_backward(int P, const float *rotation_raw, const float *normal_raw, const float *opacity,
                                              const float *albedo, const float *scaling, const float *g_opacity,
                                              const float *g_albedo, const float *g_scaling, const float *g_rotation,
                                              const float *g_normal, const float *g_occlusion, float *d_opacity_raw,
                                              float *d_albedo_raw, float *d_scaling_raw, float *d_rotation_raw,
                                              float *d_normal_raw, gsr_stream_t stream_) {
  return gsr_model_activations_backward_acc(P, rotation_raw, normal_raw, opacity, albedo, scaling, g_opacity, g_albedo, g_scaling,
                                            g_rotation, g_normal, g_occlusion, d_opacity_raw, d_albedo_raw, d_scaling_raw,
                                            d_rotation_raw, d_normal_raw, nullptr, stream_);
}
