// lbs.hip -- SMPL linear-blend-skinning kernels (placeholder until the HIP kernels land; fails loudly).
#include "gsr_common.h"

extern "C" {
int gsr_lbs_forward(int, int, const float *, const float *, const float *, const float *, const float *, const float *,
                    const float *, const float *, const float *, const float *, const float *, const float *, int *, float *,
                    float *, float *, float *, float *, float *, gsr_stream_t) {
  gsr::set_error("gsr_lbs_forward: not available in this build");
  return GSR_EINVAL;
}
int gsr_lbs_backward(int, int, const float *, const float *, const int *, const float *, const float *, const float *,
                     const float *, const float *, const float *, const float *, const float *, const float *, const float *,
                     const float *, float *, float *, float *, float *, float *, gsr_stream_t) {
  gsr::set_error("gsr_lbs_backward: not available in this build");
  return GSR_EINVAL;
}
}
