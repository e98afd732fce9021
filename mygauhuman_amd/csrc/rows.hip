// rows.hip -- fused row gather over a structure of arrays (densify / prune / optimizer-state surgery).
//
// The reference prunes and grows its Gaussian set with boolean-mask indexing and torch.cat on every parameter tensor and on
// both Adam moments of each (`_prune_optimizer`, `cat_tensors_to_optimizer`, scene/gaussian_model.py:421-487): 9 parameters
// x 3 tensors + 3 statistics = 30 tensors, each costing a nonzero + index kernel pair and a host synchronisation for the
// count.  Here the new row set is described once by an int32 source-row index (negative = a zero row, what a freshly
// created Gaussian's Adam moments are) and ONE kernel moves every array: workgroup = 256 output rows x all arrays, rows are
// copied with coalesced 4-byte accesses (row sizes here are 1..45 floats, not multiples of 4).
#include "gsr_common.h"

namespace gsr {

constexpr int ROWS_MAX_ARRAYS = 32;
struct RowsArgs {
  int n_arrays, n_out;
  const int *index;
  const float *src[ROWS_MAX_ARRAYS];
  float *dst[ROWS_MAX_ARRAYS];
  int width[ROWS_MAX_ARRAYS];
  int zero_new[ROWS_MAX_ARRAYS];  // 1: rows whose index is flagged "new" (bit 30 set) become zero rows in this array
};
constexpr int ROW_NEW_FLAG = 1 << 30;

__global__ __launch_bounds__(256) void gather_rows_kernel(const RowsArgs a) {
  __shared__ int s_idx[256];
  const int first = blockIdx.x * 256;
  const int nrows = min(256, a.n_out - first);
  if ((int)threadIdx.x < nrows) s_idx[threadIdx.x] = a.index[first + threadIdx.x];
  __syncthreads();
  for (int k = 0; k < a.n_arrays; k++) {
    const int w = a.width[k];
    const float *src = a.src[k];
    float *dst = a.dst[k] + (size_t)first * w;
    const bool zn = a.zero_new[k] != 0;
    for (int e = threadIdx.x; e < nrows * w; e += 256) {
      const int r = e / w, c = e - r * w;
      const int s = s_idx[r];
      const bool is_new = (s & ROW_NEW_FLAG) != 0 && s >= 0;
      const int row = s & (ROW_NEW_FLAG - 1);
      dst[e] = (s < 0 || (zn && is_new)) ? 0.f : src[(size_t)row * w + c];
    }
  }
}

}  // namespace gsr

extern "C" int gsr_gather_rows(int n_arrays, const float *const *src, float *const *dst, const int *row_floats,
                               const int *zero_new, int n_out, const int *index, gsr_stream_t stream_) {
  using namespace gsr;
  if (n_arrays < 1 || n_arrays > ROWS_MAX_ARRAYS || n_out < 0 || !src || !dst || !row_floats || (n_out > 0 && !index)) {
    set_error("gsr_gather_rows: bad arguments (1..%d arrays)", ROWS_MAX_ARRAYS);
    return GSR_EINVAL;
  }
  if (n_out == 0) return GSR_OK;
  RowsArgs a = {};
  a.n_arrays = n_arrays;
  a.n_out = n_out;
  a.index = index;
  for (int k = 0; k < n_arrays; k++) {
    if (!src[k] || !dst[k] || row_floats[k] < 1) {
      set_error("gsr_gather_rows: array %d has a null pointer or a row size < 1", k);
      return GSR_EINVAL;
    }
    a.src[k] = src[k];
    a.dst[k] = dst[k];
    a.width[k] = row_floats[k];
    a.zero_new[k] = zero_new ? zero_new[k] : 0;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n_out + 255) / 256), dim3(256), 0, stream, a);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}
