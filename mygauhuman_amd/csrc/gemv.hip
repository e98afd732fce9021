// gemv.hip -- row-major matrix-vector products for the SMPL pose blend shapes: offsets[V*3] = posedirs[V*3][207] . feat[207]
// (scene/gaussian_model.py:805-811,827-839).  rocBLAS runs this 17 MB GEMV at ~280 GB/s (60 us, twice per frame); it is
// a pure HBM stream: one wave per run of rows, lane k reads elements k, k+64, ... of a row (coalesced 256-B segments at any
// row alignment -- 207 is odd), a wave reduction per row.  The transposed product (the backward w.r.t. feat) keeps
// per-lane column partials over the wave's rows and flushes them with one atomic per column per wave.
#include "gsr_common.h"

namespace gsr {

constexpr int GEMV_KMAX = 256;       // columns handled (4 per lane)
constexpr int GEMV_ROWS = 4;         // rows per wave, all loaded before the first reduction (16 loads in flight per lane)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
  return v;
}

__global__ __launch_bounds__(256) void gemv_rows_kernel(int R, int K, const float *mat, const float *vec, float *out) {
  const int lane = threadIdx.x % WAVE, wave = (blockIdx.x * 256 + threadIdx.x) / WAVE;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; j++) v[j] = (lane + 64 * j) < K ? vec[lane + 64 * j] : 0.f;
  const int r0 = wave * GEMV_ROWS;
  if (r0 >= R) return;
  float m[GEMV_ROWS][4];
#pragma unroll
  for (int u = 0; u < GEMV_ROWS; u++) {
    const float *row = mat + (size_t)min(r0 + u, R - 1) * K;
#pragma unroll
    for (int j = 0; j < 4; j++) m[u][j] = (lane + 64 * j < K) ? row[lane + 64 * j] : 0.f;
  }
  float mine = 0.f;
#pragma unroll
  for (int u = 0; u < GEMV_ROWS; u++) {
    const float acc = wave_sum((m[u][0] * v[0] + m[u][1] * v[1]) + (m[u][2] * v[2] + m[u][3] * v[3]));
    mine = lane == u ? acc : mine;
  }
  if (lane < GEMV_ROWS && r0 + lane < R) out[r0 + lane] = mine;
}

// dvec[k] += sum_r dout[r] * mat[r][k]   (dvec zero-initialised by the caller)
// A fixed grid of workgroups (one per CU) strides over the rows, every wave with FOUR rows in flight (16 loads per lane); the four
// waves' column partials meet in LDS and the workgroup issues one atomic per column.  (Round 2's version gave each wave 48
// consecutive rows one after the other and let every wave add its 207 partials to the same 207 addresses: 82 us for this 17 MB
// stream -- latency of 48 dependent load rounds plus 89k same-address atomics; this one: see profiles/r3c_render_kernels.txt.)
__global__ __launch_bounds__(256) void gemv_rows_t_kernel(int R, int K, const float *mat, const float *dout, float *dvec) {
  __shared__ float s_part[4][GEMV_KMAX];
  const int lane = threadIdx.x % WAVE, wv = threadIdx.x / WAVE;
  const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = wave * GEMV_ROWS; r0 < R; r0 += n_waves * GEMV_ROWS) {
    float m[GEMV_ROWS][4], g[GEMV_ROWS];
#pragma unroll
    for (int u = 0; u < GEMV_ROWS; u++) {
      const int r = min(r0 + u, R - 1);
      const float *row = mat + (size_t)r * K;
      g[u] = r0 + u < R ? dout[r] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++) m[u][j] = (lane + 64 * j < K) ? row[lane + 64 * j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < GEMV_ROWS; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] += g[u] * m[u][j];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) s_part[wv][lane + 64 * j] = acc[j];
  __syncthreads();
  const int k = threadIdx.x;
  if (k < K) atomicAdd(&dvec[k], (s_part[0][k] + s_part[1][k]) + (s_part[2][k] + s_part[3][k]));
}

}  // namespace gsr

extern "C" {

int gsr_gemv_rows(int rows, int cols, const float *mat, const float *vec, float *out, gsr_stream_t stream_) {
  using namespace gsr;
  if (rows < 0 || cols < 1 || cols > GEMV_KMAX || (rows > 0 && (!mat || !vec || !out))) {
    set_error("gsr_gemv_rows: bad arguments (1..%d columns)", GEMV_KMAX);
    return GSR_EINVAL;
  }
  if (rows == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int waves = (rows + GEMV_ROWS - 1) / GEMV_ROWS;
  hipLaunchKernelGGL(gemv_rows_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, rows, cols, mat, vec, out);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_gemv_rows_t(int rows, int cols, const float *mat, const float *dout, float *dvec, gsr_stream_t stream_) {
  using namespace gsr;
  if (rows < 0 || cols < 1 || cols > GEMV_KMAX || (rows > 0 && (!mat || !dout)) || !dvec) {
    set_error("gsr_gemv_rows_t: bad arguments (1..%d columns)", GEMV_KMAX);
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  GSR_HIP(zero_async(dvec, sizeof(float) * cols, stream));
  if (rows == 0) return GSR_OK;
  const int groups = (rows + 4 * GEMV_ROWS - 1) / (4 * GEMV_ROWS);  // workgroups that would get one round of rows each
  hipLaunchKernelGGL(gemv_rows_t_kernel, dim3(groups < 256 ? groups : 256), dim3(256), 0, stream, rows, cols, mat, dout, dvec);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

}  // extern "C"
