// Micro-benchmark: do MFMA and VALU instructions of two waves on ONE SIMD overlap on gfx950, for the f32-input MFMA
// (v_mfma_f32_16x16x4_f32) and for the bf16 MFMA (v_mfma_f32_16x16x32_bf16)?  512-thread workgroups put two waves on each
// SIMD (waves w and w+4); role 0 = every wave runs the VALU loop, 1 = every wave runs the MFMA loop, 2 = waves 0-3 MFMA and
// waves 4-7 VALU (same per-wave iteration counts).  If the pipes are separate, time(2) ~ max(time(0), time(1)) / 2-ish;
// if the f32 MFMA executes on the vector ALUs, time(2) ~ (time(0) + time(1)) / 2.
// Not part of the product.  Build: hipcc -O3 --offload-arch=gfx950 -w mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MFMA_KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters, int role, float a, float b) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = role == 1 || (role == 2 && wave < 4);
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
  if (do_mfma) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (MFMA_KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i], x[i + 4], acc[i], 0, 0, 0);
          if (MFMA_KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
        }
      }
    }
  } else {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = fmaf(x[i], a, b);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += x[i];
#pragma unroll
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MFMA_KIND>
void run(const char *name) {
  float *out;
  (void)hipMalloc(&out, 256 * 512 * 4 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 4000;
  for (int role = 0; role < 3; role++) {
    hipLaunchKernelGGL(k<MFMA_KIND>, dim3(256), dim3(512), 0, 0, out, 10, role, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MFMA_KIND>, dim3(256), dim3(512), 0, 0, out, iters, role, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per wave: role 0: iters*128 FMAs; role 1: iters*16 MFMAs
    printf("%-22s role %d (%s): %8.3f ms   [per wave: %d FMA or %d MFMA]\n", name, role,
           role == 0 ? "all VALU" : (role == 1 ? "all MFMA" : "4 MFMA + 4 VALU waves"), ms, iters * 128, iters * 16);
  }
}

int main() {
  run<0>("mfma_f32_16x16x4_f32");
  run<1>("mfma_f32_16x16x32_bf16");
  return 0;
}
