// Micro-benchmark: issue cost of wave64 instructions on gfx950 when several waves share a SIMD (ns and cycles per
// wave-instruction per SIMD).  Decides which cross-lane primitives the blend-backward reduction may use.
// Not part of the product.  Build: hipcc -O3 --offload-arch=gfx950 -w valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DPPADD(v, ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) x[i] = fmaf(x[i], a, b);
        if (KIND == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);
        if (KIND == 2) { DPPADD(x[i], 0xB1); }    // quad_perm
        if (KIND == 3) { DPPADD(x[i], 0x128); }   // row_ror 8
        if (KIND == 4) x[i] += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x[i]), 0x041F));  // xor 1
        if (KIND == 5) x[i] += __shfl_xor(x[i], 8, 64);  // ds_bpermute
        if (KIND == 6) {
          if (i % 2 == 0) {
            const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[i]), __float_as_uint(x[i + 1]), false, false);
            x[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
          }
        }
        if (KIND == 7) {
          if (i % 2 == 0) {
            const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[i]), __float_as_uint(x[i + 1]), false, false);
            x[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
          }
        }
        if (KIND == 8) x[i] = x[i] > a ? b : x[i];   // v_cmp + v_cndmask
        if (KIND == 9) x[i] += __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[i]), 5) * 1e-30f;  // v_readlane + cvt + fma
        if (KIND == 11) {  // unfused: v_mov_b32_dpp, then a plain add
          float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[i]), 0xB1, 0xF, 0xF, false));
          asm volatile("" : "+v"(t));
          x[i] += t;
        }
        if (KIND == 12) {  // two independent accumulators per dpp source (is the DPP cost per instruction or per dependency?)
          float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[i]), 0x124, 0xF, 0xF, false));
          asm volatile("" : "+v"(t));
          x[i] = x[(i + 1) & 7] + t;
        }
        if (KIND == 13) {  // packed f32 FMA: two floats per lane per instruction (v_pk_fma_f32)
          if (i % 2 == 0) {
            f32x2 v = {x[i], x[i + 1]};
            const f32x2 av = {a, a}, bv = {b, b};
            v = __builtin_elementwise_fma(v, av, bv);
            x[i] = v.x;
            x[i + 1] = v.y;
          }
        }
        if (KIND == 14) {  // packed f32 multiply + add (v_pk_mul_f32, v_pk_add_f32)
          if (i % 2 == 0) {
            f32x2 v = {x[i], x[i + 1]};
            const f32x2 av = {a, a}, bv = {b, b};
            v = v * av;
            asm volatile("" : "+v"(v));
            v = v + bv;
            x[i] = v.x;
            x[i + 1] = v.y;
          }
        }
        if (KIND == 15) x[i] = __builtin_amdgcn_rcpf(x[i]);
        if (KIND == 16) x[i] = fminf(x[i] * a, b);   // v_mul + v_min
        if (KIND == 10) x[i] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x[i]), 0xB1, 0xF, 0xF, false)) ;  // mov_dpp only
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int instr_per_iter) {
  float *out;
  (void)hipMalloc(&out, 256 * 8 * 256 * 4 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
    const int blocks = 256 * wps;  // 256 CUs x (wps blocks of 4 waves)
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * instr_per_iter * wps;
    printf("%-28s waves/SIMD=%d  %8.3f ms  %.2f ns per op-group per SIMD\n", name, wps, ms, ms * 1e6 / n);
  }
}

int main() {
  run<0>("v_fma_f32", 64);
  run<1>("v_exp_f32", 64);
  run<2>("add + dpp quad_perm", 64);
  run<3>("add + dpp row_ror8", 64);
  run<4>("add + ds_swizzle xor1", 64);
  run<5>("add + ds_bpermute", 64);
  run<6>("permlane32_swap + add (per 2)", 32);
  run<7>("permlane16_swap + add (per 2)", 32);
  run<8>("v_cmp + v_cndmask", 64);
  run<9>("v_readlane + fma", 64);
  run<10>("v_mov_dpp only", 64);
  run<11>("mov_dpp ; add (unfused)", 64);
  run<12>("mov_dpp ; add other reg", 64);
  run<13>("v_pk_fma_f32 (per 2 floats)", 32);
  run<14>("v_pk_mul + v_pk_add (per 2)", 32);
  run<15>("v_rcp_f32", 64);
  run<16>("v_mul + v_min", 64);
  return 0;
}
