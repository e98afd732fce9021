// lds_dma_probe.hip -- does gfx950's global -> LDS DMA (global_load_lds_dwordx4 / _dword) do what blend_fwd's feature staging needs?
//   (a) lane i's data lands at M0.base + 16 i (x4) / + 4 i (dword), (b) lanes that are masked off leave their cells alone,
//   (c) a source that is only 8-byte aligned (72-byte feature rows) is fine, (d) s_waitcnt vmcnt(0) is the completion fence,
//   (e) two buffers (two M0 values) in flight at once.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/lds_dma_probe tools/ubench/lds_dma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const void *g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void dma4(const void *g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds_base) : "memory");
}

__global__ void probe(const float *table, const int *ids, unsigned long long mask, float *out) {
  __shared__ __attribute__((aligned(16))) float buf[2][5][64][4];   // [buffer][piece][lane][4 floats]; piece 4: two dword planes
  const int lane = threadIdx.x;
  for (int e = lane; e < 2 * 5 * 64 * 4; e += 64) (&buf[0][0][0][0])[e] = -1.0f;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(&buf[0][0][0][0]);    // LDS byte address (group segment offset)
  const float *row = table + (size_t)ids[lane] * 18;                 // 72-byte rows: 8-byte aligned only
  if ((mask >> lane) & 1ull) {
    for (int b = 0; b < 2; b++) {
      const float *r = row + (b ? 18 * 1000 : 0);                    // second buffer: rows of another table half
      for (int q = 0; q < 4; q++) dma16(r + 4 * q, base + (uint32_t)(((b * 5 + q) * 64) * 16));
      dma4(r + 16, base + (uint32_t)(((b * 5 + 4) * 64) * 16));
      dma4(r + 17, base + (uint32_t)(((b * 5 + 4) * 64) * 16 + 64 * 4));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int b = 0; b < 2; b++) {
    for (int q = 0; q < 4; q++)
      for (int k = 0; k < 4; k++) out[((b * 64 + lane) * 18) + 4 * q + k] = buf[b][q][lane][k];
    const float *tail = &buf[b][4][0][0];
    out[((b * 64 + lane) * 18) + 16] = tail[lane];
    out[((b * 64 + lane) * 18) + 17] = tail[64 + lane];
  }
}

int main() {
  const int ROWS = 2000;
  std::vector<float> table((size_t)ROWS * 18);
  for (size_t i = 0; i < table.size(); i++) table[i] = (float)i;
  std::vector<int> ids(64);
  for (int i = 0; i < 64; i++) ids[i] = (i * 37 + 5) % 1000;
  const unsigned long long mask = 0xF0F0A5A5DEADBEEFull;
  float *d_table, *d_out;
  int *d_ids;
  hipMalloc(&d_table, table.size() * 4);
  hipMalloc(&d_out, 2 * 64 * 18 * 4);
  hipMalloc(&d_ids, 64 * 4);
  hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_ids, ids.data(), 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_table, d_ids, mask, d_out);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
    return 2;
  }
  std::vector<float> out(2 * 64 * 18);
  hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < 2; b++)
    for (int l = 0; l < 64; l++)
      for (int k = 0; k < 18; k++) {
        const bool on = (mask >> l) & 1ull;
        const float want = on ? table[(size_t)(ids[l] + (b ? 1000 : 0)) * 18 + k] : -1.0f;
        const float got = out[(b * 64 + l) * 18 + k];
        if (got != want && bad++ < 10) printf("mismatch buffer %d lane %d k %d: got %g want %g\n", b, l, k, got, want);
      }
  printf("lds_dma_probe: %s (%d mismatches)\n", bad ? "FAILED" : "OK", bad);
  return bad ? 1 : 0;
}
