// probe.hip -- measurement only: what shader clock is this GPU running at right now?
//
// The blend kernels are bound by vector-instruction ISSUE, so their time scales with the shader clock, and the shader clock of an
// MI355X that has just been handed a process is not the clock it sustains: the first tens of milliseconds of a run execute 10-15 %
// below it (round 4: `bench.py --steps 20 --warmup 5` read 2,285-2,334 frames/s and a 300-step run 2,546 on the same box, every
// VALU-bound stage slower by the same factor, the HBM-bound ones unchanged).  A bench line therefore has to say at what clock it
// was taken, and a short run has to get the chip to its sustained clock before its warm-up steps.
//
// One wave per workgroup runs a dependent chain of `fmas` v_fma_f32 (something for the SIMDs to do: the clock is a function of
// load) bracketed by two counters: s_memrealtime (constant 100 MHz) and s_memtime, which on gfx950 ticks with the shader clock
// (measured: 2.26 ticks per ns on a cold device, 2.39-2.40 after 20 ms of load; tools/clock_ramp.py).
//   clock = s_memtime ticks / (100 MHz ticks x 10 ns).
// The chain itself runs at 8.8 cycles per dependent FMA whatever the clock: reported as a sanity value.
#include "gsr_common.h"

namespace gsr {

__global__ __launch_bounds__(WAVE) void clock_probe_kernel(unsigned long long *out, int fmas, float a, float b) {
  float x = (float)threadIdx.x;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  asm volatile("" : "+v"(x));
  for (int i = 0; i < fmas; i += 32) {
#pragma unroll
    for (int u = 0; u < 32; u++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  }
  asm volatile("" : "+v"(x));
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2 + 0] = r1 - r0;
    out[blockIdx.x * 2 + 1] = c1 - c0;
  }
  if (x == 12345.678f) out[0] = 0;  // (keeps the chain alive)
}

}  // namespace gsr

extern "C" int gsr_debug_clock_probe(int workgroups, int fmas, unsigned long long *dev_out, void *stream_) {
  using namespace gsr;
  if (workgroups < 1 || workgroups > 2048 || fmas < 32 || !dev_out) {
    set_error("gsr_debug_clock_probe: 1 .. 2048 workgroups, >= 32 FMAs and an output buffer of 2 words per workgroup are required");
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(clock_probe_kernel, dim3(workgroups), dim3(WAVE), 0, stream, dev_out, (fmas + 31) / 32 * 32, 0.9999999f, 1e-7f);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}
