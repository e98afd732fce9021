// hip_stub.cpp -- a HOST-ONLY stand-in for the handful of HIP runtime entry points libgsr's host code calls
// (nm -D --undefined-only libgsr.so), so that every .hip file of the library can be compiled with
// `hipcc --offload-host-only -fsanitize=...` and its host side (argument validation, arena carving, option map,
// profile store, error strings, workspace arithmetic, deterministic-mode malloc/free) can run under ASan/UBSan/TSan on a
// machine without a GPU.  "Device memory" is plain malloc memory, kernels are never executed (hipLaunchKernel only
// validates the launch geometry), stream order is program order.  Test infrastructure only (tests/test_host_layer_sanitized.py).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local hipError_t t_last = hipSuccess;
static std::atomic<long> g_launches{0}, g_bad_launches{0};
static std::atomic<uint32_t> g_fake_R{0xFFFFFFFFu};

extern "C" {

// test hooks
void hipstub_set_fake_readback(uint32_t R) { g_fake_R = R; }  // 0xFFFFFFFF = off: copy real bytes
long hipstub_launches(void) { return g_launches; }
long hipstub_bad_launches(void) { return g_bad_launches; }

struct StubConfig {
  dim3 grid, block;
  size_t shmem;
  hipStream_t stream;
};
static thread_local StubConfig t_cfg;

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
  t_cfg = {grid, block, shmem, stream};
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
  *grid = t_cfg.grid;
  *block = t_cfg.block;
  *shmem = t_cfg.shmem;
  *stream = t_cfg.stream;
  return hipSuccess;
}
void **__hipRegisterFatBinary(const void *) {
  static void *handle[1];
  return handle;
}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}

hipError_t hipLaunchKernel(const void *f, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t) {
  g_launches++;
  const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
  const bool ok = f && args && grid.x > 0 && grid.y > 0 && grid.z > 0 && threads > 0 && threads <= 1024 && grid.y <= 65535 &&
                  grid.z <= 65535 && (unsigned long long)grid.x * block.x <= 0xFFFFFFFFull && shmem <= 160 * 1024;
  if (!ok) {
    g_bad_launches++;
    fprintf(stderr, "hip_stub: invalid launch grid (%u,%u,%u) block (%u,%u,%u) shmem %zu\n", grid.x, grid.y, grid.z, block.x,
            block.y, block.z, shmem);
    t_last = hipErrorInvalidConfiguration;
    return hipErrorInvalidConfiguration;
  }
  return hipSuccess;
}

hipError_t hipGetLastError(void) {
  hipError_t e = t_last;
  t_last = hipSuccess;
  return e;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
hipError_t hipGetDevice(int *d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t attr, int) {
  *v = attr == hipDeviceAttributeMultiprocessorCount ? 256 : 64;
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) {
  *p = malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
  *p = calloc(1, n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t) {
  memcpy(dst, src, n);
  const uint32_t fake = g_fake_R;
  if (kind == hipMemcpyDeviceToHost && n == 8 && fake != 0xFFFFFFFFu) {  // the forward's one read-back: {R, prefilter flag}
    uint32_t w[2] = {fake, 0};
    memcpy(dst, w, 8);
  }
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

struct StubEvent {
  std::chrono::steady_clock::time_point t;
};
hipError_t hipEventCreate(hipEvent_t *e) {
  *e = reinterpret_cast<hipEvent_t>(new StubEvent());
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  reinterpret_cast<StubEvent *>(e)->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(reinterpret_cast<StubEvent *>(b)->t - reinterpret_cast<StubEvent *>(a)->t).count();
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete reinterpret_cast<StubEvent *>(e);
  return hipSuccess;
}

}  // extern "C"
