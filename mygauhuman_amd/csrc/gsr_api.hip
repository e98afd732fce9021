// gsr_api.hip -- the extern "C" surface of libgsr.so (declared in include/gsr.h) and the host-side
// orchestration of the rasterizer (replaces CudaRasterizer::Rasterizer::forward/backward/markVisible,
// CR/rasterizer_impl.cu:141-447).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "gsr_common.h"

namespace gsr {

static thread_local std::string g_error;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}

int check_hip(hipError_t e, const char *what, const char *file, int line) {
  if (e == hipSuccess) return GSR_OK;
  set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return GSR_EHIP;
}

// CR/rasterizer_impl.cu:35-50
static uint32_t higher_msb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb)
      msb += step;
    else
      msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

// ---- process-wide host state ------------------------------------------------------------------------------------------
// Everything mutable the API layer keeps between calls lives in ONE object that is created on first use and never destroyed:
// threads the library does not own (autograd's backward thread, a data-loader thread) may still be inside a call while the main
// thread runs static destructors at exit, and a destroyed mutex / map under them is undefined behaviour.  Every member is
// guarded by the mutex next to it; nothing here is touched below the API layer.
struct ProfRec {
  int stage;
  hipStream_t stream;
  hipEvent_t e0, e1;
};
struct ApiState {
  // options: process defaults + per-stream overrides, resolved once per call (gsr_common.h: Options)
  std::mutex opt_mutex;
  Options default_opt;
  std::unordered_map<hipStream_t, Options> stream_opt;
  // per-stage event timing: the autograd backward thread and the main thread both record stages
  std::mutex prof_mutex;
  std::atomic<unsigned> prof_mask{0};   // read without the mutex on the fast path (profiling off = two relaxed loads per stage)
  std::vector<ProfRec> prof_recs;                                 // recorded pairs awaiting collection
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;       // recycled events
  double prof_ms[PROF_NSTAGES] = {0, 0, 0, 0, 0, 0};
  long prof_n[PROF_NSTAGES] = {0, 0, 0, 0, 0, 0};
  // pinned 64-byte slots for the one device->host read of a blocking forward: as many as there are calls in flight at once
  // (one per calling thread at most), recycled -- not one per thread that ever called
  std::mutex pin_mutex;
  std::vector<uint32_t *> pin_free;
  // measurement: per-wave timestamps of the blend kernels (gsr_debug_wave_trace)
  std::atomic<unsigned long long *> trace{nullptr};
  std::atomic<size_t> trace_words{0};
};
static ApiState &S() {
  static ApiState *s = new ApiState();  // intentionally leaked, see above
  return *s;
}

Options options_for(hipStream_t stream) {
  ApiState &st = S();
  std::lock_guard<std::mutex> lock(st.opt_mutex);
  auto it = st.stream_opt.find(stream);
  return it == st.stream_opt.end() ? st.default_opt : it->second;
}

static int set_option(Options &o, const char *key, int v) {
  auto bad = [&](const char *what) {
    set_error("%s must be %s", key, what);
    return GSR_EINVAL;
  };
#ifdef GSR_BUILD_EXPERIMENTS
  constexpr bool EXPERIMENTS = true;
#else
  constexpr bool EXPERIMENTS = false;
#endif
  // knob values that select kernels which were built, measured and not adopted: compiled only with GSR_BUILD_EXPERIMENTS
  auto experiment = [&](bool is_experimental_value) {
    if (is_experimental_value && !EXPERIMENTS) {
      set_error("%s = %d selects an experiment kernel; this libgsr.so was built without them (python -m mygauhuman_amd.build --experiments)", key, v);
      return true;
    }
    return false;
  };
  if (!strcmp(key, "binning_mode")) {
    if (v != GSR_BINNING_GLOBAL_RADIX && v != GSR_BINNING_TILE_BUCKET) return bad("0 (global radix) or 1 (tile bucket)");
    o.binning_mode = v;
  } else if (!strcmp(key, "blend_fwd_waves")) {
    if (v != 1 && v != 2 && v != 4) return bad("1, 2 or 4");
    o.blend_fwd_waves = v;
  } else if (!strcmp(key, "blend_fwd_dma")) {
    if (v != 0 && v != 1) return bad("0 or 1");
    if (experiment(v == 1)) return GSR_EINVAL;
    o.blend_fwd_dma = v;
  } else if (!strcmp(key, "blend_bwd_waves")) {
    if (v != 1 && v != 2 && v != 4) return bad("1, 2 or 4");
    o.blend_bwd_waves = v;
  } else if (!strcmp(key, "bucket_cstride")) {
    if (v != 1 && v != 2 && v != 4 && v != 8 && v != 16) return bad("1, 2, 4, 8 or 16");
    o.bucket_cstride = v;
  } else if (!strcmp(key, "bucket_hist")) {
    if (v != 0 && v != 1) return bad("0 or 1");
    o.bucket_hist = v;
  } else if (!strcmp(key, "bucket_sort_merged")) {
    if (v != 0 && v != 1) return bad("0 or 1");
    o.bucket_sort_merged = v;
  } else if (!strcmp(key, "tile_order")) {
    if (v < 0 || v > 3) return bad("0 (natural order), 1 (longest lists first), 2 / 3 (2 x 2 / 4 x 2 tile blocks by summed length, a block per XCD)");
    o.tile_order = v;
  } else if (!strcmp(key, "tile_cull")) {
    if (v != 0 && v != 1) return bad("0 or 1");
    o.tile_cull = v;
  } else if (!strcmp(key, "blend_bwd_reduce")) {
    if (v < 0 || v > 4) return bad("0 (permlane / DPP folds), 3 (LDS folds: default); experiments: 1 (MFMA on folded rows), 2 (transposed MFMA contraction), 4 (LDS folds, the round-3 kernel)");
    if (experiment(v == 1 || v == 2 || v == 4)) return GSR_EINVAL;
    o.blend_bwd_reduce = v;
  } else if (!strcmp(key, "blend_layout")) {
    if (v != 0 && v != 1) return bad("0 (quadrant waves) or 1 (experiment: 4x4 blocks, four survivors per step)");
    if (experiment(v == 1)) return GSR_EINVAL;
    o.blend_layout = v;
  } else if (!strcmp(key, "blend_tail_cut")) {
    if (v < 0 || v > 16) return bad("0 .. 16 (sixteenths of the busy tiles at the end of the visiting order)");
    o.blend_tail_cut = v;
  } else if (!strcmp(key, "blend_segments")) {
    if (v < 0 || v > 64) return bad("0 (never) or the outlier threshold in quarters of the frame's mean list length (4 .. 64)");
    o.blend_segments = v;
  } else if (!strcmp(key, "blend_prio")) {
    if (v < 0 || v > 4) return bad("0, 1 (issue priority by list length); experiments: 2..4 (MEASUREMENT ONLY: render the longest lists alone)");
    if (experiment(v >= 2)) return GSR_EINVAL;
    o.blend_prio = v;
  } else if (!strcmp(key, "debug_no_atomics")) {
    if (v != 0 && v != 1) return bad("0 or 1");
    o.debug_no_atomics = v;
  } else if (!strcmp(key, "deterministic")) {
    if (v != 0 && v != 1) return bad("0 or 1");
    o.deterministic = v;
  } else {
    set_error("unknown tuning key %s", key);
    return GSR_EINVAL;
  }
  return GSR_OK;
}

// pinned host words for the one device->host read of a forward call: num_rendered (CR/rasterizer_impl.cu:283) and the
// "filtered although prefiltered" flag word next to it
static int readback_u32x2(const uint32_t *dev, uint32_t *out, hipStream_t stream) {
  ApiState &st = S();
  uint32_t *pinned = nullptr;
  {
    std::lock_guard<std::mutex> lock(st.pin_mutex);
    if (!st.pin_free.empty()) {
      pinned = st.pin_free.back();
      st.pin_free.pop_back();
    }
  }
  if (!pinned) GSR_HIP(hipHostMalloc(reinterpret_cast<void **>(&pinned), 64, hipHostMallocDefault));
  hipError_t e = hipMemcpyAsync(pinned, dev, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  out[0] = pinned[0];
  out[1] = pinned[1];
  {
    std::lock_guard<std::mutex> lock(st.pin_mutex);
    st.pin_free.push_back(pinned);
  }
  GSR_HIP(e);
  return GSR_OK;
}

// ---- per-stage event timing --------------------------------------------------------------------
void prof_begin(int stage, hipStream_t stream) {
  ApiState &st = S();
  if (!(st.prof_mask.load(std::memory_order_relaxed) & (1u << stage))) return;
  std::lock_guard<std::mutex> lock(st.prof_mutex);
  ProfRec r;
  r.stage = stage;
  r.stream = stream;
  if (!st.prof_pool.empty()) {
    r.e0 = st.prof_pool.back().first;
    r.e1 = st.prof_pool.back().second;
    st.prof_pool.pop_back();
  } else {
    if (hipEventCreate(&r.e0) != hipSuccess) return;
    if (hipEventCreate(&r.e1) != hipSuccess) {
      (void)hipEventDestroy(r.e0);
      return;
    }
  }
  (void)hipEventRecord(r.e0, stream);
  st.prof_recs.push_back(r);
}
void prof_end(int stage, hipStream_t stream) {
  ApiState &st = S();
  if (!(st.prof_mask.load(std::memory_order_relaxed) & (1u << stage))) return;
  std::lock_guard<std::mutex> lock(st.prof_mutex);
  for (size_t i = st.prof_recs.size(); i-- > 0;)  // the newest open record of this stage ON THIS STREAM (threads use their own streams)
    if (st.prof_recs[i].stage == stage && st.prof_recs[i].stream == stream) {
      (void)hipEventRecord(st.prof_recs[i].e1, stream);
      return;
    }
}
static void prof_collect(ApiState &st) {  // caller holds prof_mutex
  for (auto &r : st.prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      st.prof_ms[r.stage] += ms;
      st.prof_n[r.stage] += 1;
    }
    st.prof_pool.emplace_back(r.e0, r.e1);
  }
  st.prof_recs.clear();
}
static void prof_release_events(ApiState &st) {  // caller holds prof_mutex; profiling has been switched off
  for (auto &pr : st.prof_pool) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  st.prof_pool.clear();
}

}  // namespace gsr

using namespace gsr;

extern "C" {

int gsr_version(void) { return 100; }
int gsr_has_experiments(void) {
#ifdef GSR_BUILD_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}
const char *gsr_target_arch(void) { return "gfx950"; }
const char *gsr_last_error(void) { return g_error.c_str(); }

int gsr_set_tuning(const char *key, int value) {
  if (!key) {
    set_error("gsr_set_tuning: null key");
    return GSR_EINVAL;
  }
  ApiState &st = S();
  std::lock_guard<std::mutex> lock(st.opt_mutex);
  return set_option(st.default_opt, key, value);
}
int gsr_set_stream_tuning(gsr_stream_t stream_, const char *key, int value) {
  if (!key) {
    set_error("gsr_set_stream_tuning: null key");
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ApiState &st = S();
  std::lock_guard<std::mutex> lock(st.opt_mutex);
  auto it = st.stream_opt.find(stream);
  if (it != st.stream_opt.end()) return set_option(it->second, key, value);
  Options o = st.default_opt;  // starts as a copy of the defaults; a rejected value does not create an entry
  int rc = set_option(o, key, value);
  if (rc == GSR_OK) st.stream_opt.emplace(stream, o);
  return rc;
}
int gsr_clear_stream_tuning(gsr_stream_t stream_) {
  ApiState &st = S();
  std::lock_guard<std::mutex> lock(st.opt_mutex);
  st.stream_opt.erase(reinterpret_cast<hipStream_t>(stream_));
  return GSR_OK;
}
int gsr_set_binning_mode(int mode) { return gsr_set_tuning("binning_mode", mode); }
int gsr_get_binning_mode(void) {
  ApiState &st = S();
  std::lock_guard<std::mutex> lock(st.opt_mutex);
  return st.default_opt.binning_mode;
}

int gsr_debug_wave_trace(unsigned long long *device_buffer, size_t words) {
  ApiState &st = S();
  st.trace_words.store(device_buffer ? words : 0);
  st.trace.store(device_buffer);
  return GSR_OK;
}

int gsr_profile_enable(unsigned stage_mask) {
  ApiState &st = S();
  std::lock_guard<std::mutex> lock(st.prof_mutex);
  prof_collect(st);
  st.prof_mask = stage_mask & ((1u << PROF_NSTAGES) - 1);
  if (!stage_mask) prof_release_events(st);
  return GSR_OK;
}
int gsr_profile_reset(void) {
  ApiState &st = S();
  std::lock_guard<std::mutex> lock(st.prof_mutex);
  prof_collect(st);
  for (int i = 0; i < PROF_NSTAGES; i++) {
    st.prof_ms[i] = 0.0;
    st.prof_n[i] = 0;
  }
  return GSR_OK;
}
int gsr_profile_read(int stage, double *total_ms, long *launches) {
  if (stage < 0 || stage >= PROF_NSTAGES || !total_ms || !launches) {
    set_error("gsr_profile_read: bad arguments");
    return GSR_EINVAL;
  }
  ApiState &st = S();
  std::lock_guard<std::mutex> lock(st.prof_mutex);
  prof_collect(st);
  *total_ms = st.prof_ms[stage];
  *launches = st.prof_n[stage];
  return GSR_OK;
}

int gsr_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present,
                     gsr_stream_t stream_) {
  (void)projmatrix;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) {
    set_error("gsr_mark_visible: bad arguments");
    return GSR_EINVAL;
  }
  int rc = launch_mark_visible(P, means3D, viewmatrix, present, stream);
  if (rc != GSR_OK) return rc;
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

}  // extern "C"

namespace {

struct FwdIn {
  int P, D, M, width, height, prefiltered, debug;
  const float *background, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp, *viewmatrix,
      *projmatrix, *cam_pos;
  float scale_modifier, tan_fovx, tan_fovy;
  float *out_color, *out_depth, *out_alpha;
  int *radii;
  const float *extra;  // optional [P][n_extra] feature channels for the fused multi-feature blend
  int n_extra;
  float *out_extra;    // [n_extra][H][W]
  int sh_half;         // 1: shs are IEEE halves
};

int validate_forward(const FwdIn &in, const char *who) {
  if (in.P < 0 || in.width <= 0 || in.height <= 0) {
    set_error("%s: bad sizes", who);
    return GSR_EINVAL;
  }
  if (in.P == 0) return GSR_OK;
  if (!in.background || !in.means3D || !in.opacities || !in.viewmatrix || !in.projmatrix || !in.cam_pos || !in.out_color ||
      !in.out_depth || !in.out_alpha) {
    set_error("%s: null required pointer", who);
    return GSR_EINVAL;
  }
  if (!in.colors_precomp && !in.shs) {  // CR/rasterizer_impl.cu:244-247 (NUM_CHANNELS == 3 here, so SHs are acceptable)
    set_error("%s: provide SHs or precomputed colours", who);
    return GSR_EINVAL;
  }
  if (!in.cov3D_precomp && (!in.scales || !in.rotations)) {
    set_error("%s: provide scales+rotations or a precomputed 3D covariance", who);
    return GSR_EINVAL;
  }
  if (!in.colors_precomp && (in.D < 0 || in.D > 3 || in.M < (in.D + 1) * (in.D + 1))) {
    set_error("%s: SH degree %d needs M >= %d coefficients (got %d)", who, in.D, (in.D + 1) * (in.D + 1), in.M);
    return GSR_EINVAL;
  }
  return GSR_OK;
}

// preprocess + device-wide scan of tiles_touched
int forward_stage_a(const FwdIn &in, const GeomState &geom, int *radii, hipStream_t stream, bool skip_scan = false) {
  const int grid_x = (in.width + TILE - 1) / TILE, grid_y = (in.height + TILE - 1) / TILE;
  PreprocessArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.P = in.P;
  pa.D = in.D;
  pa.M = in.M;
  pa.means3D = in.means3D;
  pa.scales = in.scales;
  pa.rotations = in.rotations;
  pa.opacities = in.opacities;
  pa.shs = in.shs;
  pa.cov3D_precomp = in.cov3D_precomp;
  pa.colors_precomp = in.colors_precomp;
  pa.scale_modifier = in.scale_modifier;
  pa.view = in.viewmatrix;
  pa.proj = in.projmatrix;
  pa.campos = in.cam_pos;
  pa.W = in.width;
  pa.H = in.height;
  pa.grid_x = grid_x;
  pa.grid_y = grid_y;
  pa.tan_fovx = in.tan_fovx;
  pa.tan_fovy = in.tan_fovy;
  pa.focal_y = in.height / (2.0f * in.tan_fovy);  // CR/rasterizer_impl.cu:224-225
  pa.focal_x = in.width / (2.0f * in.tan_fovx);
  pa.radii = radii;
  pa.geom = geom;
  pa.prefiltered = in.prefiltered;
  pa.sh_half = in.sh_half;
  pa.zero_rows = (in.debug & GSR_FWD_ZERO_ROWS) ? 1 : 0;
  // flag word of the prefiltered contract (see preprocess_forward_kernel); untouched and unread unless prefiltered is set
  if (in.prefiltered) GSR_HIP(zero_async(geom.total + 1, sizeof(uint32_t), stream));
  prof_begin(PROF_PREPROCESS_FWD, stream);
  int rc = launch_preprocess_forward(pa, stream);
  prof_end(PROF_PREPROCESS_FWD, stream);
  if (rc != GSR_OK) return rc;
  GSR_LAUNCH_CHECK(stream, in.debug & 1);
  if (skip_scan) return GSR_OK;  // the histogram kernel of the binning does the second scan level itself
  prof_begin(PROF_SCAN, stream);
  rc = launch_scan_block_sums(geom, in.P, stream);
  prof_end(PROF_SCAN, stream);
  if (rc != GSR_OK) return rc;
  GSR_LAUNCH_CHECK(stream, in.debug & 1);
  return GSR_OK;
}

// binning + blend.  capacity = number of instances the binning buffer can hold; R_host < 0 means "unknown on the
// host" (asynchronous mode: tile-bucket back-end, kernels read R from geom.total and honour `capacity`).
int forward_stage_b(const FwdIn &in, const GeomState &geom, BinningState &bin, const ImageState &img, const int *radii,
                    long R_host, size_t capacity, uint32_t *dev_status, hipStream_t stream, bool scan_fused = false) {
  const Options opt = options_for(stream);
  const int grid_x = (in.width + TILE - 1) / TILE, grid_y = (in.height + TILE - 1) / TILE;
  const size_t tiles = (size_t)grid_x * grid_y;
  int rc;
  prof_begin(PROF_BINNING, stream);
  if (R_host < 0 || opt.binning_mode == GSR_BINNING_TILE_BUCKET) {
    // list segments need the forward variant that writes the checkpoints (blend_forward_kernel<1, .>)
    const int segments = (opt.tile_order == 1 && opt.blend_layout == 0 && !opt.blend_fwd_dma && (in.n_extra != 0 || opt.blend_fwd_waves == 4))
                             ? (opt.blend_segments | (opt.blend_tail_cut << 8)) : 0;
    rc = bucket_binning(geom, radii, in.P, grid_x, grid_y, capacity, R_host < 0, bin, img.ranges, img.order, img.ckpt_base, segments,
                        dev_status, in.prefiltered != 0, scan_fused, opt, stream, in.debug & 1);
    if (rc != GSR_OK) return rc;
  } else {
    const size_t R = (size_t)R_host;
    const int end_bit = 32 + (int)higher_msb((uint32_t)tiles);  // CR/rasterizer_impl.cu:302,310
    const int passes = radix_passes(end_bit);
    // duplicate into whichever buffer makes the last pass land in (keys_s, vals_s)
    uint64_t *dup_k = (passes % 2) ? bin.keys_a : bin.keys_s;
    uint32_t *dup_v = (passes % 2) ? bin.vals_a : bin.vals_s;
    uint64_t *oth_k = (passes % 2) ? bin.keys_s : bin.keys_a;
    uint32_t *oth_v = (passes % 2) ? bin.vals_s : bin.vals_a;
    rc = launch_duplicate(geom, radii, in.P, grid_x, grid_y, dup_k, dup_v, stream);
    if (rc != GSR_OK) return rc;
    GSR_LAUNCH_CHECK(stream, in.debug & 1);
    rc = radix_sort_u64(R, dup_k, dup_v, oth_k, oth_v, dup_k, dup_v, end_bit, bin.hist, stream, in.debug & 1);
    if (rc != GSR_OK) return rc;
    rc = launch_tile_ranges(R, bin.keys_s, img.ranges, tiles, stream);
    if (rc != GSR_OK) return rc;
    GSR_HIP(zero_async(img.order, sizeof(uint32_t), stream));  // mode word 0: natural tile order
    GSR_LAUNCH_CHECK(stream, in.debug & 1);
  }
  prof_end(PROF_BINNING, stream);

  BlendFwdArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.ranges = img.ranges;
  fa.order = img.order;
  fa.point_list = bin.vals_s;
  fa.recs = geom.recs;
  fa.W = in.width;
  fa.H = in.height;
  fa.grid_x = grid_x;
  fa.grid_y = grid_y;
  fa.bg = in.background;
  fa.out_color = in.out_color;
  fa.out_depth = in.out_depth;
  fa.out_alpha = in.out_alpha;
  fa.final_T = img.final_T;
  fa.n_contrib = img.n_contrib;
  fa.extra = in.extra;
  fa.CE = in.n_extra;
  fa.list_prio = opt.blend_prio;
  fa.ckpt_base = img.ckpt_base;
  fa.ckpt = img.ckpt;
  {  // (first half of the trace buffer: forward waves, 4 words each; second half: backward waves)
    ApiState &st = S();
    const size_t need = (size_t)tile_slots_max(grid_x, grid_y) * 4u * 4u;
    fa.trace = st.trace.load();
    if (fa.trace && st.trace_words.load() / 2 < need) {
      set_error("gsr_debug_wave_trace: the registered buffer holds %zu words, this image needs %zu (2 x 16 x visiting slots)",
                st.trace_words.load(), 2 * need);
      return GSR_EINVAL;
    }
  }
  fa.out_extra = in.out_extra;
  prof_begin(PROF_BLEND_FWD, stream);
  rc = launch_blend_forward(fa, opt, stream);
  prof_end(PROF_BLEND_FWD, stream);
  if (rc != GSR_OK) return rc;
  GSR_LAUNCH_CHECK(stream, in.debug & 1);
  return GSR_OK;
}

}  // namespace

extern "C" {

int gsr_rasterize_forward_ex(gsr_alloc_fn geometry_alloc, void *geometry_user, gsr_alloc_fn binning_alloc, void *binning_user,
                             gsr_alloc_fn image_alloc, void *image_user, int P, int D, int M, const float *background, int width,
                             int height, const float *means3D, const float *shs, const float *colors_precomp,
                             const float *opacities, const float *scales, float scale_modifier, const float *rotations,
                             const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *cam_pos,
                             float tan_fovx, float tan_fovy, int prefiltered, float *out_color, float *out_depth,
                             float *out_alpha, int *radii, int debug, int *host_num_rendered, const float *extra_features,
                             int n_extra, float *out_extra, int sh_dtype, gsr_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (host_num_rendered) *host_num_rendered = 0;
  if (!geometry_alloc || !binning_alloc || !image_alloc || !host_num_rendered) {
    set_error("gsr_rasterize_forward: missing allocation callbacks");
    return GSR_EINVAL;
  }
  const FwdIn in = {P, D, M, width, height, prefiltered, debug, background, means3D, shs, colors_precomp, opacities, scales,
                    rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, scale_modifier, tan_fovx, tan_fovy, out_color,
                    out_depth, out_alpha, radii, extra_features, n_extra, out_extra, sh_dtype == GSR_SH_F16 ? 1 : 0};
  int rc = validate_forward(in, "gsr_rasterize_forward");
  if (rc != GSR_OK) return rc;
  if (P == 0) return GSR_OK;  // DGR/rasterize_points.cu:84
  const int grid_x = (width + TILE - 1) / TILE, grid_y = (height + TILE - 1) / TILE;
  const size_t tiles = (size_t)grid_x * grid_y, npix = (size_t)width * height;

  char *gchunk = geometry_alloc(geometry_user, geom_bytes((size_t)P));
  if (!gchunk) {
    set_error("geometry allocation callback returned null");
    return GSR_ENOMEM;
  }
  GeomState geom = geom_from_chunk(gchunk, (size_t)P);
  if (!radii) radii = geom.internal_radii;  // CR/rasterizer_impl.cu:231-234
  char *ichunk = image_alloc(image_user, image_bytes(npix, tiles));
  if (!ichunk) {
    set_error("image allocation callback returned null");
    return GSR_ENOMEM;
  }
  ImageState img = image_from_chunk(ichunk, npix, tiles);

  rc = forward_stage_a(in, geom, radii, stream);
  if (rc != GSR_OK) return rc;
  uint32_t rb[2] = {0, 0};  // the one device -> host read of the call (CR/rasterizer_impl.cu:283)
  rc = readback_u32x2(geom.total, rb, stream);
  if (rc != GSR_OK) return rc;
  const uint32_t R = rb[0];
  if (prefiltered && rb[1]) {
    set_error("Point is filtered although prefiltered is set. This shouldn't happen!");  // CR/auxiliary.h:158
    return GSR_EINVAL;
  }
  *host_num_rendered = (int)R;
  char *bchunk = binning_alloc(binning_user, binning_bytes((size_t)R, tiles));
  if (!bchunk) {
    set_error("binning allocation callback returned null");
    return GSR_ENOMEM;
  }
  BinningState bin = binning_from_chunk(bchunk, (size_t)R, tiles);
  return forward_stage_b(in, geom, bin, img, radii, (long)R, (size_t)R, nullptr, stream);
}

size_t gsr_geometry_bytes(int P) { return geom_bytes(P > 0 ? (size_t)P : 1); }
size_t gsr_image_bytes(int width, int height) {
  const size_t tiles = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  return image_bytes((size_t)width * height, tiles);
}
size_t gsr_binning_bytes(size_t capacity, int width, int height) {
  const size_t tiles = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  return binning_bytes(capacity, tiles);
}

int gsr_rasterize_forward(gsr_alloc_fn geometry_alloc, void *geometry_user, gsr_alloc_fn binning_alloc, void *binning_user,
                          gsr_alloc_fn image_alloc, void *image_user, int P, int D, int M, const float *background, int width,
                          int height, const float *means3D, const float *shs, const float *colors_precomp,
                          const float *opacities, const float *scales, float scale_modifier, const float *rotations,
                          const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *cam_pos,
                          float tan_fovx, float tan_fovy, int prefiltered, float *out_color, float *out_depth, float *out_alpha,
                          int *radii, int debug, int *host_num_rendered, gsr_stream_t stream) {
  return gsr_rasterize_forward_ex(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D, M,
                                  background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                                  rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
                                  out_color, out_depth, out_alpha, radii, debug, host_num_rendered, nullptr, 0, nullptr, GSR_SH_F32,
                                  stream);
}

int gsr_rasterize_forward_async_ex(char *geom_buffer, char *binning_buffer, size_t binning_capacity, char *image_buffer, int P,
                                   int D, int M, const float *background, int width, int height, const float *means3D,
                                   const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                                   float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                   const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                                   float tan_fovy, int prefiltered, float *out_color, float *out_depth, float *out_alpha,
                                   int *radii, int debug, uint32_t *dev_status, const float *extra_features, int n_extra,
                                   float *out_extra, int sh_dtype, gsr_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const FwdIn in = {P, D, M, width, height, prefiltered, debug, background, means3D, shs, colors_precomp, opacities, scales,
                    rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, scale_modifier, tan_fovx, tan_fovy, out_color,
                    out_depth, out_alpha, radii, extra_features, n_extra, out_extra, sh_dtype == GSR_SH_F16 ? 1 : 0};
  int rc = validate_forward(in, "gsr_rasterize_forward_async");
  if (rc != GSR_OK) return rc;
  if (!geom_buffer || !binning_buffer || !image_buffer || !dev_status || P <= 0) {
    set_error("gsr_rasterize_forward_async: buffers, dev_status and P > 0 are required");
    return GSR_EINVAL;
  }
  const int grid_x = (width + TILE - 1) / TILE, grid_y = (height + TILE - 1) / TILE;
  const size_t tiles = (size_t)grid_x * grid_y, npix = (size_t)width * height;
  GeomState geom = geom_from_chunk(geom_buffer, (size_t)P);
  if (!radii) radii = geom.internal_radii;
  ImageState img = image_from_chunk(image_buffer, npix, tiles);
  BinningState bin = binning_from_chunk(binning_buffer, binning_capacity, tiles);
  const bool scan_fused = bucket_uses_hist(options_for(stream), P, tiles, binning_capacity);
  rc = forward_stage_a(in, geom, radii, stream, scan_fused);
  if (rc != GSR_OK) return rc;
  return forward_stage_b(in, geom, bin, img, radii, -1, binning_capacity, dev_status, stream, scan_fused);
}

int gsr_rasterize_forward_async(char *geom_buffer, char *binning_buffer, size_t binning_capacity, char *image_buffer, int P,
                                int D, int M, const float *background, int width, int height, const float *means3D,
                                const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                                float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                                float tan_fovy, int prefiltered, float *out_color, float *out_depth, float *out_alpha,
                                int *radii, int debug, uint32_t *dev_status, gsr_stream_t stream) {
  return gsr_rasterize_forward_async_ex(geom_buffer, binning_buffer, binning_capacity, image_buffer, P, D, M, background, width,
                                        height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                        cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered,
                                        out_color, out_depth, out_alpha, radii, debug, dev_status, nullptr, 0, nullptr, GSR_SH_F32,
                                        stream);
}

namespace {
struct FusedLoss {  // gsr_rasterize_backward_alpha_mask_loss: the image gradients are formed in the blend-backward kernel
  const float *color, *gt, *mask;
  float lambda_alpha;
};
}  // namespace

static int rasterize_backward_impl(const FusedLoss *fused_loss, const gsr_phase1_loss *p1, int P, int D, int M, int R, const float *background, int width, int height, const float *means3D,
                              const float *shs, const float *colors_precomp, const float *alphas, const float *scales,
                              float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                              const float *projmatrix, const float *campos, float tan_fovx, float tan_fovy, const int *radii,
                              char *geom_buffer, char *binning_buffer, char *image_buffer, const float *dL_dpix,
                              const float *dL_ddepths, const float *dL_dalphas, float *dL_dmean2D, float *dL_dconic,
                              float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                              float *dL_dscale, float *dL_drot, int debug, const float *extra_features, int n_extra,
                              const float *const *dL_dout_extra, float *dL_dextra, int sh_dtype, gsr_stream_t stream_) {
  // alphas: unused by the reference kernel (CR/backward.cu:410); read by the fused alpha-mask loss only
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int flags = debug;  // bit 0: debug mode, GSR_BWD_ROWS_ZEROED
  debug &= 1;
  if (P < 0 || R < 0 || width <= 0 || height <= 0) {
    set_error("gsr_rasterize_backward: bad sizes");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  if (P >= (1 << 25)) {  // the blend backward keeps a Gaussian's gradient row as a 32-bit byte offset (rows of up to 128 bytes)
    set_error("gsr_rasterize_backward: at most 2^25 - 1 Gaussians per call");
    return GSR_EINVAL;
  }
  if (!geom_buffer || !binning_buffer || !image_buffer || !background || !means3D || !viewmatrix || !projmatrix || !campos ||
      (!p1 && (!dL_dpix || !dL_ddepths || !dL_dalphas)) || !dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor || !dL_dmean3D ||
      !dL_dcov3D) {
    set_error("gsr_rasterize_backward: null required pointer");
    return GSR_EINVAL;
  }
  if (p1 && (n_extra != CE_MAX || !p1->gt_image || !p1->gt_normal || !p1->alpha_target || !p1->bound || !p1->color || !p1->alpha ||
             !p1->extra_images || !p1->stats || p1->normal_triple < 0 || p1->normal_triple > 5 || p1->axis_triple < 0 ||
             p1->axis_triple > 5 || p1->axis_triple == p1->normal_triple)) {
    set_error("gsr_rasterize_backward_phase1_loss: the fused multi-feature pass (n_extra = %d), the forward's images, the targets, "
              "the bound mask, stats and two different triple indices in 0..5 are required", CE_MAX);
    return GSR_EINVAL;
  }
  if ((shs && !dL_dsh) || (scales && (!rotations || !dL_dscale || !dL_drot))) {
    set_error("gsr_rasterize_backward: missing gradient output for SH / scale / rotation inputs");
    return GSR_EINVAL;
  }
  const int grid_x = (width + TILE - 1) / TILE, grid_y = (height + TILE - 1) / TILE;
  const size_t tiles = (size_t)grid_x * grid_y, npix = (size_t)width * height;
  GeomState geom = geom_from_chunk(geom_buffer, (size_t)P);
  BinningState bin = binning_from_chunk(binning_buffer, (size_t)R);
  ImageState img = image_from_chunk(image_buffer, npix, tiles);
  if (!radii) radii = geom.internal_radii;

  if (n_extra != 0 && (n_extra != CE_MAX || !extra_features || !dL_dout_extra || !dL_dextra)) {
    set_error("gsr_rasterize_backward_ex: extra feature channels need n_extra == %d and all three arrays", CE_MAX);
    return GSR_EINVAL;
  }
  const int grow = n_extra ? GROWX : GROW;
  const Options opt = options_for(stream);
  float *det_rows = nullptr;
  const size_t det_bytes = (size_t)(R > 0 ? R : 1) * 4 * GROW * sizeof(float);
  if (opt.deterministic) {  // one 64-byte slot per (instance, quadrant), zeroed: culled instances keep zeros
    // (a test mode: plain hipMalloc / hipFree around the call, no stream-ordered pool involved)
    GSR_HIP(hipMalloc(reinterpret_cast<void **>(&det_rows), det_bytes));
    GSR_HIP(zero_async(det_rows, det_bytes, stream));
  } else if (!(flags & GSR_BWD_ROWS_ZEROED)) {
    GSR_HIP(zero_async(geom.grad_rows, (size_t)P * grow * sizeof(float), stream));
  }
  BlendBwdArgs ba;
  memset(&ba, 0, sizeof(ba));
  ba.ranges = img.ranges;
  ba.order = img.order;
  ba.point_list = bin.vals_s;
  ba.recs = geom.recs;
  ba.W = width;
  ba.H = height;
  ba.grid_x = grid_x;
  ba.grid_y = grid_y;
  ba.bg = background;
  ba.final_T = img.final_T;
  ba.n_contrib = img.n_contrib;
  ba.dL_dpix = dL_dpix;
  ba.dL_ddepth = dL_ddepths;
  ba.dL_dalpha = dL_dalphas;
  if (fused_loss) {
    ba.loss_color = fused_loss->color, ba.loss_alpha = alphas, ba.loss_gt = fused_loss->gt, ba.loss_mask = fused_loss->mask;
    ba.loss_sc = 1.0f / (3.0f * (float)npix);
    ba.loss_sa = 2.0f * fused_loss->lambda_alpha / (float)npix;
  }
  ba.grad_rows = geom.grad_rows;
  ba.extra = extra_features;
  ba.CE = n_extra;
  ba.extra_mask = 0u;
  for (int t = 0; t < CE_MAX / 3; t++) {
    ba.dL_dextra_tri[t] = (n_extra && dL_dout_extra) ? dL_dout_extra[t] : nullptr;
    if (ba.dL_dextra_tri[t]) ba.extra_mask |= 1u << t;
  }
  if (p1) {
    ba.p1 = *p1;
    ba.extra_mask |= (1u << p1->normal_triple) | (1u << p1->axis_triple);  // live whether or not a further gradient arrives on them
    if (opt.deterministic || opt.blend_bwd_reduce != 3) {
      set_error("gsr_rasterize_backward_phase1_loss: built for the default reduction of the fused pass (blend_bwd_reduce = 3, not deterministic)");
      return GSR_EINVAL;
    }
  }
  ba.det_rows = det_rows;
  ba.debug_skip_atomics = opt.debug_no_atomics;
  ba.list_prio = opt.blend_prio;
  ba.ckpt_base = img.ckpt_base;
  ba.ckpt = img.ckpt;
  {
    ApiState &st = S();
    const size_t need = (size_t)tile_slots_max(ba.grid_x, ba.grid_y) * 4u * 4u;
    ba.trace = st.trace.load() ? st.trace.load() + st.trace_words.load() / 2 : nullptr;
    if (ba.trace && st.trace_words.load() / 2 < need) {
      set_error("gsr_debug_wave_trace: the registered buffer holds %zu words, this image needs %zu (2 x 16 x visiting slots)",
                st.trace_words.load(), 2 * need);
      return GSR_EINVAL;
    }
  }
  ba.radii = radii;
  ba.point_offsets = geom.point_offsets;
  ba.tiles_touched = geom.tiles_touched;
  prof_begin(PROF_BLEND_BWD, stream);
  int rc = launch_blend_backward(ba, opt, stream);
  if (rc == GSR_OK && det_rows) rc = launch_reduce_det_rows(P, geom.point_offsets, geom.tiles_touched, det_rows, (size_t)(R > 0 ? R : 1) * 4, geom.grad_rows, stream);
  prof_end(PROF_BLEND_BWD, stream);
  if (rc != GSR_OK) {
    if (det_rows) {
      (void)hipStreamSynchronize(stream);
      (void)hipFree(det_rows);
    }
    return rc;
  }
  GSR_LAUNCH_CHECK(stream, debug);

  PreprocessBwdArgs pb;
  memset(&pb, 0, sizeof(pb));
  pb.P = P;
  pb.D = D;
  pb.M = M;
  pb.means3D = means3D;
  pb.shs = shs;
  pb.scales = scales;
  pb.rotations = rotations;
  pb.cov3D = cov3D_precomp ? cov3D_precomp : geom.cov3D;  // CR/rasterizer_impl.cu:424
  pb.radii = radii;
  pb.clamped = geom.clamped;
  pb.scale_modifier = scale_modifier;
  pb.view = viewmatrix;
  pb.proj = projmatrix;
  pb.campos = campos;
  pb.W = width;
  pb.H = height;
  pb.tan_fovx = tan_fovx;
  pb.tan_fovy = tan_fovy;
  pb.focal_y = height / (2.0f * tan_fovy);
  pb.focal_x = width / (2.0f * tan_fovx);
  pb.grad_rows = geom.grad_rows;
  pb.grow = grow;
  pb.clear_rows = (flags & GSR_BWD_ROWS_ZEROED) ? 1 : 0;
  pb.CE = n_extra;
  pb.sh_half = sh_dtype == GSR_SH_F16 ? 1 : 0;
  pb.dL_dextra = dL_dextra;
  pb.recs = geom.recs;
  pb.dL_dmean2D = dL_dmean2D;
  pb.dL_dconic = dL_dconic;
  pb.dL_dopacity = dL_dopacity;
  pb.dL_dcolor = dL_dcolor;
  pb.dL_dmean3D = dL_dmean3D;
  pb.dL_dcov3D = dL_dcov3D;
  pb.dL_dsh = dL_dsh;
  pb.dL_dscale = dL_dscale;
  pb.dL_drot = dL_drot;
  prof_begin(PROF_PREPROCESS_BWD, stream);
  rc = launch_preprocess_backward(pb, stream);
  prof_end(PROF_PREPROCESS_BWD, stream);
  if (det_rows) {  // deterministic (test) mode: the slots die with the call
    (void)hipStreamSynchronize(stream);
    (void)hipFree(det_rows);
  }
  if (rc != GSR_OK) return rc;
  GSR_LAUNCH_CHECK(stream, debug);
  return GSR_OK;
}

int gsr_rasterize_backward_ex(int P, int D, int M, int R, const float *background, int width, int height, const float *means3D,
                              const float *shs, const float *colors_precomp, const float *alphas, const float *scales,
                              float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                              const float *projmatrix, const float *campos, float tan_fovx, float tan_fovy, const int *radii,
                              char *geom_buffer, char *binning_buffer, char *image_buffer, const float *dL_dpix,
                              const float *dL_ddepths, const float *dL_dalphas, float *dL_dmean2D, float *dL_dconic,
                              float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                              float *dL_dscale, float *dL_drot, int debug, const float *extra_features, int n_extra,
                              const float *const *dL_dout_extra, float *dL_dextra, int sh_dtype, gsr_stream_t stream) {
  return rasterize_backward_impl(nullptr, nullptr, P, D, M, R, background, width, height, means3D, shs, colors_precomp, alphas, scales,
                                 scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
                                 geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_ddepths, dL_dalphas, dL_dmean2D, dL_dconic,
                                 dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, extra_features,
                                 n_extra, dL_dout_extra, dL_dextra, sh_dtype, stream);
}

int gsr_rasterize_backward_phase1_loss(int P, int D, int M, int R, const float *background, int width, int height, const float *means3D,
                                       const float *shs, const float *colors_precomp, const float *alphas, const float *scales,
                                       float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                       const float *viewmatrix, const float *projmatrix, const float *campos, float tan_fovx,
                                       float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer,
                                       char *image_buffer, const float *dL_dpix, const float *dL_ddepths, const float *dL_dalphas,
                                       float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                                       float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot,
                                       int debug, const float *extra_features, int n_extra, const float *const *dL_dout_extra,
                                       float *dL_dextra, int sh_dtype, const gsr_phase1_loss *loss, gsr_stream_t stream) {
  if (!loss) {
    set_error("gsr_rasterize_backward_phase1_loss: loss descriptor required");
    return GSR_EINVAL;
  }
  return rasterize_backward_impl(nullptr, loss, P, D, M, R, background, width, height, means3D, shs, colors_precomp, alphas, scales,
                                 scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
                                 geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_ddepths, dL_dalphas, dL_dmean2D, dL_dconic,
                                 dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, extra_features,
                                 n_extra, dL_dout_extra, dL_dextra, sh_dtype, stream);
}

int gsr_rasterize_backward_alpha_mask_loss(int P, int D, int M, int R, const float *background, int width, int height,
                                           const float *means3D, const float *shs, const float *colors_precomp,
                                           const float *out_alpha, const float *scales, float scale_modifier, const float *rotations,
                                           const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                                           const float *campos, float tan_fovx, float tan_fovy, const int *radii, char *geom_buffer,
                                           char *binning_buffer, char *image_buffer, const float *out_color, const float *gt,
                                           const float *mask, float lambda_alpha, float *dL_dmean2D, float *dL_dconic,
                                           float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                                           float *dL_dscale, float *dL_drot, int debug, int sh_dtype, gsr_stream_t stream) {
  if (!out_color || !out_alpha || !gt || !mask) {
    set_error("gsr_rasterize_backward_alpha_mask_loss: the rendered colour and alpha images, gt and mask are required");
    return GSR_EINVAL;
  }
  const FusedLoss fl = {out_color, gt, mask, lambda_alpha};
  // (the three gradient-image arguments only have to be non-null: the kernel does not read them in this mode)
  return rasterize_backward_impl(&fl, nullptr, P, D, M, R, background, width, height, means3D, shs, colors_precomp, out_alpha, scales,
                                 scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
                                 geom_buffer, binning_buffer, image_buffer, out_color, out_alpha, out_alpha, dL_dmean2D, dL_dconic,
                                 dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, nullptr, 0, nullptr,
                                 nullptr, sh_dtype, stream);
}

int gsr_rasterize_backward(int P, int D, int M, int R, const float *background, int width, int height, const float *means3D,
                           const float *shs, const float *colors_precomp, const float *alphas, const float *scales,
                           float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                           const float *projmatrix, const float *campos, float tan_fovx, float tan_fovy, const int *radii,
                           char *geom_buffer, char *binning_buffer, char *image_buffer, const float *dL_dpix,
                           const float *dL_ddepths, const float *dL_dalphas, float *dL_dmean2D, float *dL_dconic,
                           float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                           float *dL_dscale, float *dL_drot, int debug, gsr_stream_t stream) {
  return gsr_rasterize_backward_ex(P, D, M, R, background, width, height, means3D, shs, colors_precomp, alphas, scales,
                                   scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                                   radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_ddepths, dL_dalphas, dL_dmean2D,
                                   dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug,
                                   nullptr, 0, nullptr, nullptr, GSR_SH_F32, stream);
}

int gsr_query_state(int what, int P, int R, int width, int height, const char *geom_buffer, const char *binning_buffer,
                    const char *image_buffer, void *dst, gsr_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int grid_x = (width + TILE - 1) / TILE, grid_y = (height + TILE - 1) / TILE;
  const size_t tiles = (size_t)grid_x * grid_y, npix = (size_t)width * height;
  if (!dst) return GSR_EINVAL;
  GeomState geom = geom_from_chunk(const_cast<char *>(geom_buffer), (size_t)P);
  BinningState bin = binning_from_chunk(const_cast<char *>(binning_buffer), (size_t)R);
  ImageState img = image_from_chunk(const_cast<char *>(image_buffer), npix, tiles);
  const void *src = nullptr;
  size_t bytes = 0;
  switch (what) {
    case GSR_Q_DEPTHS:
    case GSR_Q_MEANS2D:
    case GSR_Q_CONIC_OPACITY:
    case GSR_Q_RGB:
    case GSR_Q_CLAMPED: {
      if (!geom_buffer) return GSR_EINVAL;
      int rc = launch_query_recs(what, P, geom, dst, stream);
      if (rc != GSR_OK) return rc;
      GSR_LAUNCH_CHECK(stream, 0);
      return GSR_OK;
    }
    case GSR_Q_COV3D: src = geom.cov3D; bytes = (size_t)P * 6 * 4; break;
    case GSR_Q_TILES_TOUCHED: src = geom.tiles_touched; bytes = (size_t)P * 4; break;
    case GSR_Q_POINT_OFFSETS: src = geom.point_offsets; bytes = (size_t)P * 4; break;
    case GSR_Q_POINT_LIST: src = bin.vals_s; bytes = (size_t)R * 4; break;
    case GSR_Q_KEYS_SORTED: src = bin.keys_s; bytes = (size_t)R * 8; break;
    case GSR_Q_RANGES: src = img.ranges; bytes = tiles * 8; break;
    case GSR_Q_FINAL_T: src = img.final_T; bytes = npix * 4; break;
    case GSR_Q_N_CONTRIB: src = img.n_contrib; bytes = npix * 4; break;
    case GSR_Q_ORDER: src = img.order; bytes = order_words(tiles) * 4; break;
    default: set_error("gsr_query_state: unknown selector %d", what); return GSR_EINVAL;
  }
  if (bytes) GSR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
  return GSR_OK;
}

size_t gsr_sort_workspace_bytes(size_t n) {
  size_t m = n ? n : 1;
  return align_up(m * 8, 256) + align_up(m * 4, 256) + align_up(sort_hist_words(m) * 4, 256) + 1024;
}

static int sort_ws(size_t n, size_t key_bytes, char *ws, size_t ws_bytes, void **tk, uint32_t **tv, uint32_t **hist) {
  if (!ws || ws_bytes < gsr_sort_workspace_bytes(n)) {
    set_error("sort workspace too small (%zu < %zu)", ws_bytes, gsr_sort_workspace_bytes(n));
    return GSR_ENOMEM;
  }
  size_t m = n ? n : 1;
  char *p = reinterpret_cast<char *>(align_up(reinterpret_cast<size_t>(ws), 256));
  *tk = p;
  p += align_up(m * key_bytes, 256);
  *tv = reinterpret_cast<uint32_t *>(p);
  p += align_up(m * 4, 256);
  *hist = reinterpret_cast<uint32_t *>(p);
  return GSR_OK;
}

int gsr_sort_pairs_u64(size_t n, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                       int end_bit, char *workspace, size_t workspace_bytes, gsr_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (end_bit <= 0 || end_bit > 64 || (n && (!keys_in || !keys_out || !vals_in || !vals_out))) {
    set_error("gsr_sort_pairs_u64: bad arguments");
    return GSR_EINVAL;
  }
  void *tk;
  uint32_t *tv, *hist;
  int rc = sort_ws(n, 8, workspace, workspace_bytes, &tk, &tv, &hist);
  if (rc != GSR_OK) return rc;
  const int passes = radix_passes(end_bit);
  uint64_t *t = reinterpret_cast<uint64_t *>(tk);
  if (passes % 2)  // result lands in x
    return radix_sort_u64(n, keys_in, vals_in, keys_out, vals_out, t, tv, end_bit, hist, stream, 0);
  return radix_sort_u64(n, keys_in, vals_in, t, tv, keys_out, vals_out, end_bit, hist, stream, 0);
}

int gsr_sort_pairs_u32(size_t n, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                       int end_bit, char *workspace, size_t workspace_bytes, gsr_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (end_bit <= 0 || end_bit > 32 || (n && (!keys_in || !keys_out || !vals_in || !vals_out))) {
    set_error("gsr_sort_pairs_u32: bad arguments");
    return GSR_EINVAL;
  }
  void *tk;
  uint32_t *tv, *hist;
  int rc = sort_ws(n, 8, workspace, workspace_bytes, &tk, &tv, &hist);
  if (rc != GSR_OK) return rc;
  const int passes = radix_passes(end_bit);
  uint32_t *t = reinterpret_cast<uint32_t *>(tk);
  if (passes % 2) return radix_sort_u32(n, keys_in, vals_in, keys_out, vals_out, t, tv, end_bit, hist, stream, 0);
  return radix_sort_u32(n, keys_in, vals_in, t, tv, keys_out, vals_out, end_bit, hist, stream, 0);
}

}  // extern "C"
