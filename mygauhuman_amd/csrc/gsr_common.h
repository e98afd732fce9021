// gsr_common.h -- shared declarations of libgsr.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/gsr.h"

namespace gsr {

constexpr int TILE = 16;        // CR/config.h:16-17
constexpr int WAVE = 64;
constexpr int PRE_BLOCK = 256;  // Gaussians per preprocess / duplicate block (they share the block-local scan)

// One 48-byte record per Gaussian: everything the blend kernels gather for a (Gaussian, tile) instance,
// contiguous so that an instance costs one 48-B gather instead of four (CR/forward.cu:322-325,360-362 read
// point_list -> means2D, conic_opacity, features, depths from four arrays).
struct alignas(16) SplatRec {
  float x, y, conic_a, conic_b;        // pixel centre, conic.x, conic.y
  float conic_c, opacity, depth, r;    // conic.z (CUDA float3 .z), opacity, view depth, red
  float g, b, hx, hy;                  // half-extents (px) of the box outside which alpha < 1/255 for sure (cull only)
};
static_assert(sizeof(SplatRec) == 48, "SplatRec layout");

// Packed per-Gaussian gradient row written by the blend backward (one 64-B line per Gaussian so that a
// Gaussian's nine float atomics are one memory-side request): see blend_bwd.hip.
constexpr int GROW = 16;   // floats per row (3 colour channels): [0]=dmean2D.x [1]=dmean2D.y [2..4]=dconic.x,.y,.w [5]=dopacity [6..8]=dcolor

constexpr int GROWX = 32;  // floats per row when extra feature channels are blended too (9 + up to 18 colour gradients)
constexpr int CE_MAX = 18; // extra feature channels supported by the fused multi-feature blend (6 RGB triples)

struct GeomState {  // per Gaussian
  SplatRec *recs;
  float *cov3D;             // [P][6]
  uint8_t *clamped;         // [P] bit c = colour channel c was clamped at 0
  uint32_t *tiles_touched;  // [P]
  uint32_t *point_offsets;  // [P] inclusive scan
  int *internal_radii;      // [P]
  uint32_t *block_incl;     // [P] block-local inclusive scan of tiles_touched
  uint32_t *block_sums;     // [nblk]
  uint32_t *block_prefix;   // [nblk] exclusive prefix of block_sums
  uint32_t *total;          // [4] R; [1] = "a point was filtered although prefiltered is set" (valid only in prefiltered calls)
  float *grad_rows;         // [P][GROW or GROWX] backward accumulation rows (zeroed by the backward)
};
struct BinningState {  // per instance
  uint64_t *keys_a;  // "unsorted" role
  uint32_t *vals_a;
  uint64_t *keys_s;  // sorted keys (final)
  uint32_t *vals_s;  // point_list (final)
  uint32_t *hist;    // radix histograms
  uint32_t *tile_counts;  // [tiles] tile-bucket back-end (null if the buffer was carved without a tile count)
  uint32_t *tile_cursor;  // [tiles]
};
struct ImageState {
  float *final_T;       // [H*W]
  uint32_t *n_contrib;  // [H*W]
  uint2 *ranges;        // [tiles]
  uint32_t *order;      // [order_words(tiles)] the order in which the blend kernels visit the tiles: [0] = mode word written by
                        // every binning path (0 natural order; else entries [ORDER_HDR ..] = one per visiting slot, see tile_slots()
                        // and order_entry_*), [1] = slots in use (mode 1); an entry ORDER_NO_TILE = nothing to render in that slot
  uint32_t *ckpt_base;  // [tiles] first checkpoint record of a tile whose list is walked in segments (see "list segments" below)
  float *ckpt;          // [ckpt_records(tiles)][CKPT_PLANES][256] per-pixel blend state at the segment boundaries
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// Bump allocation at 256-byte alignment (CR/rasterizer_impl.h:21-27 `obtain`).  Works on an integer address so that the same
// code can size a buffer from address 0 (the *_bytes functions) without doing pointer arithmetic on a null pointer.
template <typename T>
inline void carve(uintptr_t &p, T *&out, size_t count) {
  p = align_up(p, 256);
  out = reinterpret_cast<T *>(p);
  p += count * sizeof(T);
}
inline uintptr_t carve_begin(const void *chunk) { return reinterpret_cast<uintptr_t>(chunk); }

inline int pre_blocks(int P) { return (P + PRE_BLOCK - 1) / PRE_BLOCK; }
constexpr int SORT_ITEMS = 16;                        // keys per thread in the radix passes
constexpr int SORT_BLOCK = 256;                       // threads
constexpr int SORT_TILE = SORT_ITEMS * SORT_BLOCK;    // keys per block
inline size_t sort_blocks(size_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
inline size_t sort_hist_words(size_t n) { return 256 * (sort_blocks(n) ? sort_blocks(n) : 1) + 256; }

inline GeomState geom_from_chunk(char *chunk_, size_t P, size_t *end = nullptr) {
  GeomState g;
  uintptr_t chunk = carve_begin(chunk_);
  size_t nb = (size_t)pre_blocks((int)P);
  carve(chunk, g.recs, P);
  carve(chunk, g.cov3D, P * 6);
  carve(chunk, g.clamped, P);
  carve(chunk, g.tiles_touched, P);
  carve(chunk, g.point_offsets, P);
  carve(chunk, g.internal_radii, P);
  carve(chunk, g.block_incl, P);
  carve(chunk, g.block_sums, nb);
  carve(chunk, g.block_prefix, nb);
  carve(chunk, g.total, 4);
  carve(chunk, g.grad_rows, P * GROWX);
  if (end) *end = chunk;
  return g;
}
inline size_t geom_bytes(size_t P) {
  size_t end = 0;
  geom_from_chunk(nullptr, P, &end);
  return end + 256;  // slack for a chunk that is not itself 256-byte aligned
}
inline BinningState binning_from_chunk(char *chunk_, size_t R, size_t tiles = 0, size_t *end = nullptr) {
  BinningState b;
  uintptr_t chunk = carve_begin(chunk_);
  size_t n = R ? R : 1;
  carve(chunk, b.keys_a, n);
  carve(chunk, b.vals_a, n);
  carve(chunk, b.keys_s, n);
  carve(chunk, b.vals_s, n);
  carve(chunk, b.hist, sort_hist_words(n));
  b.tile_counts = b.tile_cursor = nullptr;
  if (tiles) {
    carve(chunk, b.tile_counts, tiles * 16);  // one counter per 64-byte line (binning_bucket.hip CSTRIDE)
    carve(chunk, b.tile_cursor, tiles);
  }
  if (end) *end = chunk;
  return b;
}
inline size_t binning_bytes(size_t R, size_t tiles) {
  size_t end = 0;
  binning_from_chunk(nullptr, R, tiles ? tiles : 1, &end);
  return end + 256;
}
// visiting order of the tiles (ImageState::order, Options::tile_order):
//   1  tiles sorted by list length, longest first, slot k runs on XCD k % 8 (round 2)
//   2  2 x 2 tile SUPER-BLOCKS sorted by their summed list lengths, dealt round-robin to the XCDs with the tiles of a block on ONE
//      XCD -- neighbouring tiles share most of their Gaussians, so a record is fetched into one L2 instead of up to four
//   3  the same with 4 x 2 blocks (eight tiles)
// Slot layout for the block modes: block of sorted rank q occupies slots (q / 8) * 8 T + j * 8 + q % 8, j < T tiles per block:
// slot % 8 = q % 8 = the XCD (hardware runs workgroup i on XCD i % 8 and both blend kernels keep slot % 8 = workgroup % 8).
// Partial blocks at the image border and the last partial group of eight are padded with ORDER_NO_TILE entries.
#if defined(__HIPCC__)
#define GSR_HD __host__ __device__
#else
#define GSR_HD
#endif
constexpr uint32_t ORDER_NO_TILE = 0xFFFFFFFFu;
constexpr int ORDER_HDR = 2;  // order[0] = mode word, order[1] = visiting slots in use (mode 1), entries from order[2]

// ---- list segments (mode 1 only; Options::blend_segments) ------------------------------------------------------------------
// A wave walks its tile's list as one dependent chain, so a blend kernel lasts at least as long as its longest list takes BY ITSELF:
// in a close-up of a body (1,480 of 4,096 tiles busy, mean list 469, longest 1,398) the backward needs 214 of its 283 us for the
// tiles within 25 % of the longest alone (profiles/r3d_lone_wave.txt).  The backward has no early exit to respect -- final_T and
// n_contrib are known -- so the list of an outlier tile is cut into up to four SEGMENTS walked by different waves at the same time.
// What a segment needs to start in the middle of the list is the pixel's state at its far boundary b: the transmittance in front of
// entry b and the colour "behind" it, sum_{k >= b} w_k (c_k . dL_dpixel) = dL_dpixel . (C_final - C_prefix(b)) for every blended
// channel.  The FORWARD wave of such a tile therefore writes its accumulators (T, C[3], depth, weight sum, 18 extra channels) at
// every boundary and once more at the end: the checkpoint records.  Who is cut and where is decided once per frame by the
// workgroup that builds the visiting order (it sees every list length) and travels in the order entries, so forward and backward
// agree by construction:  entry = tile | segment << 22 | (segments - 1) << 25.
constexpr uint32_t ORDER_TILE_MASK = (1u << 22) - 1u;
constexpr int SEG_MAX = 4;
constexpr int CKPT_PLANES = 6 + CE_MAX;  // T, C0, C1, C2, depth sum, weight sum, extra channels
GSR_HD inline uint32_t order_entry(uint32_t tile, uint32_t seg, uint32_t nseg) { return tile | (seg << 22) | ((nseg - 1u) << 25); }
GSR_HD inline uint32_t order_entry_tile(uint32_t e) { return e & ORDER_TILE_MASK; }
GSR_HD inline uint32_t order_entry_seg(uint32_t e) { return (e >> 22) & 7u; }
GSR_HD inline uint32_t order_entry_nseg(uint32_t e) { return ((e >> 25) & 7u) + 1u; }
// entries of one segment (a multiple of the 64-entry batch: the forward writes its checkpoints between batches)
GSR_HD inline int segment_len(int n, int nseg) { return ((n + nseg - 1) / nseg + 63) & ~63; }
// the most EXTRA visiting slots (segments beyond a tile's first) a frame may use, and the checkpoint records that go with them
// (a tile cut into s segments takes s records: s - 1 boundaries + the final state, <= 2 per extra slot)
// (tiles / 8: the close-up of a body uses 445 extra slots for its 4,096 tiles at the default threshold; when they run out the remaining
// long lists are walked whole.  Every extra slot costs empty workgroups in uniform frames and 2 x 24 KB of checkpoint pool.)
GSR_HD inline uint32_t seg_extra_max(uint32_t tiles) { return tiles / 8u; }
GSR_HD inline size_t ckpt_records(size_t tiles) { return 2 * (size_t)seg_extra_max((uint32_t)tiles); }
// mode word + the most visiting slots any mode needs: 4 x 2 blocks on a grid one tile wide pad every block from two tiles to eight
// ((nsb + 7) * 8 <= (gx + 3)(gy + 1) + 56 <= 4 tiles + 63)
inline size_t order_words(size_t tiles) { return 4 * tiles + 64 + ORDER_HDR; }
GSR_HD inline int order_block_tiles(int mode) { return mode == 2 ? 4 : (mode == 3 ? 8 : 1); }
// visiting slots of a tile grid under a tile_order mode (>= grid_x * grid_y); the mode a frame was binned with is only known on
// the device (order[0]), so the blend kernels are LAUNCHED over tile_slots_max() and bound themselves by tile_slots(mode word)
GSR_HD inline uint32_t tile_slots(int grid_x, int grid_y, int mode) {
  const uint32_t tiles = (uint32_t)grid_x * (uint32_t)grid_y;
  if (mode == 1) return tiles + seg_extra_max(tiles);  // (upper bound: the frame's own count is order[1], tile_slots_of)
  if (mode < 2) return tiles;
  const int bx = mode == 3 ? 4 : 2, by = 2;
  const uint32_t nsb = (uint32_t)((grid_x + bx - 1) / bx) * (uint32_t)((grid_y + by - 1) / by);
  return (nsb + 7u) / 8u * 8u * (uint32_t)(bx * by);
}
inline uint32_t tile_slots_max(int grid_x, int grid_y) {
  uint32_t m = tile_slots(grid_x, grid_y, 0);
  for (int mode = 1; mode <= 3; mode++) m = tile_slots(grid_x, grid_y, mode) > m ? tile_slots(grid_x, grid_y, mode) : m;
  return m;
}

inline ImageState image_from_chunk(char *chunk_, size_t npix, size_t tiles, size_t *end = nullptr) {
  ImageState s;
  uintptr_t chunk = carve_begin(chunk_);
  carve(chunk, s.final_T, npix);
  carve(chunk, s.n_contrib, npix);
  carve(chunk, s.ranges, tiles);
  carve(chunk, s.order, order_words(tiles));
  carve(chunk, s.ckpt_base, tiles);
  carve(chunk, s.ckpt, ckpt_records(tiles) * (size_t)CKPT_PLANES * 256);
  if (end) *end = chunk;
  return s;
}
inline size_t image_bytes(size_t npix, size_t tiles) {
  size_t end = 0;
  image_from_chunk(nullptr, npix, tiles, &end);
  return end + 256;
}

// ---- error plumbing (gsr_api.hip) -------------------------------------------------------------
void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what, const char *file, int line);
#define GSR_HIP(expr)                                                        \
  do {                                                                       \
    int _rc = ::gsr::check_hip((expr), #expr, __FILE__, __LINE__);           \
    if (_rc != GSR_OK) return _rc;                                           \
  } while (0)
// after a kernel launch: always collect launch errors; in debug mode also synchronise (CR/auxiliary.h:166-173)
#define GSR_LAUNCH_CHECK(stream, debug)                                      \
  do {                                                                       \
    GSR_HIP(hipGetLastError());                                              \
    if (debug) GSR_HIP(hipStreamSynchronize(stream));                        \
  } while (0)

// ---- optional per-stage HIP-event timing (gsr_api.hip): bench.py brackets the dominant kernel with events on the
// launch stream inside its timed region.  Disabled (zero cost) unless gsr_profile_enable() selected the stage.
enum ProfStage { PROF_PREPROCESS_FWD = 0, PROF_SCAN = 1, PROF_BINNING = 2, PROF_BLEND_FWD = 3, PROF_BLEND_BWD = 4,
                 PROF_PREPROCESS_BWD = 5, PROF_NSTAGES = 6 };
void prof_begin(int stage, hipStream_t stream);
void prof_end(int stage, hipStream_t stream);

// ---- per-call options (gsr_api.hip) ----------------------------------------------------------------------------------
// Every knob a call consults.  A call resolves them ONCE, at entry, from its stream: options set for that stream with
// gsr_set_stream_tuning, else the process defaults (gsr_set_tuning / gsr_set_binning_mode).  Nothing below the API layer
// reads mutable global state, so calls on different streams -- from any threads, the autograd backward thread included --
// do not see each other's knobs.
struct Options {
  int binning_mode = GSR_BINNING_TILE_BUCKET;
  int tile_cull = 1;         // exact ellipse-vs-tile culling of instances in the tile-bucket back-end
  int tile_order = 1;        // blend kernels visit the tiles longest list first, spread over the XCDs (1, default) or in the natural order (0)
  int bucket_hist = 1;       // atomics-free counting of the tile-bucket back-end (LDS histograms per workgroup); 0 = global atomics
  int bucket_sort_merged = 1;  // histogram path: the per-tile sorts of short and long lists as ONE launch (0: two launches)
  int bucket_cstride = 4;    // counters per 64-byte line = 16 / stride (interleaved A/B at C3, us of binning: 1: 112, 2: 106, 4: 100, 16: 126)
  int blend_fwd_dma = 0;     // 1: fused multi-feature forward with the feature rows staged half a batch ahead by global -> LDS DMA
                             // (parity-green experiment, SLOWER: 191 vs 144 us in the render() frame, profiles/r3_fwd_dma_experiment.txt)
  int blend_fwd_waves = 4;   // waves that cooperate on one 16x16 tile
  int blend_bwd_waves = 4;
  int blend_bwd_reduce = 3;  // 3 LDS folds (default), 0 permlane / DPP folds, 1 MFMA on folded rows, 2 transposed MFMA contraction
  int deterministic = 0;     // backward: fixed-order reduction of the gradient rows instead of float atomics
  int blend_layout = 0;      // 0: a wave per 8x8 quadrant, one survivor at a time; 1: a wave per 4x4 block, four survivors per step (forward)
  int blend_tail_cut = 0;    // sixteenths of the busy tiles, counted from the END of the visiting order, whose lists are cut in two as well
                             // (measured at C3 with a pool of tiles / 4: 4/16 = +-0, 6/16 and 8/16 = -1 .. -3 %: off)
  int blend_segments = 8;    // > 0: lists of at least blend_segments / 4 x the frame's mean list are walked in segments by the backward
                             // (forward checkpoints), see ORDER_HDR; 0: never
  int blend_prio = 1;        // 1: blend waves take an issue priority from the length of their tile's list (list_priority)
  int debug_no_atomics = 0;  // MEASUREMENT ONLY: the plain blend backward without its gradient-row atomics (gradients are wrong)
};
Options options_for(hipStream_t stream);

// ---- kernel launchers (one translation unit each) ---------------------------------------------
struct PreprocessArgs {
  int P, D, M;
  const float *means3D, *scales, *rotations, *opacities, *shs, *cov3D_precomp, *colors_precomp;
  float scale_modifier;
  const float *view, *proj, *campos;  // device pointers, read with scalar loads (wave-uniform)
  int W, H, grid_x, grid_y;
  float tan_fovx, tan_fovy, focal_x, focal_y;
  int *radii;
  GeomState geom;
  int prefiltered;
  int sh_half;  // 1: shs points at IEEE half coefficients [P][16][3] (extension: fp16 SH storage), converted while staging
  int zero_rows;  // 1: also zero the Gaussian's gradient accumulation row (GSR_FWD_ZERO_ROWS)
};
int launch_preprocess_forward(const PreprocessArgs &a, hipStream_t stream);
int launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t stream);
int launch_scan_block_sums(const GeomState &g, int P, hipStream_t stream);
int launch_duplicate(const GeomState &g, const int *radii, int P, int grid_x, int grid_y, uint64_t *keys,
                     uint32_t *vals, hipStream_t stream);
// generic stable LSD radix sort: pass 0 reads in_*, writes x_*; then ping-pongs x <-> y. Result is in x if the
// number of passes is odd, else in y.  Returns the number of passes through *passes_out.
int radix_sort_u64(size_t n, const uint64_t *in_k, const uint32_t *in_v, uint64_t *x_k, uint32_t *x_v, uint64_t *y_k,
                   uint32_t *y_v, int end_bit, uint32_t *hist, hipStream_t stream, int debug);
int radix_sort_u32(size_t n, const uint32_t *in_k, const uint32_t *in_v, uint32_t *x_k, uint32_t *x_v, uint32_t *y_k,
                   uint32_t *y_v, int end_bit, uint32_t *hist, hipStream_t stream, int debug);
inline int radix_passes(int end_bit) { return (end_bit + 7) / 8; }
int launch_tile_ranges(size_t R, const uint64_t *keys_sorted, uint2 *ranges, size_t tiles, hipStream_t stream);

struct BlendFwdArgs {
  const uint32_t *order;  // ImageState::order (null: natural order)
  const uint2 *ranges;
  const uint32_t *point_list;
  const SplatRec *recs;
  int W, H, grid_x, grid_y;
  const float *bg;  // device pointer [3]
  float *out_color, *out_depth, *out_alpha, *final_T;
  uint32_t *n_contrib;
  const float *extra;  // [P][CE] extra feature channels blended with the same weights (null: none)
  int CE;              // 0 or CE_MAX
  float *out_extra;    // [CE][H][W]
  int list_prio;       // Options::blend_prio
  const uint32_t *ckpt_base;  // ImageState (null: this kernel variant writes no checkpoints -- the frame must then have no segments)
  float *ckpt;
  unsigned long long *trace;  // measurement (gsr_debug_wave_trace): per wave {start, end (100 MHz ticks), list length, batches walked}
};
int launch_blend_forward(const BlendFwdArgs &a, const Options &opt, hipStream_t stream);
// is ImageState::order to be used (its mode word)?  Then the work items are NOT remapped to keep neighbouring tiles on one XCD:
// the long lists at the front of the order must spread over all eight XCDs (hardware assigns workgroup i to XCD i % 8).  With
// the remap they all landed on XCD 0: 377 instead of 212 us in the render() frame.
// (the mode word: bits 0..7 the mode, bits 8..31 the frame's longest per-tile list, clamped)
__device__ __forceinline__ int tile_order_mode(const uint32_t *order) { return order ? (int)(order[0] & 0xFFu) : 0; }
__device__ __forceinline__ uint32_t tile_order_longest(const uint32_t *order) { return order ? order[0] >> 8 : 0u; }
// A blend wave's issue priority from the length of its tile's list (knob "blend_prio").  When the lists of a frame are very
// unequal (a body: 1,480 of 4,096 tiles busy, mean list 469, longest 1,398) the kernel lasts as long as the wave with the longest
// list, and that wave spends the first half of its life sharing its SIMD's issue slots with four short-list waves that could as well
// run later (tools/tile_cost_census.py: the busiest slot walks 1,107 entries in ANY visiting order, the mean slot 461).  s_setprio lets
// the instruction arbiter pick the long-list wave first whenever it has an instruction ready.
__device__ __forceinline__ void list_priority(const uint32_t *order, int n, int enabled) {
  const uint32_t longest = tile_order_longest(order), n4 = 4u * (uint32_t)n;
  if (!enabled || longest == 0u) return;
  // MEASUREMENT ONLY (knob values 2 / 3 / 4: results are WRONG): render nothing but the tiles whose list is within 25 / 50 / 75 % of the
  // longest -- what the longest lists cost when they have the GPU to themselves (profiles/r3d_lone_wave.txt)
#ifdef GSR_BUILD_EXPERIMENTS
  if (enabled >= 2 && n4 < (uint32_t)(enabled == 2 ? 3 : (enabled == 3 ? 2 : 1)) * longest) __builtin_amdgcn_endpgm();
#endif
  if (n4 >= 3u * longest) __builtin_amdgcn_s_setprio(3);
  else if (n4 >= 2u * longest) __builtin_amdgcn_s_setprio(2);
  else if (n4 >= longest) __builtin_amdgcn_s_setprio(1);
}
// visiting slots of THIS frame (mode 1: tiles + the extra segments the order builder made)
__device__ __forceinline__ uint32_t tile_slots_of(const uint32_t *order, int grid_x, int grid_y, int mode) {
  return mode == 1 ? order[1] : tile_slots(grid_x, grid_y, mode);
}
// the entry of a visiting slot (ORDER_NO_TILE: padding, or a slot beyond what this frame's mode uses); see order_entry_*
__device__ __forceinline__ uint32_t tile_of_slot(const uint32_t *order, int mode, uint32_t slot, uint32_t n_slots) {
  if (slot >= n_slots) return ORDER_NO_TILE;
  return mode ? order[ORDER_HDR + slot] : slot;
}
// ordered visiting with four waves per tile: workgroup i -> (order slot, quadrant) such that consecutive slots go to different
// XCDs (i % 8) while the four quadrants of a slot share one (their Gaussians are fetched into one L2, not four)
__device__ __forceinline__ uint32_t ordered_item4(uint32_t i, uint32_t n_slots) {
  const uint32_t full = (n_slots / 8u) * 32u;  // workgroups of the complete groups of 8 slots x 4 quadrants
  if (i >= full) return i;                     // the last, partial group: plain (slot = i / 4, quadrant = i % 4)
  const uint32_t slot = (i / 32u) * 8u + (i % 8u), part = (i / 8u) % 4u;
  return slot * 4u + part;
}

struct BlendBwdArgs {
  const uint32_t *order;  // ImageState::order (null: natural order)
  const uint2 *ranges;
  const uint32_t *point_list;
  const SplatRec *recs;
  int W, H, grid_x, grid_y;
  const float *bg;  // device pointer [3]
  const float *final_T;
  const uint32_t *n_contrib;
  const float *dL_dpix, *dL_ddepth, *dL_dalpha;
  // loss_gt != null: the image gradients are those of the alpha-mask loss  mean|color - gt| + lambda mean (alpha - mask)^2  and
  // are formed per pixel in the kernel's prologue (same expressions as loss.hip) instead of being read: dL_dpix = +-loss_sc,
  // dL_dalpha = loss_sa (alpha - mask), dL_ddepth = 0; dL_dpix / dL_ddepth / dL_dalpha are not touched then
  const float *loss_color, *loss_alpha, *loss_gt, *loss_mask;
  float loss_sc, loss_sa;
  float *grad_rows;  // [P][GROW] (CE == 0) or [P][GROWX] (CE > 0), zeroed
  const float *extra;          // [P][CE]
  int CE;
  const float *dL_dextra_tri[CE_MAX / 3];  // per colour triple: [3][H][W] gradient image, null = no gradient
  uint32_t extra_mask;         // bit t: colour triple t (channels 3t..3t+2) has an incoming gradient
  // deterministic mode (Options::deterministic): the nine sums of a (Gaussian, tile, quadrant) go to their own 64-byte slot
  // det_rows[((point_offsets[g] - tiles_touched[g] + k) * 4 + quadrant)][GROW], k = index of the tile in the Gaussian's
  // rectangle, instead of being added atomically; launch_reduce_det_rows sums a Gaussian's slots in index order afterwards
  float *det_rows;             // null = atomics
  const int *radii;
  const uint32_t *point_offsets, *tiles_touched;
  int list_prio;               // Options::blend_prio
  const uint32_t *ckpt_base;   // ImageState: checkpoint records of the tiles the forward cut into segments
  const float *ckpt;
  unsigned long long *trace;   // measurement (gsr_debug_wave_trace), as BlendFwdArgs::trace
  int debug_skip_atomics;      // measurement knob "debug_no_atomics" (LDS-fold plain kernel only): results are WRONG when set
  // fused phase-1 training loss (gsr_rasterize_backward_phase1_loss; 18-channel kernels only): p1.gt_image != null -> the image
  // gradients are FORMED in the prologue from the forward's images and the targets, and dL_dpix / dL_ddepth / dL_dalpha /
  // dL_dextra_tri[] (each may then be null) are added on top
  gsr_phase1_loss p1;
};
int launch_reduce_det_rows(int P, const uint32_t *point_offsets, const uint32_t *tiles_touched, const float *det_rows,
                           size_t n_slots, float *grad_rows, hipStream_t stream);
int launch_blend_backward(const BlendBwdArgs &a, const Options &opt, hipStream_t stream);

struct PreprocessBwdArgs {
  int P, D, M;
  const float *means3D, *shs, *scales, *rotations, *cov3D;  // cov3D = precomputed or geom.cov3D
  const int *radii;
  const uint8_t *clamped;
  float scale_modifier;
  const float *view, *proj, *campos;
  int W, H;
  float tan_fovx, tan_fovy, focal_x, focal_y;
  float *grad_rows;
  int grow;          // row stride of grad_rows (GROW or GROWX)
  int clear_rows;    // 1: leave the rows zero again once they are consumed (GSR_BWD_ROWS_ZEROED: no memset in the next backward)
  int CE;            // extra feature channels (their gradients sit in row columns 9 .. 9+CE-1)
  float *dL_dextra;  // [P][CE]
  const SplatRec *recs;
  float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
  int sh_half;  // 1: shs are IEEE halves (see PreprocessArgs)
};
int launch_preprocess_backward(const PreprocessBwdArgs &a, hipStream_t stream);

int launch_query_recs(int what, int P, const GeomState &g, void *dst, hipStream_t stream);

// tile-bucket binning (binning_bucket.hip)
// capacity = instances the binning buffer holds.  device_sized: the host does not know R; the kernels read it from
// g.total, write dev_status[0] = R, dev_status[1] = (R > capacity) | 2 * (prefilter violation) and render nothing on overflow.
int bucket_binning(const GeomState &g, const int *radii, int P, int grid_x, int grid_y, size_t capacity, bool device_sized,
                   BinningState &b, uint2 *ranges, uint32_t *order, uint32_t *ckpt_base, int segments, uint32_t *dev_status,
                   bool check_prefilter, bool scan_fused, const Options &opt, hipStream_t stream, int debug);
// true if bucket_binning will take its atomics-free histogram path (which can also do the block-sums scan: scan_fused)
bool bucket_uses_hist(const Options &opt, int P, size_t tiles, size_t capacity);

}  // namespace gsr

// ---- device helpers -----------------------------------------------------------------------------
#if defined(__HIPCC__)
namespace gsr {

// float -> int like v_cvt_i32_f32 / cvt.rzi.s32.f32: truncate, saturate, NaN -> 0 (explicit so the oracle and the
// kernels agree for any input).
__device__ __forceinline__ int f2i_sat(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}

// CR/auxiliary.h:46-56
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int &x0, int &y0, int &x1,
                                          int &y1) {
  const float r = (float)radius;
  x0 = min(gx, max(0, f2i_sat((px - r) / (float)TILE)));
  y0 = min(gy, max(0, f2i_sat((py - r) / (float)TILE)));
  x1 = min(gx, max(0, f2i_sat((px + r + (float)(TILE - 1)) / (float)TILE)));
  y1 = min(gy, max(0, f2i_sat((py + r + (float)(TILE - 1)) / (float)TILE)));
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Conservative "can this Gaussian reach alpha >= 1/255 anywhere in the pixel rectangle [x0,x1]x[y0,y1]?" used by the
// blend kernels when they compact a tile's list for one wave (never changes a result: it only removes entries whose
// every pixel test would say "skip").  alpha >= 1/255 needs q(d) = A dx^2 + 2 B dx dy + C dy^2 <= 2 ln(255 o); the
// minimum of the convex quadratic over the rectangle is 0 if the centre is inside, else it lies on one of the four
// edges (1-D quadratic, clamped).  Falls back to "keep" for a non positive-definite conic.
__device__ __forceinline__ bool ellipse_hits_rect(float gx, float gy, float A, float B, float C, float opacity, float x0,
                                                  float x1, float y0, float y1) {
  if (!(A > 0.f && C > 0.f && A * C - B * B > 0.f)) return true;
  const float thr = 2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(255.0f * opacity) * 1.0001f + 0.05f;
  // offsets d = g - p over the rectangle: dx in [gx - x1, gx - x0], dy in [gy - y1, gy - y0]
  const float dxl = gx - x1, dxh = gx - x0, dyl = gy - y1, dyh = gy - y0;
  if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return true;  // centre inside
  float qmin = 3.0e38f;
  const float rA = 1.0f / A, rC = 1.0f / C;
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const float dx = e ? dxh : dxl;  // vertical edges: dx fixed, best dy = -B dx / C clamped
    const float dy = fminf(dyh, fmaxf(dyl, -B * dx * rC));
    qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
    const float ey = e ? dyh : dyl;  // horizontal edges: dy fixed, best dx = -B dy / A clamped
    const float ex = fminf(dxh, fmaxf(dxl, -B * ey * rA));
    qmin = fminf(qmin, A * ex * ex + 2.0f * B * ex * ey + C * ey * ey);
  }
  return qmin <= thr;
}

// The same test for the blend kernels' per-wave compaction, where it is pure work saving (an entry kept in error costs a cut-off
// test, nothing else, and the 1.0001 / 0.05 margins of the threshold cover a rounding in 1/A, 1/C): hardware reciprocals instead of
// two IEEE divisions (2 instead of ~22 VALU instructions per batch of 64 list entries) and log2(255 o) handed in by the caller, who
// needs it for the survivor's LDS row anyway.  The binning's tile culling keeps ellipse_hits_rect: its decisions define the lists.
__device__ __forceinline__ bool ellipse_hits_rect_fast(float gx, float gy, float A, float B, float C, float log2_255o, float x0,
                                                       float x1, float y0, float y1) {
  if (!(A > 0.f && C > 0.f && A * C - B * B > 0.f)) return true;
  const float thr = 2.0f * 0.6931471805599453f * log2_255o * 1.0001f + 0.05f;
  const float dxl = gx - x1, dxh = gx - x0, dyl = gy - y1, dyh = gy - y0;
  if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return true;  // centre inside
  float qmin = 3.0e38f;
  const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const float dx = e ? dxh : dxl;
    const float dy = fminf(dyh, fmaxf(dyl, -B * dx * rC));
    qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
    const float ey = e ? dyh : dyl;
    const float ex = fminf(dxh, fmaxf(dxl, -B * ey * rA));
    qmin = fminf(qmin, A * ex * ex + 2.0f * B * ex * ey + C * ey * ey);
  }
  // (a clamped minimiser that is off by an ulp of 1/A still lies ON the edge: q there is >= the edge's true minimum, and the margins
  // of thr are orders of magnitude above that difference)
  return qmin <= thr * 1.00001f + 1e-3f;
}

// Zero-fill as a KERNEL, not as a hipMemsetAsync: under the HIP runtime's graph packet capture (ROCm 7.2 default) a memset NODE
// on memory of a hipGraph's private pool replays wrong once other GPU work has run between two replays
// (tools/graph_bisect.py, profiles/r3_graph_bisect.txt: a bare hipMemsetAsync + one torch add, no libgsr involved, is off by
// 3e32; a kernel writing the same memory is right).  The library therefore records no memset nodes at all.
static __global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *p, size_t n_words) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n_words; i += stride) {
    if (i + 4 <= n_words && (reinterpret_cast<uintptr_t>(p + i) & 15) == 0) {
      *reinterpret_cast<uint4 *>(p + i) = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (size_t k = i; k < n_words && k < i + 4; k++) p[k] = 0u;
    }
  }
}
// bytes must be a multiple of 4 (every array of the library is made of 4-byte words or larger)
static inline hipError_t zero_async(void *p, size_t bytes, hipStream_t stream) {
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  const size_t groups = (n + 1023) / 1024;  // 256 threads x 4 words
  const unsigned grid = (unsigned)(groups < 2048 ? groups : 2048);
  hipLaunchKernelGGL(zero_words_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(p), n);
  return hipGetLastError();
}

// wave64 inclusive scan (uint32 add) with shuffles
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t o = __shfl_up(v, d, WAVE);
    if (lane >= (uint32_t)d) v += o;
  }
  return v;
}

}  // namespace gsr
#endif
