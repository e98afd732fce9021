/*
 * gsr.h -- C ABI of libgsr.so: the MI355X (gfx950) articulated Gaussian-splat hot path.
 *
 * Drop-in boundary.  Every entry point replaces one interface of the reference
 * (paths relative to the reference tree; DGR = submodules/diff-gaussian-rasterization,
 * CR = DGR/cuda_rasterizer, SK = submodules/simple-knn):
 *
 *   gsr_rasterize_forward   <- CudaRasterizer::Rasterizer::forward      CR/rasterizer.h:32-56, CR/rasterizer_impl.cu:198-341
 *   gsr_rasterize_backward  <- CudaRasterizer::Rasterizer::backward     CR/rasterizer.h:58-89, CR/rasterizer_impl.cu:345-447
 *   gsr_mark_visible        <- CudaRasterizer::Rasterizer::markVisible  CR/rasterizer.h:25-30, CR/rasterizer_impl.cu:141-153
 *   gsr_dist2               <- SimpleKNN::knn                           SK/simple_knn.h:17, SK/simple_knn.cu:185-221
 *   gsr_lbs_*               <- GaussianModel.coarse_deform_c2source     scene/gaussian_model.py:768-872
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is enqueued on it;
 *   - the three growable scratch buffers of the reference (geometry / binning / image state,
 *     std::function<char*(size_t)> callbacks in CR/rasterizer.h:33-35) are requested through
 *     gsr_alloc_fn callbacks: the library calls alloc(user, bytes) once per buffer per forward call
 *     and the caller returns a device pointer that stays valid until the matching backward;
 *     the contents are private to the library;
 *   - a null `shs` / `colors_precomp` / `scales` / `rotations` / `cov3D_precomp` pointer selects the
 *     other input mode exactly like the reference (CR/forward.cu:205,241);
 *   - every function returns GSR_OK (0) or a negative GSR_E* code; gsr_last_error() returns a
 *     thread-local message for the last failure on the calling thread.  With debug != 0 the
 *     library synchronises and checks after every kernel (CR/auxiliary.h:166-173).
 */
#ifndef GSR_H_INCLUDED
#define GSR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_EINVAL (-1)  /* bad argument (shape/size/null) */
#define GSR_EHIP (-2)    /* a HIP runtime call or kernel failed */
#define GSR_ENOMEM (-3)  /* an allocation callback returned null / workspace too small */

typedef void *gsr_stream_t;
typedef char *(*gsr_alloc_fn)(void *user, size_t bytes);

/* Library version (major*10000 + minor*100 + patch) and the gfx target it was built for ("gfx950"). */
int gsr_version(void);
/* 1 if the library was built with GSR_BUILD_EXPERIMENTS (python -m mygauhuman_amd.build --experiments): the kernels that were built,
 * measured and NOT adopted (the MFMA reductions, the global -> LDS DMA forward, the 4x4-block forward, round 3's
 * backward, the "render the longest lists alone" measurement) are then compiled in and selectable through gsr_set_tuning.  The
 * default build does not contain them, and their knob values are refused. */
int gsr_has_experiments(void);
const char *gsr_target_arch(void);
const char *gsr_last_error(void);

/* Binning back-ends (gsr_set_binning_mode; default GSR_BINNING_TILE_BUCKET):
 *   GSR_BINNING_GLOBAL_RADIX: duplicate keys + device-wide stable LSD radix sort on the low 32+bit key bits +
 *                             identifyTileRanges -- the reference's structure (CR/rasterizer_impl.cu:291-320);
 *   GSR_BINNING_TILE_BUCKET : per-tile counting + per-tile register / LDS sort on (depth bits, Gaussian id).
 * With the tuning knob "tile_cull" = 0 both produce the reference's point lists and tile ranges bit for bit.  With
 * "tile_cull" = 1 (default, tile-bucket only) instances whose tile the Gaussian cannot reach with alpha >= 1/255 are dropped:
 * images, radii and gradients are unchanged, the per-tile lists are sublists of the reference's. */
#define GSR_BINNING_GLOBAL_RADIX 0
#define GSR_BINNING_TILE_BUCKET 1
int gsr_set_binning_mode(int mode);
int gsr_get_binning_mode(void);

/* Knobs (images, radii and gradients never change beyond summation order):
 *   "binning_mode" (= gsr_set_binning_mode);
 *   "blend_fwd_waves" / "blend_bwd_waves" in {1, 2, 4}: waves that cooperate on one 16x16 tile (a lane owns 4 / waves pixels);
 *   "blend_bwd_reduce" in {0, 3}: cross-lane reduction of the backward: 3 = two hops through LDS (default: the per-Gaussian
 *       sums of a quadrant are formed by reader lanes from transposed (r, w) planes; plain-pass only, other configurations use 0),
 *       0 = v_permlane swaps + DPP on the VALU.  Experiment builds only (gsr_has_experiments): 1 = MFMA on the folded rows,
 *       2 = transposed MFMA contraction, 4 = round 3's kernel of the LDS design (all slower on gfx950);
 *   "bucket_hist" in {0, 1}: tile-bucket counting without global atomics (per-workgroup LDS histograms + a dense prefix table,
 *       default) or with one returning global atomic per instance (also taken for tile grids beyond 8192 tiles);
 *   "bucket_cstride" in {1, 2, 4, 8, 16}: spacing (in 4-byte words) of the per-tile counters of the atomic variant;
 *   "bucket_sort_merged" in {0, 1}: histogram path -- the sort of the short lists (a wave per tile) and of the long ones (a
 *       workgroup per list, from a work list) in one launch (default) or in two;
 *   "tile_order" in {0, 1, 2, 3}: the order in which the blend kernels visit the tiles (tile-bucket back-end, histogram path): 1 =
 *       longest list first, dealt round-robin to the XCDs (default), 0 = the natural order (a contiguous band of tile rows per
 *       XCD), 2 / 3 = blocks of 2 x 2 / 4 x 2 tiles by summed length, a block per XCD; results do not depend on it;
 *   "blend_segments" in 0 .. 64: 0 = every list is walked whole by the backward; v > 0 (default 8) = a list of at least v / 4 times
 *       the frame's mean list is walked in up to four segments by different waves, each started from a checkpoint of the blend
 *       state the forward writes at the segment boundary (gradients agree with the whole walk to rounding; needs tile_order 1);
 *   "blend_tail_cut" in 0 .. 16 (default 0), "blend_prio" in {0, 1} (default 1); experiment builds only: "blend_prio" 2..4
 *       (MEASUREMENT ONLY, renders wrong images), "blend_layout" = 1 (forward with a wave per 4x4 pixel block), "blend_fwd_dma" = 1
 *       (see DESIGN.md section 4);
 *   "tile_cull" in {0, 1}: exact ellipse-vs-tile culling of instances in the tile-bucket back-end (see above);
 *   "deterministic" in {0, 1}: the backward reduces its per-(Gaussian, tile-quadrant) partial sums in a fixed order instead of
 *       with float atomics: run-to-run bit-identical gradients (for tests; costs a 256-byte slot per instance quadrant).
 *       PLAIN PASS ONLY, by design: the mode exists so that tests can demand bit equality where two code paths must produce the same
 *       sums (the fused alpha-mask loss against loss kernel + backward; a knob against its default).  The fused 18-channel backward
 *       (gsr_rasterize_backward_ex with extra features) returns GSR_EINVAL under it: a slot would be 27 columns wide (432 bytes per
 *       instance quadrant, ~1 GB for a render() frame of 200k Gaussians), and what that pass has to be checked against -- seven
 *       plain passes and the oracle, at 1e-4 -- differs from it in summation order anyway (tests/test_gpu_multi.py,
 *       tests/test_gpu_render.py).
 * gsr_set_tuning sets the PROCESS DEFAULTS.  gsr_set_stream_tuning gives one stream its own set (initialised from the
 * defaults at its first call); every API call resolves its knobs once, at entry, from the stream it is given, so calls on
 * different streams are independent of each other whatever threads they come from (no mutable global state is consulted
 * below the API layer).  gsr_clear_stream_tuning returns a stream to the defaults. */
int gsr_set_tuning(const char *key, int value);
int gsr_set_stream_tuning(gsr_stream_t stream, const char *key, int value);
int gsr_clear_stream_tuning(gsr_stream_t stream);

/* Optional stage timing with HIP events recorded on the caller's stream around the selected stages' kernels
 * (no synchronisation until gsr_profile_read).  Stage ids: 0 forward preprocess, 1 scan, 2 binning (duplicate+sort+
 * ranges or the tile-bucket kernels), 3 blend forward, 4 blend backward, 5 backward preprocess.  stage_mask bit i
 * enables stage i; 0 disables.  gsr_profile_read returns the accumulated milliseconds and launch count of a stage. */
int gsr_profile_enable(unsigned stage_mask);
/* Measurement only: while a zero-filled device buffer is registered here, every wave of the blend kernels (default variants)
 * leaves {start, end} in 100 MHz ticks + its list length: forward waves in the first half of the buffer (4 words per wave,
 * wave index = workgroup * 4 + wave), backward waves in the second half.  words >= 2 * 16 * visiting slots of the image (32 * (4 * tiles
 * + 64) always suffices; a rasterizer call that finds the buffer too small fails with GSR_EINVAL); NULL switches it off. */
int gsr_debug_wave_trace(unsigned long long *device_buffer, size_t words);
/* Measurement only: the shader clock the device runs at right now.  `workgroups` (1 .. 2048) single-wave workgroups each run a
 * dependent chain of `fmas` v_fma_f32 bracketed by the constant 100 MHz counter and by s_memtime (which ticks with the shader
 * clock on gfx950): device_out[2 i] = 10 ns ticks, device_out[2 i + 1] = shader cycles of workgroup i;
 * clock = cycles / (ticks x 10 ns).  bench.py prints it next to every figure that scales with the clock and runs the probe until
 * the clock has settled before its warm-up steps (a freshly leased MI355X needs ~20 ms of load to go from 2.26 to 2.39 GHz). */
int gsr_debug_clock_probe(int workgroups, int fmas, unsigned long long *device_out, gsr_stream_t stream);
int gsr_profile_reset(void);
int gsr_profile_read(int stage, double *total_ms, long *launches);

/* present[i] = (view-space z of means3D[i]) > 0.2          (CR/rasterizer_impl.cu:54-66, CR/auxiliary.h:139-164) */
int gsr_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                     uint8_t *present, gsr_stream_t stream);

/* Forward rasterisation.  Argument meaning and order follow Rasterizer::forward (CR/rasterizer.h:32-56):
 *   P Gaussians, D active SH degree, M SH coefficients per Gaussian (shs is [P][M][3]);
 *   background[3]; means3D[P][3]; colors_precomp[P][3]; opacities[P]; scales[P][3]; rotations[P][4] (r,x,y,z);
 *   cov3D_precomp[P][6]; viewmatrix/projmatrix[16] (row-vector convention); cam_pos[3];
 *   out_color[3][H][W], out_depth[H][W], out_alpha[H][W] (alpha = sum of blending weights), radii[P] (may be null).
 * Every element of the outputs is written when P > 0 (the reference binding zero-fills them first,
 * DGR/rasterize_points.cu:69-72; callers only need that for P == 0);
 * *host_num_rendered receives the number of (Gaussian, tile) instances (the reference's return value). */
int gsr_rasterize_forward(gsr_alloc_fn geometry_alloc, void *geometry_user, gsr_alloc_fn binning_alloc,
                          void *binning_user, gsr_alloc_fn image_alloc, void *image_user, int P, int D, int M,
                          const float *background, int width, int height, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *opacities, const float *scales,
                          float scale_modifier, const float *rotations, const float *cov3D_precomp,
                          const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                          float tan_fovy, int prefiltered, float *out_color, float *out_depth, float *out_alpha,
                          int *radii, int debug, int *host_num_rendered, gsr_stream_t stream);

/* Asynchronous forward (extension: no counterpart in the reference, which blocks on a device->host copy of R every
 * call, CR/rasterizer_impl.cu:283).  The caller owns the three scratch buffers (gsr_geometry_bytes(P),
 * gsr_binning_bytes(capacity, w, h), gsr_image_bytes(w, h)) and picks `binning_capacity` = the number of
 * (Gaussian, tile) instances the binning buffer can hold; nothing is read back and the host never waits.
 * dev_status is device uint32[2]: [0] = R, [1] = flags: bit 0 = R > capacity, in which case NOTHING was rendered (outputs
 * hold the background) and the call must be repeated with a larger capacity; bit 1 = a point was filtered although
 * `prefiltered` was set (the reference traps the device there, CR/auxiliary.h:156-160; here the point is culled and the call is
 * flagged -- the blocking entry points return GSR_EINVAL with the reference's message).  Always uses the tile-bucket binning.
 * The matching backward is gsr_rasterize_backward with R = binning_capacity. */
size_t gsr_geometry_bytes(int P);
size_t gsr_image_bytes(int width, int height);
size_t gsr_binning_bytes(size_t binning_capacity, int width, int height);
int gsr_rasterize_forward_async(char *geom_buffer, char *binning_buffer, size_t binning_capacity, char *image_buffer, int P,
                                int D, int M, const float *background, int width, int height, const float *means3D,
                                const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                                float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                                float tan_fovy, int prefiltered, float *out_color, float *out_depth, float *out_alpha,
                                int *radii, int debug, uint32_t *dev_status, gsr_stream_t stream);

/* Backward rasterisation, Rasterizer::backward (CR/rasterizer.h:58-89).  R = num_rendered of the forward call;
 * geom/binning/image buffers are the ones the forward call filled.  The library writes every element of the dL_d*
 * outputs it owns, zeros for culled Gaussians (the reference relies on zero-filled tensors,
 * DGR/rasterize_points.cu:159-167); dL_dsh / dL_dscale / dL_drot are only touched when shs / scales are given:
 * dL_dmean2D[P][3], dL_dconic[P][4], dL_dopacity[P], dL_dcolor[P][3],
 * dL_dmean3D[P][3], dL_dcov3D[P][6], dL_dsh[P][M][3], dL_dscale[P][3], dL_drot[P][4].
 * `alphas` is accepted and ignored like in the reference kernel (CR/backward.cu:410). */
int gsr_rasterize_backward(int P, int D, int M, int R, const float *background, int width, int height,
                           const float *means3D, const float *shs, const float *colors_precomp,
                           const float *alphas, const float *scales, float scale_modifier, const float *rotations,
                           const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                           const float *campos, float tan_fovx, float tan_fovy, const int *radii,
                           char *geom_buffer, char *binning_buffer, char *image_buffer, const float *dL_dpix,
                           const float *dL_ddepths, const float *dL_dalphas, float *dL_dmean2D, float *dL_dconic,
                           float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D,
                           float *dL_dsh, float *dL_dscale, float *dL_drot, int debug, gsr_stream_t stream);

/* Fused multi-feature variants (extension).  The reference's render() rasterises the same geometry
 * seven times per frame with different colours (gaussian_renderer/__init__.py:203-272).  The _ex entry points blend
 * n_extra extra colour channels (extra_features[P][n_extra], n_extra == 18 = six RGB triples, each triple composited
 * over `background` like the main colour) in the SAME pass: out_extra[n_extra][H][W]; the backward takes
 * dL_dout_extra = HOST array of n_extra / 3 device pointers, one [3][H][W] gradient image per colour triple, and returns
 * dL_dextra[P][n_extra], every geometric gradient being the sum over all images.  A null entry = that image received no
 * gradient: it is treated as zero and costs no work -- a training loss typically touches two or three of the six feature
 * images (train.py:256-286).  sh_dtype: GSR_SH_F32 or GSR_SH_F16 (how the `shs` pointer is to be read).
 * With extra_features == NULL / n_extra == 0 and GSR_SH_F32 the _ex entry points are
 * identical to the plain ones. */
#define GSR_SH_F32 0 /* shs: float32 [P][M][3] */
#define GSR_SH_F16 1 /* shs: IEEE half [P][16][3] (extension: fp16 SH storage, BASELINE configs[4]); M must be 16, the array
                      * 16-byte aligned; coefficients are widened exactly on load, dL_dsh stays float32 */
/* Flag for the `debug` argument of the backward entry points (bit 0 = the reference's debug mode: synchronise and check after
 * every kernel).  GSR_BWD_ROWS_ZEROED: the caller guarantees that the gradient accumulation rows inside geom_buffer are all zero
 * on entry -- true for a geometry buffer that was zero-filled once and has since only been used by backward calls carrying this
 * flag -- and the call leaves them zero again (the backward preprocess clears every row it consumes): no 64-byte-per-Gaussian
 * memset per backward.  Without the flag the rows are zeroed by the call, as before. */
#define GSR_BWD_ROWS_ZEROED 2
/* Flag for the `debug` argument of the FORWARD entry points (bit 0 = debug mode as above): the forward preprocess also zeroes the
 * gradient accumulation rows inside geom_buffer (128 B per Gaussian, written by the thread that writes the Gaussian's record
 * anyway), so that the backward of this frame may carry GSR_BWD_ROWS_ZEROED whatever the buffer held before -- for callers that
 * know a backward will follow (the autograd wrappers) and allocate a fresh geometry buffer per frame. */
#define GSR_FWD_ZERO_ROWS 2

int gsr_rasterize_forward_ex(gsr_alloc_fn geometry_alloc, void *geometry_user, gsr_alloc_fn binning_alloc, void *binning_user,
                             gsr_alloc_fn image_alloc, void *image_user, int P, int D, int M, const float *background, int width,
                             int height, const float *means3D, const float *shs, const float *colors_precomp,
                             const float *opacities, const float *scales, float scale_modifier, const float *rotations,
                             const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *cam_pos,
                             float tan_fovx, float tan_fovy, int prefiltered, float *out_color, float *out_depth,
                             float *out_alpha, int *radii, int debug, int *host_num_rendered, const float *extra_features,
                             int n_extra, float *out_extra, int sh_dtype, gsr_stream_t stream);
int gsr_rasterize_forward_async_ex(char *geom_buffer, char *binning_buffer, size_t binning_capacity, char *image_buffer, int P,
                                   int D, int M, const float *background, int width, int height, const float *means3D,
                                   const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                                   float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                   const float *viewmatrix, const float *projmatrix, const float *cam_pos, float tan_fovx,
                                   float tan_fovy, int prefiltered, float *out_color, float *out_depth, float *out_alpha,
                                   int *radii, int debug, uint32_t *dev_status, const float *extra_features, int n_extra,
                                   float *out_extra, int sh_dtype, gsr_stream_t stream);
int gsr_rasterize_backward_ex(int P, int D, int M, int R, const float *background, int width, int height, const float *means3D,
                              const float *shs, const float *colors_precomp, const float *alphas, const float *scales,
                              float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                              const float *projmatrix, const float *campos, float tan_fovx, float tan_fovy, const int *radii,
                              char *geom_buffer, char *binning_buffer, char *image_buffer, const float *dL_dpix,
                              const float *dL_ddepths, const float *dL_dalphas, float *dL_dmean2D, float *dL_dconic,
                              float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                              float *dL_dscale, float *dL_drot, int debug, const float *extra_features, int n_extra,
                              const float *const *dL_dout_extra, float *dL_dextra, int sh_dtype, gsr_stream_t stream);

/* The training loss render() feeds before the PBR phase (train.py:261-265 with utils/loss_utils.py:20-24), fused:
 *     L = w_image L1_b(image, gt_image) + w_alpha L2_b(alpha, alpha_target) + w_normal L1_b(normal, gt_normal) + w_axis L1_b(axis, gt_normal)
 * where L1_b / L2_b are the MEANS over the pixels with bound != 0 (the reference indexes with bound_mask == 1; a mean over n_bound x 3
 * colour values resp. n_bound alphas).  gsr_phase1_loss_forward reduces the forward's images to the loss value and leaves
 * (loss, n_bound, 1 / (3 n_bound), 1 / n_bound, the four unweighted terms) in stats[8] on the device -- no host read.
 * gsr_rasterize_backward_phase1_loss is gsr_rasterize_backward_ex whose blend-backward prologue FORMS this loss's per-pixel
 * gradient from the same images (bound . w sign(image - gt) / (3 n_bound) etc., times *upstream, the incoming dL/dL: null = 1)
 * instead of reading it: dL_dpix / dL_ddepths / dL_dalphas and the entries of dL_dout_extra may each be null there (no further
 * gradient on that image) or carry the gradient of OTHER loss terms (SSIM, LPIPS, ...), which is added.  What it saves per frame
 * against the same loss in torch ops: four masked reductions, their expand / divide / sign backward kernels and four full-image
 * gradient tensors written and read (~125 us of a 790 us render() frame at 1024^2, profiles/r3h_render_kernels.txt). */
typedef struct gsr_phase1_loss {
  const float *gt_image;      /* [3][H][W] */
  const float *gt_normal;     /* [3][H][W]: the target of BOTH the normal and the axis image (train.py:263-264) */
  const float *alpha_target;  /* [H][W]    (bkgd_mask[0]) */
  const float *bound;         /* [H][W]    != 0 inside bound_mask */
  float w_image, w_alpha, w_normal, w_axis;
  int normal_triple, axis_triple; /* which of the six extra colour triples are the normal / the axis image (render(): 0 and 5) */
  const float *color;         /* the forward's images: [3][H][W] */
  const float *alpha;         /* [H][W] */
  const float *extra_images;  /* [18][H][W] */
  float *stats;               /* device [8], written by gsr_phase1_loss_forward, read by the backward */
  const float *upstream;      /* device [1]: dL/dloss arriving from autograd (null = 1) */
} gsr_phase1_loss;
/* partials: device workspace of at least gsr_phase1_loss_partials() floats */
size_t gsr_phase1_loss_partials(void);
int gsr_phase1_loss_forward(int width, int height, const gsr_phase1_loss *loss, float *partials, gsr_stream_t stream);
int gsr_rasterize_backward_phase1_loss(int P, int D, int M, int R, const float *background, int width, int height, const float *means3D,
                                       const float *shs, const float *colors_precomp, const float *alphas, const float *scales,
                                       float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                       const float *viewmatrix, const float *projmatrix, const float *campos, float tan_fovx,
                                       float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer,
                                       char *image_buffer, const float *dL_dpix, const float *dL_ddepths, const float *dL_dalphas,
                                       float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                                       float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot,
                                       int debug, const float *extra_features, int n_extra, const float *const *dL_dout_extra,
                                       float *dL_dextra, int sh_dtype, const gsr_phase1_loss *loss, gsr_stream_t stream);

/* Fused gradient of L = mean|color - gt| + lambda_alpha * mean (alpha - mask)^2 (train.py:261-262 with the masks set to
 * the whole image): dL_dcolor[3][H][W] = sign(color - gt) / (3 H W), dL_dalpha[H][W] = 2 lambda (alpha - mask) / (H W). */
int gsr_alpha_mask_loss_backward(int width, int height, const float *color, const float *alpha, const float *gt,
                                 const float *mask, float lambda_alpha, float *dL_dcolor, float *dL_dalpha,
                                 gsr_stream_t stream);

/* gsr_rasterize_backward_ex with the image gradients of THAT loss formed inside the blend-backward kernel (per pixel, the same
 * expressions) instead of read from three gradient images: out_color / out_alpha are the forward's images, gt [3][H][W],
 * mask [1][H][W]; dL_ddepth = 0.  One launch and 48 B per pixel of traffic less than gsr_alpha_mask_loss_backward +
 * gsr_rasterize_backward_ex; gradients bit-identical to that pair.  (No extra feature channels in this entry.) */
int gsr_rasterize_backward_alpha_mask_loss(int P, int D, int M, int R, const float *background, int width, int height,
                                           const float *means3D, const float *shs, const float *colors_precomp,
                                           const float *out_alpha, const float *scales, float scale_modifier,
                                           const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                           const float *projmatrix, const float *campos, float tan_fovx, float tan_fovy,
                                           const int *radii, char *geom_buffer, char *binning_buffer, char *image_buffer,
                                           const float *out_color, const float *gt, const float *mask, float lambda_alpha,
                                           float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                                           float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot,
                                           int debug, int sh_dtype, gsr_stream_t stream);

/* Introspection of the private scratch buffers, for the parity tests only (copies device -> device):
 * what = one of GSR_Q_*; dst must hold the documented element count. */
#define GSR_Q_DEPTHS 0        /* float[P]            geom  */
#define GSR_Q_MEANS2D 1       /* float[P][2]         geom  */
#define GSR_Q_CONIC_OPACITY 2 /* float[P][4]         geom  */
#define GSR_Q_RGB 3           /* float[P][3]         geom  */
#define GSR_Q_COV3D 4         /* float[P][6]         geom  (only valid if computed from scales/rotations) */
#define GSR_Q_TILES_TOUCHED 5 /* uint32[P]           geom  */
#define GSR_Q_POINT_OFFSETS 6 /* uint32[P]           geom  (inclusive scan) */
#define GSR_Q_CLAMPED 7       /* uint8[P][3]         geom  */
#define GSR_Q_POINT_LIST 8    /* uint32[R]           binning (sorted Gaussian ids) */
#define GSR_Q_KEYS_SORTED 9   /* uint64[R]           binning (tile << 32 | depth bits) */
#define GSR_Q_RANGES 10       /* uint32[tiles][2]    image */
#define GSR_Q_FINAL_T 11      /* float[H][W]         image */
#define GSR_Q_N_CONTRIB 12    /* uint32[H][W]        image */
#define GSR_Q_ORDER 13        /* uint32[4 tiles + 66] image: [0] visiting mode | longest list << 8, [1] slots in use (mode 1),
                               * [2 ..] one entry per visiting slot: tile | segment << 22 | (segments - 1) << 25 */
int gsr_query_state(int what, int P, int R, int width, int height, const char *geom_buffer,
                    const char *binning_buffer, const char *image_buffer, void *dst, gsr_stream_t stream);

/* simple-knn: mean_dists[i] = mean of the 3 smallest squared distances from points[i] to the other points
 * (SK/simple_knn.cu:185-221).  workspace must hold gsr_dist2_workspace_bytes(P) bytes. */
size_t gsr_dist2_workspace_bytes(int P);
int gsr_dist2(int P, const float *points, float *mean_dists, char *workspace, size_t workspace_bytes,
              gsr_stream_t stream);

/* Stand-alone stable LSD radix sort of (key, value) pairs on key bits [0, end_bit), the replacement of
 * cub::DeviceRadixSort::SortPairs at CR/rasterizer_impl.cu:305-310 (u64 keys) and SK/simple_knn.cu:213 (u32 keys).
 * workspace: gsr_sort_workspace_bytes(n). Results land in keys_out / vals_out; inputs are preserved. */
size_t gsr_sort_workspace_bytes(size_t n);
int gsr_sort_pairs_u64(size_t n, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                       uint32_t *vals_out, int end_bit, char *workspace, size_t workspace_bytes,
                       gsr_stream_t stream);
int gsr_sort_pairs_u32(size_t n, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                       uint32_t *vals_out, int end_bit, char *workspace, size_t workspace_bytes,
                       gsr_stream_t stream);

/* SMPL linear-blend skinning of P canonical points (the per-point part of coarse_deform_c2source,
 * scene/gaussian_model.py:776-872, batch size 1).
 *   query[P][3], normals[P][3] (may be null); smpl_verts[V][3] = big-pose vertices searched for the nearest
 *   vertex (k = 1, lowest index wins ties); weights[V][24]; lbs_offsets[P][24] or null
 *   (softmax(log(w + 1e-9) + offset)); A_big / A_pose [24][16] joint transforms; off_big / off_shape / off_pose
 *   [V][3] per-vertex offset tables (PoseOff(theta_big), ShapeOff(beta), PoseOff(theta, dR)); R[9], Th[3].
 * Outputs (any may be null except world_pts): vert_ids int32[P], bweights[P][24], smpl_pts[P][3],
 *   world_pts[P][3], transforms[P][9], translation[P][3], world_normals[P][3]. */
int gsr_lbs_forward(int P, int V, const float *query, const float *normals, const float *smpl_verts,
                    const float *weights, const float *lbs_offsets, const float *A_big, const float *A_pose,
                    const float *off_big, const float *off_shape, const float *off_pose, const float *R,
                    const float *Th, int *vert_ids, float *bweights, float *smpl_pts, float *world_pts,
                    float *transforms, float *translation, float *world_normals, gsr_stream_t stream);

/* Same results as gsr_lbs_forward, bit for bit, with the nearest-vertex search run through a uniform grid over the V
 * reference vertices (rebuilt on every call by one workgroup) instead of the brute-force scan: ~30x fewer distance
 * evaluations when the query points lie near the vertex cloud, as canonical Gaussians do.  `workspace` must hold
 * gsr_lbs_workspace_bytes(V) bytes, 16-byte aligned.  grid_is_built != 0: the workspace already holds the grid of exactly
 * these vertices (gsr_lbs_grid_build, or an earlier call) and the rebuild is skipped -- the big-pose vertices of a subject
 * never change between frames. */
size_t gsr_lbs_workspace_bytes(int V);
int gsr_lbs_grid_build(int V, const float *smpl_verts, char *workspace, size_t workspace_bytes, gsr_stream_t stream);
int gsr_lbs_forward_grid(int P, int V, const float *query, const float *normals, const float *smpl_verts,
                         const float *weights, const float *lbs_offsets, const float *A_big, const float *A_pose,
                         const float *off_big, const float *off_shape, const float *off_pose, const float *R,
                         const float *Th, int *vert_ids, float *bweights, float *smpl_pts, float *world_pts,
                         float *transforms, float *translation, float *world_normals, char *workspace,
                         size_t workspace_bytes, int grid_is_built, gsr_stream_t stream);

/* gsr_lbs_forward_grid with an EXACT temporal cache of the nearest vertex (the reference searches every
 * frame, scene/gaussian_model.py:775).  The canonical points move by an optimizer step per iteration, the vertices not at all.
 * nn_cache (gsr_lbs_nn_cache_bytes(P) bytes, 16-byte aligned, caller-owned, tied to ONE vertex grid) holds per point slot
 * (x0, id, rho): vertex id is the strict nearest vertex of every point within rho of x0 -- rho = 0.49 x (lower bound on the
 * distance of every other vertex - distance of id), minus a guard for rounding -- which is a statement about the vertex set, so
 * it stays true whatever point sits in the slot.  cache_is_valid = 0: full grid search that also makes the entries.
 * cache_is_valid != 0: one kernel checks every point against its entry and searches (and re-centres) the misses, the
 * skinning takes the ids as given: results bit-identical to the search.  `workspace` must hold the BUILT grid
 * (gsr_lbs_grid_build).  The last 64 bytes of nn_cache are counters: word 1 = searches since the cache was made, word 2 = misses
 * of the last cached call. */
size_t gsr_lbs_nn_cache_bytes(int P);
int gsr_lbs_forward_cached(int P, int V, const float *query, const float *normals, const float *smpl_verts,
                           const float *weights, const float *lbs_offsets, const float *A_big, const float *A_pose,
                           const float *off_big, const float *off_shape, const float *off_pose, const float *R,
                           const float *Th, int *vert_ids, float *bweights, float *smpl_pts, float *world_pts,
                           float *transforms, float *translation, float *world_normals, char *workspace,
                           size_t workspace_bytes, char *nn_cache, size_t nn_cache_bytes, int cache_is_valid,
                           gsr_stream_t stream);

/* Backward of gsr_lbs_forward w.r.t. query, normals, lbs_offsets, A_pose, off_pose (A_big/off_big/off_shape
 * belong to the constant big pose / shape and get no gradient in the reference training loop).
 * Incoming: dL_dworld_pts[P][3], dL_dtransforms[P][9], dL_dworld_normals[P][3] (each may be null = zero).
 * Outgoing (zero-filled by the caller, accumulated into): dL_dquery[P][3], dL_dnormals[P][3] (or null),
 *   dL_dlbs_offsets[P][24] (or null), dL_dA_pose[24][16] (or null), dL_doff_pose[V][3] (or null).
 * dA_pose_partials (optional): [gsr_lbs_backward_workgroups(P)][24 * 12] floats.  When given (and dL_dA_pose is non-null)
 *   every workgroup stores its own sum there instead of issuing 288 atomics onto dL_dA_pose; the caller adds the rows up
 *   (entry j * 12 + k belongs to dL_dA_pose[j][k / 4][k % 4]) -- dL_dA_pose itself is then left untouched. */
int gsr_lbs_backward_workgroups(int P);
int gsr_lbs_backward(int P, int V, const float *query, const float *normals, const int *vert_ids,
                     const float *weights, const float *lbs_offsets, const float *A_big, const float *A_pose,
                     const float *off_big, const float *off_shape, const float *off_pose, const float *R,
                     const float *dL_dworld_pts, const float *dL_dtransforms, const float *dL_dworld_normals,
                     float *dL_dquery, float *dL_dnormals, float *dL_dlbs_offsets, float *dL_dA_pose,
                     float *dL_doff_pose, float *dA_pose_partials, gsr_stream_t stream);

/* Fused SSIM (extension): utils/loss_utils.py:25-66 -- 11x11 Gaussian window (sigma 1.5), zero
 * padding, C1 = 0.01^2, C2 = 0.03^2 -- over `planes` independent H x W planes (batch x channels of the reference's grouped
 * conv2d).  forward: ssim_map[planes][H][W] (may be null) and the three derivative maps dA, dB, dC (all three or none) that
 * the backward consumes.  backward: dL_dimg1 from dL_dmap[planes][H][W], or from the constant dL_dmap_scalar when dL_dmap is
 * null (the mean reduction of the reference: scalar = upstream / (planes * H * W)).  img2 is the ground truth: no gradient. */
int gsr_ssim_forward(int planes, int height, int width, const float *img1, const float *img2, float *ssim_map, float *dA,
                     float *dB, float *dC, gsr_stream_t stream);
int gsr_ssim_backward(int planes, int height, int width, const float *img1, const float *img2, const float *dL_dmap,
                      float dL_dmap_scalar, const float *dA, const float *dB, const float *dC, float *dL_dimg1,
                      gsr_stream_t stream);

/* Fused row gather over a structure of arrays (extension): the data movement of the reference's
 * prune_points / cat_tensors_to_optimizer / densification_postfix (scene/gaussian_model.py:421-512) for ALL parameter
 * tensors, both Adam moments of each and the statistics in one launch.  For every array a < n_arrays (<= 32; src / dst /
 * row_floats / zero_new are HOST arrays, the pointers in them device pointers):
 *   dst[a][j][:] = src[a][index[j] & 0x3FFFFFFF][:]          for j < n_out, rows of row_floats[a] floats,
 *   dst[a][j][:] = 0   if index[j] < 0, or if bit 30 of index[j] is set ("new Gaussian") and zero_new[a] != 0
 * (the Adam moments of cloned / split Gaussians start at zero, their parameters are copies of the parent's row).
 * index is a device array; dst must not alias src. */
int gsr_gather_rows(int n_arrays, const float *const *src, float *const *dst, const int *row_floats, const int *zero_new,
                    int n_out, const int *index, gsr_stream_t stream);

/* k-NN service replacing the un-vendored KNN_CUDA dependency (scene/gaussian_model.py:87-89).
 * Semantics assumed for KNN_CUDA 0.2 (parity unpinned): exact brute-force k-NN, Euclidean distances in ascending order;
 * here ties resolve to the lowest index.
 *   gsr_knn_self: the k <= 3 nearest of every point among the SAME P points, the point itself included (what
 *     knn(xyz, xyz) returns; :176,573,621,671): idx[P][k] int32, dist[P][k].  Workspace: gsr_dist2_workspace_bytes(P).
 *   gsr_knn_nearest: nearest of N reference points for each of M queries (:727, distance of the Gaussians to the SMPL
 *     vertices; :775 is fused into gsr_lbs_forward_grid): idx[M] and/or dist[M] (either may be null).
 *     Workspace: gsr_lbs_workspace_bytes(N), 16-byte aligned. */
int gsr_knn_self(int P, const float *points, int k, int *idx, float *dist, char *workspace, size_t workspace_bytes,
                 gsr_stream_t stream);
int gsr_knn_nearest(int M, const float *query, int N, const float *ref, int *idx, float *dist, char *workspace,
                    size_t workspace_bytes, gsr_stream_t stream);

/* Compact exchange of the SH-coefficient gradient between view-parallel ranks (extension).  For one view
 * dL_dsh[i][k][c] = w_k(dir_i) * dL_dRGB[i][c] (dir = normalise(mean - campos); dL_dRGB zeroed on clamped channels,
 * CR/backward.cu:40-116): ranks all-gather 12 B per Gaussian instead of all-reducing 12 M B.
 *   gsr_sh_view_pack: packed[P][3] = dL_dcolor (the backward's dL_dcolor output of an SH-mode call) with the channels the
 *     forward clamped set to zero; geom_buffer is that call's geometry buffer.
 *   gsr_sh_grad_from_views: dL_dsh[P][M][3] = scale * sum_v w_k(normalise(means3D - campos_v)) * packed_v, v in fixed
 *     order; `views` holds n_views blocks of view_stride floats: [P*3 packed | campos xyz | padding].  dev_scale (device
 *     pointer to one float, or null) multiplies `scale`: the view-parallel step passes 0 there for a step some rank could not
 *     render (binning overflow), so that every replica skips it, without a host read. */
int gsr_sh_view_pack(int P, const char *geom_buffer, const float *dL_dcolor, float *packed, gsr_stream_t stream);
int gsr_sh_grad_from_views(int P, int sh_degree, int M, int n_views, const float *means3D, const float *views,
                           size_t view_stride, float scale, const float *dev_scale, float *dL_dsh, gsr_stream_t stream);

/* The same exchange for ARTICULATED Gaussians (render(): colours come from the per-frame attribute kernel, and every view poses
 * the Gaussians differently, so the positions travel with the view).  A view block is [P*3 masked dL_dRGB | ... ] with the
 * posed positions at means_offset (>= 3 P, floats) and the camera position at cam_offset (>= means_offset + 3 P):
 *   gsr_sh_view_pack_posed: fills the three parts of this rank's block in one launch; `colors` = the forward's colours
 *     max(SH + 0.5, 0) (a channel that came out 0 was clamped: its gradient is dropped, gaussian_renderer/__init__.py:195),
 *     dL_dcolors = the rasterizer's gradient with respect to them, means3D_view = the view's posed positions, campos device [3].
 *   gsr_sh_grad_from_views_posed: the mean SH gradient written in the model's two parameter layouts, dL_dsh_dc [P][1][3] and
 *     dL_dsh_rest [P][15][3] (16-byte aligned); scale / dev_scale as above. */
int gsr_sh_view_pack_posed(int P, const float *colors, const float *dL_dcolors, const float *means3D_view, const float *campos,
                           float *view_block, size_t means_offset, size_t cam_offset, gsr_stream_t stream);
int gsr_sh_grad_from_views_posed(int P, int sh_degree, int n_views, const float *views, size_t view_stride, size_t means_offset,
                                 size_t cam_offset, float scale, const float *dev_scale, float *dL_dsh_dc, float *dL_dsh_rest,
                                 gsr_stream_t stream);

/* Bookkeeping of one view-parallel step around its gradient all-reduce, one single-thread launch, no host read
 * (extension).  status = dev_status of the step's gsr_rasterize_forward_async (R, overflow flag); overflow_slot = one float
 * inside the all-reduced gradient bucket.
 *   phase 0 (before the reduction): overflow_slot[0] = overflow flag as 0/1
 *   phase 1 (after the SUM reduction): scale[0] = overflow_slot[0] > 0 ? 0 : inv_world  (a step some rank could not render
 *            is skipped by every replica); report[0..2] = {ranks that overflowed, R, own flag} -- `report` may be pinned,
 *            device-mapped host memory, examined by the host after an event
 *   phase 2: both (single process). */
int gsr_step_status(int phase, const uint32_t *status, float *overflow_slot, float inv_world, float *scale, uint32_t *report,
                    gsr_stream_t stream);
/* gsr_step_status phase 1 AND the division of the reduced bucket in ONE launch: bucket[0 .. n_floats) is the flat fp32 buffer
 * that was SUM all-reduced (16-byte aligned), bucket[overflow_index] the number of ranks whose binning overflowed; every other
 * element is multiplied by scale = (that count > 0 ? 0 : inv_world); scale[0] (optional) and report[0..2] (optional; status may
 * be null: words 1, 2 are then 0) as in phase 1. */
int gsr_step_finish(const uint32_t *status, float *bucket, size_t n_floats, size_t overflow_index, float inv_world, float *scale,
                    uint32_t *report, gsr_stream_t stream);

/* SMPL pose -> joint transforms (batch size 1): rodrigues of the 24 axis-angle vectors (angle = |theta + 1e-8|), the
 * optional pose-refinement product R_j <- R_j correct_Rs[j-1] (j >= 1), the kinematic chain and the removal of the rest
 * pose -- scene/gaussian_model.py:894-980 (batch_rodrigues_torch, get_rigid_transformation_torch,
 * get_transform_params_torch) and :822-825 -- in one single-wave launch.
 *   poses[72], correct_Rs[23][9] or null, joints[24][3] (device); parents_host[24]: HOST array, parents[i] < i, entry 0
 *   ignored.  Outputs (device): rot_mats[24][9] (may be null), A[24][16] row-major 4x4. */
int gsr_smpl_pose_forward(const float *poses, const float *correct_Rs, const float *joints, const int *parents_host,
                          float *rot_mats, float *A, gsr_stream_t stream);

/* Adjoint of gsr_smpl_pose_forward: dL_dA[24][16] (row 3 ignored), dL_drot_mats[24][9] or null (the pose blend shapes'
 * use of rot_mats) -> dL_dposes[72], dL_dcorrect_Rs[23][9] (null if correct_Rs is null), dL_djoints[24][3]; each output
 * may be null and is fully written otherwise. */
int gsr_smpl_pose_backward(const float *poses, const float *correct_Rs, const float *joints, const int *parents_host,
                           const float *dL_dA, const float *dL_drot_mats, float *dL_dposes, float *dL_dcorrect_Rs,
                           float *dL_djoints, gsr_stream_t stream);

/* Row-major matrix-vector products for the SMPL pose blend shapes (scene/gaussian_model.py:805-811,827-839):
 *   gsr_gemv_rows:   out[r] = sum_k mat[r][k] * vec[k]          (offsets[V*3] = posedirs[V*3][207] . pose_feature[207])
 *   gsr_gemv_rows_t: dvec[k] = sum_r dout[r] * mat[r][k]        (its backward w.r.t. the pose feature; dvec is overwritten)
 * cols <= 256. */
int gsr_gemv_rows(int rows, int cols, const float *mat, const float *vec, float *out, gsr_stream_t stream);
int gsr_gemv_rows_t(int rows, int cols, const float *mat, const float *dout, float *dvec, gsr_stream_t stream);

/* The parameter activations render() applies every frame (the reference's property getters, scene/gaussian_model.py:157-199,
 * and the occlusion placeholder of gaussian_renderer/__init__.py:141), P rows each, one kernel:
 *   opacity[1] = sigmoid(opacity_raw)   albedo[3] = sigmoid(albedo_raw)   scaling[3] = exp(scaling_raw)
 *   rotation[4] = rotation_raw / max(|rotation_raw|, 1e-12)   normal[3] = normal_raw / |normal_raw|   occlusion[3] = opacity x3
 * (get_roughness reads _albedo as well in the reference, :197-199: callers alias the albedo output for it). */
int gsr_model_activations_forward(int P, const float *opacity_raw, const float *albedo_raw, const float *scaling_raw,
                                  const float *rotation_raw, const float *normal_raw, float *opacity, float *albedo,
                                  float *scaling, float *rotation, float *normal, float *occlusion, gsr_stream_t stream);

/* Backward of gsr_model_activations_forward: opacity / albedo / scaling are the forward's outputs, g_* the gradients of the six
 * outputs (a null g_* = that output did not reach the loss), d_*_raw are fully written. */
int gsr_model_activations_backward(int P, const float *rotation_raw, const float *normal_raw, const float *opacity,
                                   const float *albedo, const float *scaling, const float *g_opacity, const float *g_albedo,
                                   const float *g_scaling, const float *g_rotation, const float *g_normal,
                                   const float *g_occlusion, float *d_opacity_raw, float *d_albedo_raw, float *d_scaling_raw,
                                   float *d_rotation_raw, float *d_normal_raw, gsr_stream_t stream);
/* The same with acc_drotation_raw [P][4] (or null) ADDED to d_rotation_raw: a gradient of the raw quaternion that already exists
 * (render() hands the raw quaternion to the covariance as well, gaussian_renderer/__init__.py:128-131 / scene/gaussian_model.py:35-42:
 * the leaf would otherwise receive two gradients and autograd an add kernel per frame). */
int gsr_model_activations_backward_acc(int P, const float *rotation_raw, const float *normal_raw, const float *opacity,
                                       const float *albedo, const float *scaling, const float *g_opacity, const float *g_albedo,
                                       const float *g_scaling, const float *g_rotation, const float *g_normal,
                                       const float *g_occlusion, float *d_opacity_raw, float *d_albedo_raw, float *d_scaling_raw,
                                       float *d_rotation_raw, float *d_normal_raw, const float *acc_drotation_raw,
                                       gsr_stream_t stream);

/* Per-frame, per-Gaussian attributes render() derives between the LBS deform and the rasterizer
 * (gaussian_renderer/__init__.py:128-198; scene/gaussian_model.py:35-42,186-190; utils/general_utils.py:64-157;
 * utils/sh_utils.py:57-117; transform.py:9-17) -- one kernel instead of the reference's torch op chain.
 * Inputs, P rows each: means3D[3] (world), transforms[9] (row-major LBS rotation), world_normals[3] (LBS-rotated canonical
 *   normal, any length), scales[3] (activated), rot_cov[4] (raw quaternion of get_covariance, normalised inside),
 *   rot_axis[4] (quaternion of get_minimum_axis, normalised inside), albedo[3], roughness[3], occlusion[3],
 *   shs[M][3] or null (null: no colours), campos[3], viewmatrix[16] (row-vector convention, as the rasterizer's).
 * Outputs: cov3D[P][6] (upper triangle of T R S S R^T T^T, S = scale_modifier*scales), colors[P][3] =
 *   max(SH_sh_degree(dir) + 0.5, 0) (needs shs), features[P][18] = normal | world_normal | albedo | occlusion |
 *   roughness mean x3 | minimum axis, i.e. the `extra_features` array of gsr_rasterize_forward_ex.
 * Equal scales: the minimum-axis order is the stable one (lowest index first); torch.argsort leaves it unspecified. */
int gsr_frame_attributes_forward(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                 const float *world_normals, const float *scales, float scale_modifier,
                                 const float *rot_cov, const float *rot_axis, const float *albedo,
                                 const float *roughness, const float *occlusion, const float *shs,
                                 const float *campos, const float *viewmatrix, float *cov3D, float *colors,
                                 float *features, gsr_stream_t stream);

/* Backward of gsr_frame_attributes_forward (what autograd derives for the reference's op chain).  Incoming gradients
 * dL_dcov3D[P][6], dL_dcolors[P][3], dL_dfeatures[P][18] may each be null (= zero).  Every outgoing array is fully
 * written (no accumulation): dL_dmeans3D[P][3] (view-direction dependence of the colours), dL_dtransforms[P][9],
 * dL_dworld_normals[P][3], dL_dscales[P][3], dL_drot_cov[P][4], dL_drot_axis[P][4], dL_dalbedo / dL_droughness /
 * dL_docclusion [P][3], dL_dshs[P][M][3] (needs shs; may be null for M = 16 when the caller does not want the SH gradient -- the
 * view-parallel compact exchange rebuilds it from dL_dcolors, gsr_sh_grad_from_views_posed). */
int gsr_frame_attributes_backward(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                  const float *world_normals, const float *scales, float scale_modifier,
                                  const float *rot_cov, const float *rot_axis, const float *albedo,
                                  const float *roughness, const float *occlusion, const float *shs,
                                  const float *campos, const float *viewmatrix, const float *dL_dcov3D,
                                  const float *dL_dcolors, const float *dL_dfeatures, float *dL_dmeans3D,
                                  float *dL_dtransforms, float *dL_dworld_normals, float *dL_dscales,
                                  float *dL_drot_cov, float *dL_drot_axis, float *dL_dalbedo, float *dL_droughness,
                                  float *dL_docclusion, float *dL_dshs, gsr_stream_t stream);

/* The same two operations reading the SH coefficients from the model's TWO parameter tensors in place -- shs = the DC
 * coefficient [P][1][3], shs_rest = the other 15 [P][15][3] (16-byte aligned), M = 16 -- instead of through the torch.cat of
 * get_features (scene/gaussian_model.py:167-171; 38 MB copied per frame at 200k Gaussians, and split again in backward); the
 * backward writes dL_dshs [P][1][3] and dL_dshs_rest [P][15][3].  shs_rest = null: exactly the functions above. */
int gsr_frame_attributes_forward_split(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                       const float *world_normals, const float *scales, float scale_modifier,
                                       const float *rot_cov, const float *rot_axis, const float *albedo, const float *roughness,
                                       const float *occlusion, const float *shs, const float *shs_rest, const float *campos,
                                       const float *viewmatrix, float *cov3D, float *colors, float *features, gsr_stream_t stream);
int gsr_frame_attributes_backward_split(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                        const float *world_normals, const float *scales, float scale_modifier,
                                        const float *rot_cov, const float *rot_axis, const float *albedo, const float *roughness,
                                        const float *occlusion, const float *shs, const float *shs_rest, const float *campos,
                                        const float *viewmatrix, const float *dL_dcov3D, const float *dL_dcolors,
                                        const float *dL_dfeatures, float *dL_dmeans3D, float *dL_dtransforms,
                                        float *dL_dworld_normals, float *dL_dscales, float *dL_drot_cov, float *dL_drot_axis,
                                        float *dL_dalbedo, float *dL_droughness, float *dL_docclusion, float *dL_dshs,
                                        float *dL_dshs_rest, gsr_stream_t stream);
/* gsr_frame_attributes_backward_split with two accumulations that save autograd an add kernel each in render()'s frame:
 * acc_dmeans3D [P][3] (or null) -- the rasterizer's dL_dmeans3D of the same frame -- is ADDED to dL_dmeans3D; and
 * dL_droughness == dL_dalbedo (the same pointer: albedo and roughness are one tensor, scene/gaussian_model.py:197-199) makes the
 * kernel write their SUM there.  (Both also hold for the function above: it passes acc_dmeans3D = null.) */
int gsr_frame_attributes_backward_acc(int P, int sh_degree, int M, const float *means3D, const float *transforms,
                                      const float *world_normals, const float *scales, float scale_modifier,
                                      const float *rot_cov, const float *rot_axis, const float *albedo, const float *roughness,
                                      const float *occlusion, const float *shs, const float *shs_rest, const float *campos,
                                      const float *viewmatrix, const float *dL_dcov3D, const float *dL_dcolors,
                                      const float *dL_dfeatures, float *dL_dmeans3D, float *dL_dtransforms,
                                      float *dL_dworld_normals, float *dL_dscales, float *dL_drot_cov, float *dL_drot_axis,
                                      float *dL_dalbedo, float *dL_droughness, float *dL_docclusion, float *dL_dshs,
                                      float *dL_dshs_rest, const float *acc_dmeans3D, gsr_stream_t stream);

/* The per-Gaussian skinning-weight offset network of render() (nets/mlp_delta_weight_lbs.py:5-32: a 63-d positional embedding of
 * the canonical position through 63-128-128-128-(63+128)-128-24 with ReLU; gaussian_renderer/__init__.py:100-106 runs it every frame
 * when motion_offset_flag is set) on the matrix cores (csrc/mlp.hip: f32 MFMA, activations in registers): forward = one kernel.
 *   gsr_lbs_offset_mlp_pack: weights[5] / biases[5] = the module's tensors in its own layout -- bw_linears.0..3 ([128][63],
 *     [128][128], [128][128], [128][191]: Conv1d weight [out][in][1]) and bw_fc ([24][128]), biases [128] x 4 and [24] -- host arrays
 *     of DEVICE pointers; packed: gsr_lbs_offset_mlp_packed_floats() floats, 16-byte aligned (re-pack after every parameter update);
 *   gsr_lbs_offset_mlp_forward: xyz [P][3] -> out [P][24] (the module returns [1][24][P]: the transposed view of this). */
size_t gsr_lbs_offset_mlp_packed_floats(void);
int gsr_lbs_offset_mlp_pack(const float *const *weights, const float *const *biases, float *packed, gsr_stream_t stream);
int gsr_lbs_offset_mlp_forward(int P, const float *xyz, const float *packed, float *out, gsr_stream_t stream);
/* Which matrix instruction the three kernels run on (process-wide): 1 (default) = v_mfma_f32_32x32x16_bf16 with BOTH
 * operands split in two bf16 terms, three products per step (error 2^-16 of a product: 7e-6 of the output's scale against float64,
 * 16 x the rate of the f32 instruction); 0 = v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, 6e-7). */
int gsr_lbs_offset_mlp_set_precision(int mode);
/* The forward on the bf16 instruction whatever the mode (comparisons: tools/mlp_bench.py). */
int gsr_debug_lbs_offset_mlp_forward_bf16x3(int P, const float *xyz, const float *packed, float *out, gsr_stream_t stream);
/* Backward w.r.t. the parameters (the positions arrive detached, gaussian_renderer/__init__.py:104: no input gradient).  The forward
 * is run again inside (no activation was kept), dL_dout is [P][24]; the gradients are ADDED into dL_dweights[5] / dL_dbiases[5] (host
 * arrays of device pointers, the module's own layouts: zero them first); workspace: gsr_lbs_offset_mlp_backward_workspace_floats(P)
 * floats, 16-byte aligned (every layer's activations and pre-activation gradients, feature-major, for the weight-gradient products). */
size_t gsr_lbs_offset_mlp_backward_workspace_floats(int P);
int gsr_lbs_offset_mlp_backward(int P, const float *xyz, const float *packed, const float *dL_dout, float *workspace,
                                float *const *dL_dweights, float *const *dL_dbiases, gsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_INCLUDED */
