// driver.cpp -- exercises the HOST side of libgsr's C ABI (include/gsr.h) against hip_stub.cpp under sanitizers.
// No kernel runs: what is checked is everything the library does on the host around its launches -- argument validation and
// error strings, arena carving (every scratch array inside the byte count the library asked for), per-stream option map,
// stage-profile store, deterministic-mode allocation, workspace arithmetic, launch geometry (the stub rejects empty / oversized
// grids) -- single-threaded over ragged / tiny / large shapes and from several threads at once (the autograd backward thread
// and the main thread share this state in the product).  Exit code 0 = clean; sanitizer reports abort the run.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../mygauhuman_amd/csrc/gsr_common.h"

extern "C" void hipstub_set_fake_readback(uint32_t R);
extern "C" long hipstub_launches(void);
extern "C" long hipstub_bad_launches(void);

static std::atomic<int> g_fail{0};
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) {                                                            \
      fprintf(stderr, "FAIL %s:%d: %s  (last error: %s)\n", __FILE__, __LINE__, #cond, gsr_last_error()); \
      g_fail++;                                                               \
    }                                                                         \
  } while (0)

struct Arena {  // one growable scratch buffer behind a gsr_alloc_fn callback, with the exact size the library asked for
  std::vector<char *> blocks;
  size_t last = 0;
  static char *cb(void *user, size_t bytes) {
    Arena *a = static_cast<Arena *>(user);
    char *p = static_cast<char *>(calloc(1, bytes ? bytes : 1));  // exact size: ASan sees any host access beyond it
    a->blocks.push_back(p);
    a->last = bytes;
    return p;
  }
  char *ptr() const { return blocks.empty() ? nullptr : blocks.back(); }
  ~Arena() {
    for (char *p : blocks) free(p);
  }
};

template <typename T>
struct Dev {  // "device" array of exactly n elements
  T *p;
  explicit Dev(size_t n) : p(static_cast<T *>(calloc(n ? n : 1, sizeof(T)))) {}
  ~Dev() { free(p); }
  operator T *() const { return p; }
};

// the carve functions must stay inside the byte counts the library reports, without overlap, at 256-byte alignment
static void check_carving() {
  using namespace gsr;
  const size_t Ps[] = {1, 2, 255, 256, 257, 1000, 200000, 500000, 3000000};
  for (size_t P : Ps) {
    char *base = reinterpret_cast<char *>(uintptr_t(1) << 20);
    GeomState g = geom_from_chunk(base, P);
    const size_t bytes = gsr_geometry_bytes((int)P);
    const char *end = base + bytes;
    const void *starts[] = {g.recs, g.cov3D, g.clamped, g.tiles_touched, g.point_offsets, g.internal_radii, g.block_incl,
                            g.block_sums, g.block_prefix, g.total, g.grad_rows};
    const size_t sizes[] = {P * sizeof(SplatRec), P * 24, P, P * 4, P * 4, P * 4, P * 4, (size_t)pre_blocks((int)P) * 4,
                            (size_t)pre_blocks((int)P) * 4, 16, P * GROWX * 4};
    for (int i = 0; i < 11; i++) {
      const char *s = static_cast<const char *>(starts[i]);
      EXPECT(s >= base && s + sizes[i] <= end && (reinterpret_cast<uintptr_t>(s) & 255) == 0);
      if (i) EXPECT(static_cast<const char *>(starts[i - 1]) + sizes[i - 1] <= s);
    }
  }
  // order entries and list segments (gsr_common.h "list segments"): the packing round-trips, never looks like the padding entry, and
  // the segment arithmetic the forward (checkpoints) and the backward (walk bounds) share covers a list exactly once
  for (uint32_t tile : {0u, 1u, 4095u, 8191u, ORDER_TILE_MASK})
    for (uint32_t nseg = 1; nseg <= (uint32_t)SEG_MAX; nseg++)
      for (uint32_t seg = 0; seg < nseg; seg++) {
        const uint32_t e = order_entry(tile, seg, nseg);
        EXPECT(e != ORDER_NO_TILE && order_entry_tile(e) == tile && order_entry_seg(e) == seg && order_entry_nseg(e) == nseg);
      }
  for (int n : {1, 63, 64, 65, 384, 385, 511, 512, 704, 896, 897, 1000, 1398, 2047, 2048, 4097, 100000})
    for (int nseg = 1; nseg <= SEG_MAX; nseg++) {
      const int len = segment_len(n, nseg);
      EXPECT(len % 64 == 0 && len >= (n + nseg - 1) / nseg && (long)len * nseg >= n);
      long covered = 0;
      for (int sgm = 0; sgm < nseg; sgm++) {
        const int lo = sgm * len, hi = lo + len < n ? lo + len : n;
        if (lo < n) covered += hi - lo;   // (a segment that starts beyond the list is empty: the kernels return at once)
      }
      EXPECT(covered == n);
    }
  const int dims[][2] = {{1, 1}, {16, 16}, {17, 33}, {160, 96}, {512, 512}, {1024, 1024}, {1920, 1080}, {4096, 4096},
                         {16, 400}, {16, 16000}, {16000, 16}, {48, 7000}, {33, 33}};
  const size_t Rs[] = {0, 1, 4095, 4096, 4097, 1326873, 5000000};
  for (auto &d : dims)
    for (size_t R : Rs) {
      const size_t tiles = (size_t)((d[0] + 15) / 16) * ((d[1] + 15) / 16), npix = (size_t)d[0] * d[1];
      char *base = reinterpret_cast<char *>(uintptr_t(1) << 20);
      BinningState b = binning_from_chunk(base, R, tiles);
      const size_t n = R ? R : 1;
      const char *end = base + gsr_binning_bytes(R, d[0], d[1]);
      EXPECT(reinterpret_cast<char *>(b.keys_a + n) <= reinterpret_cast<char *>(b.vals_a));
      EXPECT(reinterpret_cast<char *>(b.vals_a + n) <= reinterpret_cast<char *>(b.keys_s));
      EXPECT(reinterpret_cast<char *>(b.keys_s + n) <= reinterpret_cast<char *>(b.vals_s));
      EXPECT(reinterpret_cast<char *>(b.vals_s + n) <= reinterpret_cast<char *>(b.hist));
      EXPECT(reinterpret_cast<char *>(b.hist + sort_hist_words(n)) <= reinterpret_cast<char *>(b.tile_counts));
      EXPECT(reinterpret_cast<char *>(b.tile_counts + tiles * 16) <= reinterpret_cast<char *>(b.tile_cursor));
      EXPECT(reinterpret_cast<char *>(b.tile_cursor + tiles) <= end);
      // the backward carves the same buffer WITHOUT the tile count: the arrays it reads must sit at the same addresses
      BinningState b2 = binning_from_chunk(base, R);
      EXPECT(b2.vals_s == b.vals_s && b2.keys_s == b.keys_s);
      ImageState s = image_from_chunk(base, npix, tiles);
      const char *iend = base + gsr_image_bytes(d[0], d[1]);
      EXPECT(reinterpret_cast<char *>(s.final_T + npix) <= reinterpret_cast<char *>(s.n_contrib));
      EXPECT(reinterpret_cast<char *>(s.n_contrib + npix) <= reinterpret_cast<char *>(s.ranges));
      EXPECT(reinterpret_cast<char *>(s.ranges + tiles) <= reinterpret_cast<char *>(s.order));
      EXPECT(reinterpret_cast<char *>(s.order + order_words(tiles)) <= reinterpret_cast<char *>(s.ckpt_base));
      EXPECT(reinterpret_cast<char *>(s.ckpt_base + tiles) <= reinterpret_cast<char *>(s.ckpt));
      EXPECT(reinterpret_cast<char *>(s.ckpt + ckpt_records(tiles) * (size_t)CKPT_PLANES * 256) <= iend);
      // every visiting-order mode fits the order array of this grid (header + slots), also with every extra segment slot in use
      for (int mode = 0; mode <= 3; mode++)
        EXPECT(ORDER_HDR + (size_t)tile_slots((d[0] + 15) / 16, (d[1] + 15) / 16, mode) <= order_words(tiles));
      EXPECT(tile_slots_max((d[0] + 15) / 16, (d[1] + 15) / 16) >= tiles + seg_extra_max((uint32_t)tiles));
    }
}

struct Scene {
  int P, W, H, M;
  Dev<float> bg, means, shs, colors, opac, scales, rots, cov, view, proj, campos, out_color, out_depth, out_alpha, extra, out_extra;
  Dev<int> radii;
  Dev<float> dpix, ddepth, dalpha, g_mean2D, g_conic, g_opac, g_color, g_mean3D, g_cov, g_sh, g_scale, g_rot, g_extra;
  Scene(int P_, int W_, int H_, int M_)
      : P(P_), W(W_), H(H_), M(M_), bg(3), means((size_t)P_ * 3), shs((size_t)P_ * M_ * 3), colors((size_t)P_ * 3), opac(P_),
        scales((size_t)P_ * 3), rots((size_t)P_ * 4), cov((size_t)P_ * 6), view(16), proj(16), campos(3),
        out_color((size_t)3 * W_ * H_), out_depth((size_t)W_ * H_), out_alpha((size_t)W_ * H_), extra((size_t)P_ * 18),
        out_extra((size_t)18 * W_ * H_), radii(P_), dpix((size_t)3 * W_ * H_), ddepth((size_t)W_ * H_), dalpha((size_t)W_ * H_),
        g_mean2D((size_t)P_ * 3), g_conic((size_t)P_ * 4), g_opac(P_), g_color((size_t)P_ * 3), g_mean3D((size_t)P_ * 3),
        g_cov((size_t)P_ * 6), g_sh((size_t)P_ * M_ * 3), g_scale((size_t)P_ * 3), g_rot((size_t)P_ * 4), g_extra((size_t)P_ * 18) {}
};

static void forward_backward(Scene &s, bool sh_mode, bool with_extra, uint32_t fake_R, int debug, gsr_stream_t stream,
                             bool shared_fake = false) {
  Arena ga, ba, ia;
  int R = -1;
  if (!shared_fake) hipstub_set_fake_readback(fake_R);
  int rc = gsr_rasterize_forward_ex(Arena::cb, &ga, Arena::cb, &ba, Arena::cb, &ia, s.P, sh_mode ? 3 : 0, sh_mode ? s.M : 0, s.bg,
                                    s.W, s.H, s.means, sh_mode ? s.shs.p : nullptr, sh_mode ? nullptr : s.colors.p, s.opac,
                                    sh_mode ? s.scales.p : nullptr, 1.0f, sh_mode ? s.rots.p : nullptr, sh_mode ? nullptr : s.cov.p,
                                    s.view, s.proj, s.campos, 0.5f, 0.5f, 0, s.out_color, s.out_depth, s.out_alpha, s.radii, debug, &R,
                                    with_extra ? s.extra.p : nullptr, with_extra ? 18 : 0, with_extra ? s.out_extra.p : nullptr,
                                    GSR_SH_F32, stream);
  EXPECT(rc == GSR_OK);
  if (s.P == 0) {
    EXPECT(R == 0);
    return;
  }
  EXPECT((uint32_t)R == fake_R);
  EXPECT(ga.last == gsr_geometry_bytes(s.P) && ia.last == gsr_image_bytes(s.W, s.H) && ba.last == gsr_binning_bytes((size_t)R, s.W, s.H));
  const float *tri[6] = {s.dpix, nullptr, s.dpix, nullptr, nullptr, s.dpix};
  rc = gsr_rasterize_backward_ex(s.P, sh_mode ? 3 : 0, sh_mode ? s.M : 0, R, s.bg, s.W, s.H, s.means, sh_mode ? s.shs.p : nullptr,
                                 sh_mode ? nullptr : s.colors.p, s.out_alpha, sh_mode ? s.scales.p : nullptr, 1.0f,
                                 sh_mode ? s.rots.p : nullptr, sh_mode ? nullptr : s.cov.p, s.view, s.proj, s.campos, 0.5f, 0.5f,
                                 s.radii, ga.ptr(), ba.ptr(), ia.ptr(), s.dpix, s.ddepth, s.dalpha, s.g_mean2D, s.g_conic, s.g_opac,
                                 s.g_color, s.g_mean3D, s.g_cov, sh_mode ? s.g_sh.p : nullptr, sh_mode ? s.g_scale.p : nullptr,
                                 sh_mode ? s.g_rot.p : nullptr, debug, with_extra ? s.extra.p : nullptr, with_extra ? 18 : 0,
                                 with_extra ? tri : nullptr, with_extra ? s.g_extra.p : nullptr, GSR_SH_F32, stream);
  EXPECT(rc == GSR_OK);
  // every introspection selector copies within the buffers
  const size_t tiles = (size_t)((s.W + 15) / 16) * ((s.H + 15) / 16);
  size_t big = (size_t)s.P * 24;
  if ((size_t)R * 8 > big) big = (size_t)R * 8;
  if ((size_t)s.W * s.H * 4 > big) big = (size_t)s.W * s.H * 4;
  if (tiles * 8 > big) big = tiles * 8;
  Dev<char> dst(big);
  for (int q = 0; q <= 12; q++) EXPECT(gsr_query_state(q, s.P, R, s.W, s.H, ga.ptr(), ba.ptr(), ia.ptr(), dst, stream) == GSR_OK);
  EXPECT(gsr_query_state(99, s.P, R, s.W, s.H, ga.ptr(), ba.ptr(), ia.ptr(), dst, stream) == GSR_EINVAL);
}

static void async_forward_backward(Scene &s, size_t capacity, gsr_stream_t stream, bool fused_loss) {
  Dev<char> geom(gsr_geometry_bytes(s.P)), bin(gsr_binning_bytes(capacity, s.W, s.H)), img(gsr_image_bytes(s.W, s.H));
  Dev<uint32_t> status(2);
  int rc = gsr_rasterize_forward_async(geom, bin, capacity, img, s.P, 3, s.M, s.bg, s.W, s.H, s.means, s.shs, nullptr, s.opac, s.scales,
                                       1.0f, s.rots, nullptr, s.view, s.proj, s.campos, 0.5f, 0.5f, 0, s.out_color, s.out_depth,
                                       s.out_alpha, s.radii, 0, status, stream);
  EXPECT(rc == GSR_OK);
  if (fused_loss)
    rc = gsr_rasterize_backward_alpha_mask_loss(s.P, 3, s.M, (int)capacity, s.bg, s.W, s.H, s.means, s.shs, nullptr, s.out_alpha,
                                                s.scales, 1.0f, s.rots, nullptr, s.view, s.proj, s.campos, 0.5f, 0.5f, s.radii, geom,
                                                bin, img, s.out_color, s.dpix, s.dalpha, 0.1f, s.g_mean2D, s.g_conic, s.g_opac,
                                                s.g_color, s.g_mean3D, s.g_cov, s.g_sh, s.g_scale, s.g_rot, GSR_BWD_ROWS_ZEROED,
                                                GSR_SH_F32, stream);
  else
    rc = gsr_rasterize_backward(s.P, 3, s.M, (int)capacity, s.bg, s.W, s.H, s.means, s.shs, nullptr, s.out_alpha, s.scales, 1.0f,
                                s.rots, nullptr, s.view, s.proj, s.campos, 0.5f, 0.5f, s.radii, geom, bin, img, s.dpix, s.ddepth,
                                s.dalpha, s.g_mean2D, s.g_conic, s.g_opac, s.g_color, s.g_mean3D, s.g_cov, s.g_sh, s.g_scale, s.g_rot,
                                0, stream);
  EXPECT(rc == GSR_OK);
}

static void error_paths() {
  Scene s(10, 32, 32, 16);
  Arena ga, ba, ia;
  int R = 0;
  // neither SHs nor colours (CR/rasterizer_impl.cu:244-247)
  EXPECT(gsr_rasterize_forward(Arena::cb, &ga, Arena::cb, &ba, Arena::cb, &ia, s.P, 0, 0, s.bg, s.W, s.H, s.means, nullptr, nullptr,
                               s.opac, s.scales, 1.0f, s.rots, nullptr, s.view, s.proj, s.campos, 0.5f, 0.5f, 0, s.out_color,
                               s.out_depth, s.out_alpha, s.radii, 0, &R, nullptr) == GSR_EINVAL);
  EXPECT(strstr(gsr_last_error(), "SHs or precomputed") != nullptr);
  EXPECT(gsr_rasterize_forward(nullptr, nullptr, Arena::cb, &ba, Arena::cb, &ia, s.P, 0, 0, s.bg, s.W, s.H, s.means, nullptr, s.colors,
                               s.opac, s.scales, 1.0f, s.rots, nullptr, s.view, s.proj, s.campos, 0.5f, 0.5f, 0, s.out_color,
                               s.out_depth, s.out_alpha, s.radii, 0, &R, nullptr) == GSR_EINVAL);
  EXPECT(gsr_set_tuning("no_such_key", 1) == GSR_EINVAL && strstr(gsr_last_error(), "unknown tuning key") != nullptr);
  EXPECT(gsr_set_tuning("blend_bwd_waves", 3) == GSR_EINVAL);
  EXPECT(gsr_set_tuning(nullptr, 3) == GSR_EINVAL);
  double ms;
  long n;
  EXPECT(gsr_profile_read(77, &ms, &n) == GSR_EINVAL);
  Dev<uint64_t> k(100), ko(100);
  Dev<uint32_t> v(100), vo(100);
  Dev<char> ws(gsr_sort_workspace_bytes(100));
  EXPECT(gsr_sort_pairs_u64(100, k, ko, v, vo, 45, ws, gsr_sort_workspace_bytes(100), nullptr) == GSR_OK);
  EXPECT(gsr_sort_pairs_u64(100, k, ko, v, vo, 45, ws, 16, nullptr) == GSR_ENOMEM);
  EXPECT(gsr_sort_pairs_u64(100, k, ko, v, vo, 65, ws, gsr_sort_workspace_bytes(100), nullptr) == GSR_EINVAL);
  EXPECT(gsr_sort_pairs_u32(100, v, vo, v, vo, 30, ws, gsr_sort_workspace_bytes(100), nullptr) == GSR_OK);
  Dev<float> pts(300 * 3), d2(300);
  Dev<char> kws(gsr_dist2_workspace_bytes(300));
  EXPECT(gsr_dist2(300, pts, d2, kws, gsr_dist2_workspace_bytes(300), nullptr) == GSR_OK);
  EXPECT(gsr_dist2(300, pts, d2, kws, 8, nullptr) != GSR_OK);
  Dev<int> idx(300 * 3);
  Dev<float> dist(300 * 3);
  EXPECT(gsr_knn_self(300, pts, 3, idx, dist, kws, gsr_dist2_workspace_bytes(300), nullptr) == GSR_OK);
  EXPECT(gsr_knn_self(300, pts, 4, idx, dist, kws, gsr_dist2_workspace_bytes(300), nullptr) != GSR_OK);
  Dev<char> lws(gsr_lbs_workspace_bytes(6890) + 16);
  Dev<float> verts(6890 * 3);
  EXPECT(gsr_lbs_grid_build(6890, verts, lws, gsr_lbs_workspace_bytes(6890), nullptr) == GSR_OK);
  EXPECT(gsr_knn_nearest(300, pts, 6890, verts, idx, dist, lws, gsr_lbs_workspace_bytes(6890), nullptr) == GSR_OK);
  Dev<float> img1(3 * 50 * 37), img2(3 * 50 * 37), map(3 * 50 * 37), dA(3 * 50 * 37), dB(3 * 50 * 37), dC(3 * 50 * 37);
  EXPECT(gsr_ssim_forward(3, 50, 37, img1, img2, map, dA, dB, dC, nullptr) == GSR_OK);
  EXPECT(gsr_ssim_forward(3, 50, 37, img1, img2, map, dA, nullptr, dC, nullptr) == GSR_EINVAL);
  EXPECT(gsr_ssim_backward(3, 50, 37, img1, img2, nullptr, 0.1f, dA, dB, dC, map, nullptr) == GSR_OK);
  Dev<uint8_t> vis(10);
  EXPECT(gsr_mark_visible(10, s.means, s.view, s.proj, vis, nullptr) == GSR_OK);
  EXPECT(gsr_mark_visible(10, nullptr, s.view, s.proj, vis, nullptr) == GSR_EINVAL);
  EXPECT(gsr_mark_visible(0, nullptr, nullptr, nullptr, nullptr, nullptr) == GSR_OK);
}

static void single_thread_sweep() {
  const int shapes[][3] = {{1, 1, 1}, {1, 16, 16}, {7, 17, 33}, {300, 160, 96}, {5000, 97, 131}, {70000, 512, 512}, {200000, 1024, 1024}};
  for (auto &sh : shapes) {
    Scene s(sh[0], sh[1], sh[2], 16);
    for (int mode = 0; mode < 2; mode++) {
      EXPECT(gsr_set_binning_mode(mode) == GSR_OK && gsr_get_binning_mode() == mode);
      const uint32_t Rs[] = {0u, 1u, 4097u, (uint32_t)sh[0] * 7u};
      for (uint32_t R : Rs) {
        forward_backward(s, true, false, R, 0, nullptr);
        forward_backward(s, false, true, R, 1, nullptr);
      }
    }
    EXPECT(gsr_set_binning_mode(GSR_BINNING_TILE_BUCKET) == GSR_OK);
    EXPECT(gsr_set_tuning("deterministic", 1) == GSR_OK);
    forward_backward(s, true, false, (uint32_t)sh[0] * 3u, 0, nullptr);
    EXPECT(gsr_set_tuning("deterministic", 0) == GSR_OK);
    for (int red = 0; red <= 3; red++)
      for (int waves = 1; waves <= 4; waves *= 2) {
        if ((red == 1 || red == 2) && !gsr_has_experiments()) {  // experiment kernels: refused by the default build, with a message
          EXPECT(gsr_set_tuning("blend_bwd_reduce", red) == GSR_EINVAL && strstr(gsr_last_error(), "experiment") != nullptr);
          continue;
        }
        EXPECT(gsr_set_tuning("blend_bwd_reduce", red) == GSR_OK && gsr_set_tuning("blend_bwd_waves", waves) == GSR_OK);
        EXPECT(gsr_set_tuning("blend_fwd_waves", waves) == GSR_OK);
        forward_backward(s, true, red == 3, 1000u, 0, nullptr);
      }
    EXPECT(gsr_set_tuning("blend_bwd_reduce", 3) == GSR_OK && gsr_set_tuning("blend_bwd_waves", 4) == GSR_OK);
    EXPECT(gsr_set_tuning("blend_fwd_waves", 4) == GSR_OK);
    for (int hist = 0; hist < 2; hist++) {
      EXPECT(gsr_set_tuning("bucket_hist", hist) == GSR_OK);
      async_forward_backward(s, 4096, nullptr, false);
      async_forward_backward(s, (size_t)sh[0] * 8 + 100, nullptr, true);
    }
    EXPECT(gsr_set_tuning("bucket_hist", 1) == GSR_OK);
  }
  Scene empty(0, 64, 64, 16);
  forward_backward(empty, true, false, 0, 0, nullptr);
}

// what the product does concurrently: forwards on the main thread, backwards on autograd's thread, knobs per stream, stage
// profiling switched and read from a third place
static void threaded() {
  std::atomic<bool> stop{false};
  hipstub_set_fake_readback(777u);  // one value for all threads: the hook is process-wide
  std::vector<std::thread> ts;
  for (int t = 0; t < 4; t++)
    ts.emplace_back([t, &stop] {
      gsr_stream_t stream = reinterpret_cast<gsr_stream_t>(uintptr_t(0x1000 + 0x100 * t));  // four distinct stream handles
      Scene s(2000 + 100 * t, 160 + 16 * t, 96, 16);
      for (int it = 0; it < 60 && !stop; it++) {
        EXPECT(gsr_set_stream_tuning(stream, "binning_mode", it & 1) == GSR_OK);
        EXPECT(gsr_set_stream_tuning(stream, "blend_bwd_waves", 1 << (it % 3)) == GSR_OK);
        forward_backward(s, (it & 2) != 0, (it & 4) != 0, 777u, 0, stream, true);
        if (it % 7 == 0) {
          EXPECT(gsr_set_stream_tuning(stream, "tile_cull", 5) == GSR_EINVAL);   // per-thread error string
          EXPECT(strstr(gsr_last_error(), "tile_cull") != nullptr);
        }
        if (it % 11 == 0) EXPECT(gsr_clear_stream_tuning(stream) == GSR_OK);
      }
      EXPECT(gsr_clear_stream_tuning(stream) == GSR_OK);
    });
  for (int it = 0; it < 200; it++) {
    EXPECT(gsr_profile_enable(it % 3 ? 0x3Fu : 0x10u) == GSR_OK);
    double ms;
    long n;
    for (int st = 0; st < 6; st++) EXPECT(gsr_profile_read(st, &ms, &n) == GSR_OK);
    if (it % 5 == 0) EXPECT(gsr_profile_reset() == GSR_OK);
    EXPECT(gsr_set_tuning("tile_order", it & 3) == GSR_OK);
    std::this_thread::yield();
  }
  for (auto &t : ts) t.join();
  EXPECT(gsr_profile_enable(0) == GSR_OK);
  EXPECT(gsr_set_tuning("tile_order", 1) == GSR_OK);
}

// the skinning-offset network's entry points (csrc/mlp.hip): argument checks and launch geometries, host layer only
static void offset_network() {
  const size_t n = gsr_lbs_offset_mlp_packed_floats();
  EXPECT(n > 0 && n % 4 == 0);
  std::vector<float> packed(n + 4), xyz(3 * 1000), out(24 * 1000), dout(24 * 1000);
  float *pk = packed.data();
  while (reinterpret_cast<uintptr_t>(pk) % 16) pk++;
  const size_t shape_w[5] = {128 * 63, 128 * 128, 128 * 128, 128 * 191, 24 * 128}, shape_b[5] = {128, 128, 128, 128, 24};
  std::vector<std::vector<float>> w(5), b(5), dw(5), db(5);
  const float *wp[5], *bp[5];
  float *dwp[5], *dbp[5];
  for (int l = 0; l < 5; l++) {
    w[l].assign(shape_w[l], 0.f), b[l].assign(shape_b[l], 0.f), dw[l].assign(shape_w[l], 0.f), db[l].assign(shape_b[l], 0.f);
    wp[l] = w[l].data(), bp[l] = b[l].data(), dwp[l] = dw[l].data(), dbp[l] = db[l].data();
  }
  EXPECT(gsr_lbs_offset_mlp_pack(wp, bp, pk, nullptr) == GSR_OK);
  EXPECT(gsr_lbs_offset_mlp_pack(nullptr, bp, pk, nullptr) == GSR_EINVAL);
  const float *hole[5] = {wp[0], wp[1], nullptr, wp[3], wp[4]};
  EXPECT(gsr_lbs_offset_mlp_pack(hole, bp, pk, nullptr) == GSR_EINVAL && strstr(gsr_last_error(), "layer 2") != nullptr);
  for (int P : {0, 1, 255, 256, 257, 1000}) EXPECT(gsr_lbs_offset_mlp_forward(P, xyz.data(), pk, out.data(), nullptr) == GSR_OK);
  EXPECT(gsr_lbs_offset_mlp_forward(-1, xyz.data(), pk, out.data(), nullptr) == GSR_EINVAL);
  EXPECT(gsr_lbs_offset_mlp_forward(10, xyz.data(), pk + 1, out.data(), nullptr) == GSR_EINVAL);   // misaligned fragments
  EXPECT(gsr_lbs_offset_mlp_forward(10, nullptr, pk, out.data(), nullptr) == GSR_EINVAL);
  EXPECT(gsr_lbs_offset_mlp_backward_workspace_floats(1000) == (size_t)1120 * 1024 && gsr_lbs_offset_mlp_backward_workspace_floats(0) == 0);
  std::vector<float> ws(gsr_lbs_offset_mlp_backward_workspace_floats(1000) + 4);
  float *wsp = ws.data();
  while (reinterpret_cast<uintptr_t>(wsp) % 16) wsp++;
  for (int P : {0, 1, 1000}) EXPECT(gsr_lbs_offset_mlp_backward(P, xyz.data(), pk, dout.data(), wsp, dwp, dbp, nullptr) == GSR_OK);
  EXPECT(gsr_lbs_offset_mlp_backward(10, xyz.data(), pk, dout.data(), wsp + 1, dwp, dbp, nullptr) == GSR_EINVAL);
  float *dhole[5] = {dwp[0], nullptr, dwp[2], dwp[3], dwp[4]};
  EXPECT(gsr_lbs_offset_mlp_backward(10, xyz.data(), pk, dout.data(), wsp, dhole, dbp, nullptr) == GSR_EINVAL);
}

int main(int argc, char **argv) {
  const bool threads_only = argc > 1 && !strcmp(argv[1], "threads");
  if (!threads_only) {
    check_carving();
    error_paths();
    single_thread_sweep();
    offset_network();
  }
  threaded();
  if (hipstub_bad_launches()) {
    fprintf(stderr, "FAIL: %ld launches with an invalid geometry\n", hipstub_bad_launches());
    g_fail++;
  }
  printf("host layer: %ld launches validated, %d failures\n", hipstub_launches(), (int)g_fail);
  return g_fail ? 1 : 0;
}
