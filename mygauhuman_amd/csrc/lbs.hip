// lbs.hip -- SMPL linear-blend skinning of the canonical Gaussians: the per-point part of
// GaussianModel.coarse_deform_c2source (scene/gaussian_model.py:776-872), forward and backward, batch size 1.
//
// Reference = ~40 small torch kernels per frame + a third-party brute-force k-NN (KNN_CUDA) + an autograd graph.
// Here one forward kernel and one backward kernel do everything per point:
//   nearest big-pose SMPL vertex (brute force over V vertices staged through LDS in 1024-vertex tiles, coalesced;
//   lowest index wins ties) -> skinning weights (optionally softmax(log(w + 1e-9) + learned offset)) ->
//   blended joint transforms A_big, A_pose (24-term sums of 3x4 matrices) -> inverse big-pose rotation ->
//   per-vertex offsets -> posed point / normal / 3x3 transform / translation -> world frame (R^-1, Th).
// The joint transforms A[24][16] and the per-vertex offset tables are tiny per-frame inputs (host side,
// mygauhuman_amd/lbs.py).  The 24 x 12 blend is done on the vector ALU: per point it is a K = 24 dot product against
// a table that lives in SGPRs/LDS, i.e. bandwidth-trivial; the MFMA path is not used here.
//
// Backward: gradients w.r.t. the query points, normals, the learned weight offsets, A_pose [24][16] and the target-pose
// offset table [V][3] (the quantities that receive gradients in the reference training loop: xyz, the LBS-weight
// MLP, the pose-refinement MLP through A_pose and the pose blend shapes).  dA_pose is reduced per workgroup in LDS
// as a [24 x 256] x [256 x 12] LDS product and flushed with 288 atomics per workgroup; d(off_pose) is a per-vertex scatter-add.
#include <float.h>

#include "gsr_common.h"

namespace gsr {

constexpr int NJ = 24;
constexpr int LBS_BLOCK = 256;
constexpr int VTILE = 1024;

// squared distance with one IEEE rounding per operation, (dx*dx + dy*dy) + dz*dz -- the oracle's expression; both
// search variants use it, so the grid search returns bit-identical indices to the brute-force scan
__device__ __forceinline__ float sqdist_exact(float vx, float vy, float vz, const float *q) {
  const float dx = vx - q[0], dy = vy - q[1], dz = vz - q[2];
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- uniform grid over the V reference vertices (rebuilt per call by one workgroup; V is a few thousand) ----------------
// workspace: GridHeader | cell_start[ncell + 1] | float4 sorted[V] (x, y, z, original index bits), cells x-fastest.
constexpr int GRID_RES = 48;                                   // cells along the longest bounding-box axis
constexpr int GRID_MAXCELLS = (GRID_RES + 1) * (GRID_RES + 1) * (GRID_RES + 1);
constexpr int GRID_BLOCK = 1024;
struct GridHeader {
  float bbmin[3], h, inv_h;
  int dims[3];
};
constexpr size_t GRID_HDR_BYTES = 64;
__host__ __device__ inline size_t grid_cells_offset() { return GRID_HDR_BYTES; }
__host__ __device__ inline size_t grid_sorted_offset() {
  return (GRID_HDR_BYTES + (size_t)(GRID_MAXCELLS + 1) * 4 + 15) / 16 * 16;
}
inline size_t grid_workspace_bytes(int V) { return grid_sorted_offset() + (size_t)V * 16; }

__device__ __forceinline__ int grid_cell_coord(float x, float lo, float inv_h, int dim) {
  const int c = (int)floorf((x - lo) * inv_h);
  return min(max(c, 0), dim - 1);
}

__global__ __launch_bounds__(GRID_BLOCK) void lbs_grid_build_kernel(int V, const float *verts, char *ws) {
  __shared__ float s_red[6][GRID_BLOCK / WAVE];
  __shared__ GridHeader s_h;
  __shared__ uint32_t s_scan[GRID_BLOCK];
  GridHeader *hdr = reinterpret_cast<GridHeader *>(ws);
  uint32_t *cells = reinterpret_cast<uint32_t *>(ws + grid_cells_offset());
  float4 *sorted = reinterpret_cast<float4 *>(ws + grid_sorted_offset());
  const int t = threadIdx.x;
  // bounding box
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int v = t; v < V; v += GRID_BLOCK)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float x = verts[3 * (size_t)v + k];
      lo[k] = fminf(lo[k], x);
      hi[k] = fmaxf(hi[k], x);
    }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
    }
    if ((t & 63) == 0) {
      s_red[k][t >> 6] = lo[k];
      s_red[3 + k][t >> 6] = hi[k];
    }
  }
  __syncthreads();
  if (t == 0) {
    float ext = 0.f;
    for (int k = 0; k < 3; k++) {
      float l = FLT_MAX, h = -FLT_MAX;
      for (int w = 0; w < GRID_BLOCK / WAVE; w++) {
        l = fminf(l, s_red[k][w]);
        h = fmaxf(h, s_red[3 + k][w]);
      }
      lo[k] = l;
      hi[k] = h;
      ext = fmaxf(ext, h - l);
    }
    const float h = fmaxf(ext / (float)GRID_RES, 1e-12f);
    s_h.h = h;
    s_h.inv_h = 1.0f / h;
    for (int k = 0; k < 3; k++) {
      s_h.bbmin[k] = lo[k];
      s_h.dims[k] = min(GRID_RES + 1, (int)floorf((hi[k] - lo[k]) / h) + 1);
    }
    *hdr = s_h;
  }
  __syncthreads();
  const GridHeader g = s_h;
  const int ncell = g.dims[0] * g.dims[1] * g.dims[2];
  for (int c = t; c <= ncell; c += GRID_BLOCK) cells[c] = 0u;
  __syncthreads();
  // count (cells[c + 1] holds the count of cell c)
  for (int v = t; v < V; v += GRID_BLOCK) {
    const int cx = grid_cell_coord(verts[3 * (size_t)v], g.bbmin[0], g.inv_h, g.dims[0]);
    const int cy = grid_cell_coord(verts[3 * (size_t)v + 1], g.bbmin[1], g.inv_h, g.dims[1]);
    const int cz = grid_cell_coord(verts[3 * (size_t)v + 2], g.bbmin[2], g.inv_h, g.dims[2]);
    atomicAdd(&cells[(cz * g.dims[1] + cy) * g.dims[0] + cx + 1], 1u);
  }
  __syncthreads();
  // exclusive scan of the counts, in place: cells[c + 1] <- start of cell c
  const int chunk = (ncell + GRID_BLOCK - 1) / GRID_BLOCK;
  const int c0 = min(t * chunk, ncell), c1 = min(c0 + chunk, ncell);
  uint32_t sum = 0;
  for (int c = c0; c < c1; c++) sum += cells[c + 1];
  s_scan[t] = sum;
  __syncthreads();
  for (int off = 1; off < GRID_BLOCK; off <<= 1) {
    const uint32_t add = t >= off ? s_scan[t - off] : 0u;
    __syncthreads();
    s_scan[t] += add;
    __syncthreads();
  }
  uint32_t run = s_scan[t] - sum;
  for (int c = c0; c < c1; c++) {
    const uint32_t n = cells[c + 1];
    cells[c + 1] = run;
    run += n;
  }
  __syncthreads();
  // scatter: the cursor of cell c is cells[c + 1]; afterwards cells[k] = start of cell k for k = 0..ncell
  for (int v = t; v < V; v += GRID_BLOCK) {
    const float x = verts[3 * (size_t)v], y = verts[3 * (size_t)v + 1], z = verts[3 * (size_t)v + 2];
    const int cx = grid_cell_coord(x, g.bbmin[0], g.inv_h, g.dims[0]);
    const int cy = grid_cell_coord(y, g.bbmin[1], g.inv_h, g.dims[1]);
    const int cz = grid_cell_coord(z, g.bbmin[2], g.inv_h, g.dims[2]);
    const uint32_t pos = atomicAdd(&cells[(cz * g.dims[1] + cy) * g.dims[0] + cx + 1], 1u);
    sorted[pos] = make_float4(x, y, z, __int_as_float(v));
  }
}

// exact nearest vertex of q through the grid: rings of cells of growing Chebyshev radius around q's cell; after ring r every
// unvisited vertex is farther than r*h (r whole cells lie in between), so the search stops once best < that bound.
// (distance, index) is compared lexicographically: the lowest index wins ties, like the brute-force scan.
// other_d2 (optional): a LOWER bound on the squared distance from q to every vertex other than the returned one -- the second-best
// distance among the vertices examined, or the distance below which the rings left unexamined cannot reach, whichever is smaller.
__device__ __forceinline__ int grid_nearest(const char *ws, const float *q, float *best_d2 = nullptr, float *other_d2 = nullptr) {
  const GridHeader *g = reinterpret_cast<const GridHeader *>(ws);
  const uint32_t *cells = reinterpret_cast<const uint32_t *>(ws + grid_cells_offset());
  const float4 *sorted = reinterpret_cast<const float4 *>(ws + grid_sorted_offset());
  const float h = g->h, inv_h = g->inv_h;
  const int d0 = g->dims[0], d1 = g->dims[1], d2 = g->dims[2];
  const int c0 = grid_cell_coord(q[0], g->bbmin[0], inv_h, d0);
  const int c1 = grid_cell_coord(q[1], g->bbmin[1], inv_h, d1);
  const int c2 = grid_cell_coord(q[2], g->bbmin[2], inv_h, d2);
  const int rmax = max(max(max(c0, d0 - 1 - c0), max(c1, d1 - 1 - c1)), max(c2, d2 - 1 - c2));
  const float eps = 1e-3f * h;  // slack for the rounding of the cell assignment
  float best = FLT_MAX, second = FLT_MAX;
  int bid = 0x7fffffff;
  // Rings 0 and 1 in one go: the nine (z, y) rows around q's cell, each ONE contiguous run of the sorted list over x in
  // [c0 - 1, c0 + 1].  All eighteen run bounds are requested before any vertex is looked at: ring by ring (1 + 10 runs, each a
  // pair of dependent loads in front of its vertices) the search was a chain of ~22 memory round trips per point.
  {
    uint32_t rb[9], re[9];
    const int xa = max(c0 - 1, 0), xb = min(c0 + 1, d0 - 1);
#pragma unroll
    for (int j = 0; j < 9; j++) {
      const int z = c2 + j / 3 - 1, y = c1 + j % 3 - 1;
      const bool in = z >= 0 && z < d2 && y >= 0 && y < d1;
      const int row = ((in ? z : 0) * d1 + (in ? y : 0)) * d0;
      rb[j] = cells[row + xa];
      re[j] = in ? cells[row + xb + 1] : rb[j];
    }
#pragma unroll
    for (int j = 0; j < 9; j++)
      for (uint32_t i = rb[j]; i < re[j]; i++) {
        const float4 v = sorted[i];
        const float d = sqdist_exact(v.x, v.y, v.z, q);
        const int id = __float_as_int(v.w);
        if (d < best || (d == best && id < bid)) {
          second = best;
          best = d;
          bid = id;
        } else {
          second = fminf(second, d);
        }
      }
  }
  float unseen = FLT_MAX;  // squared distance every vertex of the rings NOT examined exceeds (all examined: no bound needed)
  for (int r = 2; r <= rmax; r++) {
    {
      const float lb = (float)(r - 1) * h - eps;
      if (best < lb * lb) {
        unseen = lb * lb;
        break;
      }
    }
    const int z0 = max(c2 - r, 0), z1 = min(c2 + r, d2 - 1), y0 = max(c1 - r, 0), y1 = min(c1 + r, d1 - 1);
    const int x0 = max(c0 - r, 0), x1 = min(c0 + r, d0 - 1);
    for (int z = z0; z <= z1; z++)
      for (int y = y0; y <= y1; y++) {
        const int row = (z * d1 + y) * d0;
        const bool shell = (z - c2 == r) || (c2 - z == r) || (y - c1 == r) || (c1 - y == r);
        // shell rows: the whole x range is one contiguous run of the sorted list; inner rows: the two end cells only
        for (int part = 0; part < (shell ? 1 : 2); part++) {
          uint32_t b, e;
          if (shell) {
            b = cells[row + x0];
            e = cells[row + x1 + 1];
          } else {
            const int x = part == 0 ? c0 - r : c0 + r;
            if (x < 0 || x >= d0) continue;
            b = cells[row + x];
            e = cells[row + x + 1];
          }
          for (uint32_t i = b; i < e; i++) {
            const float4 v = sorted[i];
            const float d = sqdist_exact(v.x, v.y, v.z, q);
            const int id = __float_as_int(v.w);
            if (d < best || (d == best && id < bid)) {
              second = best;
              best = d;
              bid = id;
            } else {
              second = fminf(second, d);
            }
          }
        }
      }
  }
  if (best_d2) *best_d2 = best;
  if (other_d2) *other_d2 = fminf(second, unseen);
  return bid;
}

// stand-alone nearest-reference-point query through the grid (the k = 1, ref != query use of KNN_CUDA,
// scene/gaussian_model.py:727: distance of every Gaussian to the SMPL surface)
__global__ __launch_bounds__(256) void grid_nearest_kernel(int M, const float *query, const char *ws, int *idx, float *dist) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const float q[3] = {query[3 * (size_t)i], query[3 * (size_t)i + 1], query[3 * (size_t)i + 2]};
  float d2;
  const int id = grid_nearest(ws, q, &d2);
  if (idx) idx[i] = id;
  if (dist) dist[i] = sqrtf(d2);
}

// ---- exact temporal cache of the nearest vertex (scene/gaussian_model.py:775 searches every frame) ----
// The canonical Gaussians move by an optimizer step per iteration; the vertex cloud they are matched against (the subject's
// big-pose vertices) does not move at all.  An entry (x0, id, rho) says: vertex `id` is the strict nearest vertex of EVERY point
// within rho of x0.  It is made by a full search that also returns a lower bound D2 on the distance of every other vertex:
// with d1 = |x0 - v_id|, for |x - x0| = delta the triangle inequality gives |x - v_id| <= d1 + delta and |x - v| >= D2 - delta for
// every other v, so delta < (D2 - d1) / 2 keeps `id`.  rho = 0.49 (D2 - d1) minus a relative guard for the rounding of the two
// distances (an entry whose gap is within rounding of zero gets rho = 0 and is searched every frame).  The statement is about the
// VERTEX SET only: it stays true whatever point sits in slot p (densify / prune may reuse the slot), so the cache is invalidated
// only when the vertex tensor changes; a slot whose point has moved too far is searched again and re-centred.
// One small kernel in front of the skinning kernel (a per-lane fall-back search inside it would leave every wave waiting for its one
// or two misses): check every point against its entry, compact the workgroup's misses, search them.  The skinning kernel then takes
// the ids as given.
struct NnCacheView {
  float4 *entry;       // [P] x0, y0, z0, rho  (rho < 0: never searched)
  int *ids;            // [P]
  uint32_t *miss;      // [P] (unused since the check and the search became one kernel; kept in the buffer layout)
  uint32_t *count;     // [0] misses of this frame, [1] searches since the cache was made (statistics)
};
__device__ __forceinline__ float nn_cache_rho(float best_d2, float other_d2) {
  if (!(other_d2 < FLT_MAX)) return 1e30f;  // a single vertex
  const float d1 = sqrtf(best_d2), D2 = sqrtf(other_d2);
  const float rho = 0.49f * (D2 - d1) - 4e-6f * D2 - 1e-12f;
  return rho > 0.f ? rho : 0.f;
}
// ONE kernel (round 4, after measuring the two-kernel form: a separate search launch cost 17 us per frame even with an empty miss
// list): every thread checks its point; the workgroup compacts its misses into LDS (ballot per wave, one LDS add per wave) and its
// first threads search them -- a few percent of the points in a training loop, i.e. a dozen searches on one wave of a workgroup
// whose other waves have already left.  No global list, no global atomics on the hot path (one statistics add per workgroup).
__global__ __launch_bounds__(256) void nn_cache_update_kernel(int P, const float *query, const char *grid, NnCacheView c) {
  __shared__ uint32_t s_miss[256];
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  bool miss = false;
  if (p < P) {
    const float4 e = c.entry[p];
    const float dx = query[3 * (size_t)p] - e.x, dy = query[3 * (size_t)p + 1] - e.y, dz = query[3 * (size_t)p + 2] - e.z;
    const float d2 = dx * dx + dy * dy + dz * dz;
    // (NaN positions miss, like a negative rho = never searched.  A point that has not moved AT ALL keeps its id whatever the radius:
    // entries whose two nearest vertices tie within rounding have rho = 0 and were searched again every frame -- in a loop over
    // static points (render.py, evaluation) their serial searches were the whole 15 us of this kernel)
    miss = !(e.w >= 0.f && (d2 == 0.f || d2 < e.w * e.w));
  }
  const uint64_t m = __ballot(miss);
  if (m != 0ull) {
    const uint32_t lane = lane_id();
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__builtin_popcountll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (miss) s_miss[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)p;
  }
  __syncthreads();
  const uint32_t n = s_n;
  if (n == 0u) return;
  if (threadIdx.x == 0) atomicAdd(&c.count[0], n);  // statistics only
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    const uint32_t q_i = s_miss[i];
    const float q[3] = {query[3 * (size_t)q_i], query[3 * (size_t)q_i + 1], query[3 * (size_t)q_i + 2]};
    float best, other;
    const int id = grid_nearest(grid, q, &best, &other);
    c.ids[q_i] = id;
    c.entry[q_i] = make_float4(q[0], q[1], q[2], nn_cache_rho(best, other));
  }
}
// (the frame's bookkeeping -- statistics += this frame's misses, counter back to zero for the next frame -- is done by thread 0 of
// the skinning kernel that follows the search: LbsArgs::cache_count)
__host__ __device__ inline size_t nn_cache_bytes(size_t P) { return P * (sizeof(float4) + 2 * sizeof(uint32_t)) + 64; }
inline NnCacheView nn_cache_view(char *buf, size_t P) {
  NnCacheView c;
  c.entry = reinterpret_cast<float4 *>(buf);
  c.ids = reinterpret_cast<int *>(buf + P * sizeof(float4));
  c.miss = reinterpret_cast<uint32_t *>(buf + P * (sizeof(float4) + sizeof(uint32_t)));
  c.count = reinterpret_cast<uint32_t *>(buf + P * (sizeof(float4) + 2 * sizeof(uint32_t)));
  return c;
}

struct LbsArgs {
  int P, V;
  const float *query, *normals, *smpl_verts, *weights, *lbs_offsets, *A_big, *A_pose, *off_big, *off_shape, *off_pose, *R, *Th;
  int *vert_ids;
  float *bweights, *smpl_pts, *world_pts, *transforms, *translation, *world_normals;
  const char *grid;  // vertex grid workspace (GRID variant)
  const int *given_ids;  // GRID variant: the nearest vertex of every point is already known (the temporal cache): no search
  float4 *cache_entry;   // GRID variant with a search: also leave (x0, rho) and the id for the temporal cache (null: none)
  int *cache_ids;
  uint32_t *cache_count; // given_ids: [0] this frame's misses -> [2], added to [1], reset
};

__device__ __forceinline__ void inv3(const float *m, float *o) {
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const float id = 1.0f / det;
  o[0] = c00 * id;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
__device__ __forceinline__ void mat3_vec(const float *M, const float *v, float *o) {
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2];
}
__device__ __forceinline__ void mat3T_vec(const float *M, const float *v, float *o) {
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = M[c] * v[0] + M[3 + c] * v[1] + M[6 + c] * v[2];
}
__device__ __forceinline__ void mat3_mul(const float *A, const float *B, float *o) {
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) o[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// blend weights of one point (scene/gaussian_model.py:776-781)
__device__ __forceinline__ void blend_weights(const float *w_row, const float *off_row, float *bw) {
#pragma unroll
  for (int j = 0; j < NJ; j++) bw[j] = w_row[j];
  if (off_row) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      bw[j] = logf(bw[j] + 1e-9f) + off_row[j];
      mx = fmaxf(mx, bw[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      bw[j] = expf(bw[j] - mx);
      sum += bw[j];
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) bw[j] = bw[j] / sum;
  }
}

// rows 0..2 of the blended 4x4 transform: out[12] = sum_j bw[j] * A[j][0..11]   (A in LDS)
__device__ __forceinline__ void blend_A(const float *bw, const float *sA, float *out) {
#pragma unroll
  for (int k = 0; k < 12; k++) out[k] = 0.f;
  // joints two at a time: fully unrolled, the 24 x 12 LDS operands get hoisted and the kernel ends up at 256 VGPRs + scratch
#pragma unroll 2
  for (int j = 0; j < NJ; j++) {
    const float4 r0 = *reinterpret_cast<const float4 *>(&sA[16 * j]);
    const float4 r1 = *reinterpret_cast<const float4 *>(&sA[16 * j + 4]);
    const float4 r2 = *reinterpret_cast<const float4 *>(&sA[16 * j + 8]);
    const float b = bw[j];
    out[0] += b * r0.x;
    out[1] += b * r0.y;
    out[2] += b * r0.z;
    out[3] += b * r0.w;
    out[4] += b * r1.x;
    out[5] += b * r1.y;
    out[6] += b * r1.z;
    out[7] += b * r1.w;
    out[8] += b * r2.x;
    out[9] += b * r2.y;
    out[10] += b * r2.z;
    out[11] += b * r2.w;
  }
}

template <bool GRID>
__global__ __launch_bounds__(LBS_BLOCK) void lbs_forward_kernel(const LbsArgs a) {
  __shared__ float svx[GRID ? 1 : VTILE], svy[GRID ? 1 : VTILE], svz[GRID ? 1 : VTILE];
  __shared__ __attribute__((aligned(16))) float sAb[NJ * 16], sAp[NJ * 16];
  const int p = blockIdx.x * LBS_BLOCK + threadIdx.x;
  const bool live = p < a.P;
  for (int k = threadIdx.x; k < NJ * 16; k += LBS_BLOCK) {
    sAb[k] = a.A_big[k];
    sAp[k] = a.A_pose[k];
  }
  float q[3] = {0, 0, 0};
  if (live) {
    q[0] = a.query[3 * (size_t)p];
    q[1] = a.query[3 * (size_t)p + 1];
    q[2] = a.query[3 * (size_t)p + 2];
  }
  // ---- nearest vertex (k = 1, squared Euclidean distance, first minimum wins)
  float best = FLT_MAX;
  int bid = 0;
  if (GRID) {
    __syncthreads();  // sAb / sAp
    if (live) {
      if (a.given_ids) {
        bid = a.given_ids[p];
        if (p == 0 && a.cache_count) {
          a.cache_count[1] += a.cache_count[0];
          a.cache_count[2] = a.cache_count[0];
          a.cache_count[0] = 0u;
        }
      } else if (a.cache_entry) {  // a full search that also makes the cache entries
        float bd, od;
        bid = grid_nearest(a.grid, q, &bd, &od);
        a.cache_entry[p] = make_float4(q[0], q[1], q[2], nn_cache_rho(bd, od));
        a.cache_ids[p] = bid;
      } else {
        bid = grid_nearest(a.grid, q);
      }
    }
  }
  for (int v0 = 0; !GRID && v0 < a.V; v0 += VTILE) {
    __syncthreads();
    const int cnt = min(VTILE, a.V - v0);
    for (int i = threadIdx.x; i < cnt; i += LBS_BLOCK) {
      svx[i] = a.smpl_verts[3 * (size_t)(v0 + i)];
      svy[i] = a.smpl_verts[3 * (size_t)(v0 + i) + 1];
      svz[i] = a.smpl_verts[3 * (size_t)(v0 + i) + 2];
    }
    __syncthreads();
    if (live) {
      for (int i = 0; i < cnt; i++) {
        const float d = sqdist_exact(svx[i], svy[i], svz[i], q);
        if (d < best) {
          best = d;
          bid = v0 + i;
        }
      }
    }
  }
  if (!live) return;
  if (a.vert_ids) a.vert_ids[p] = bid;
  float bw[NJ];
  blend_weights(a.weights + (size_t)bid * NJ, a.lbs_offsets ? a.lbs_offsets + (size_t)p * NJ : nullptr, bw);
  if (a.bweights)
#pragma unroll
    for (int j = 0; j < NJ; j++) a.bweights[(size_t)p * NJ + j] = bw[j];
  float Ab[12], Ap[12];
  blend_A(bw, sAb, Ab);
  blend_A(bw, sAp, Ap);
  const float Rb[9] = {Ab[0], Ab[1], Ab[2], Ab[4], Ab[5], Ab[6], Ab[8], Ab[9], Ab[10]};
  float Ri[9];
  inv3(Rb, Ri);
  const float q0[3] = {q[0] - Ab[3], q[1] - Ab[7], q[2] - Ab[11]};
  const float t0[3] = {-Ab[3], -Ab[7], -Ab[11]};
  float q1[3], n1[3] = {0, 0, 0}, tr[3];
  mat3_vec(Ri, q0, q1);
  mat3_vec(Ri, t0, tr);
  if (a.normals) {
    const float nn[3] = {a.normals[3 * (size_t)p], a.normals[3 * (size_t)p + 1], a.normals[3 * (size_t)p + 2]};
    mat3_vec(Ri, nn, n1);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float ob = a.off_big[3 * (size_t)bid + k], os = a.off_shape[3 * (size_t)bid + k], op = a.off_pose[3 * (size_t)bid + k];
    q1[k] = ((q1[k] - ob) + os) + op;
    tr[k] = ((tr[k] - ob) + os) + op;
  }
  const float Rp[9] = {Ap[0], Ap[1], Ap[2], Ap[4], Ap[5], Ap[6], Ap[8], Ap[9], Ap[10]};
  const float tp[3] = {Ap[3], Ap[7], Ap[11]};
  float can[3], sn[3], tr2[3], M1[9];
  mat3_vec(Rp, q1, can);
  mat3_vec(Rp, n1, sn);
  mat3_vec(Rp, tr, tr2);
  mat3_mul(Rp, Ri, M1);
  float Rw[9], Rinv[9];
#pragma unroll
  for (int k = 0; k < 9; k++) Rw[k] = a.R[k];
  inv3(Rw, Rinv);
  const float src[3] = {can[0] + tp[0], can[1] + tp[1], can[2] + tp[2]};
  float wsrc[3], wn[3], wt[3];
  const float tr3[3] = {tr2[0] + tp[0], tr2[1] + tp[1], tr2[2] + tp[2]};
  mat3T_vec(Rinv, src, wsrc);  // row vector times R_inv (gaussian_model.py:864)
  mat3T_vec(Rinv, sn, wn);
  mat3T_vec(Rinv, tr3, wt);
  float Mw[9];
  mat3_mul(Rw, M1, Mw);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (a.smpl_pts) a.smpl_pts[3 * (size_t)p + k] = src[k];
    a.world_pts[3 * (size_t)p + k] = wsrc[k] + a.Th[k];
    if (a.world_normals) a.world_normals[3 * (size_t)p + k] = wn[k];
    if (a.translation) a.translation[3 * (size_t)p + k] = wt[k] + a.Th[k];
  }
  if (a.transforms)
#pragma unroll
    for (int k = 0; k < 9; k++) a.transforms[9 * (size_t)p + k] = Mw[k];
}

struct LbsBwdArgs {
  int P, V;
  const float *query, *normals, *weights, *lbs_offsets, *A_big, *A_pose, *off_big, *off_shape, *off_pose, *R;
  const int *vert_ids;
  const float *dL_dworld_pts, *dL_dtransforms, *dL_dworld_normals;
  float *dL_dquery, *dL_dnormals, *dL_dlbs_offsets, *dL_dA_pose, *dL_doff_pose;
  float *partials;  // [workgroups][24 * 12] per-workgroup dA_pose sums (no atomics) or null
};

// (amdgpu_waves_per_eu(4): the allocator stopped at 132 VGPRs = three waves per SIMD where the 40 KB of LDS admit four; at 128 with
// four spilled registers the kernel takes 25.6 instead of 32.9 us in the render() frame.  The same hint on the forward kernel -- 108
// VGPRs, asked for 80 -- spills in the search loop: 76 instead of 54 us.)
__global__ __launch_bounds__(LBS_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void lbs_backward_kernel(const LbsBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sAb[NJ * 16], sAp[NJ * 16];
  constexpr int ROW = NJ + 12 + 1;                // bw[24] | g_Ap[12] | pad (odd stride: conflict-free column reads)
  __shared__ float s_rows[LBS_BLOCK * ROW];       // per-point operands of the workgroup-level dA_pose product
  for (int k = threadIdx.x; k < NJ * 16; k += LBS_BLOCK) {
    sAb[k] = a.A_big[k];
    sAp[k] = a.A_pose[k];
  }
  for (int k = threadIdx.x; k < LBS_BLOCK * ROW; k += LBS_BLOCK) s_rows[k] = 0.f;
  __syncthreads();
  const int p = blockIdx.x * LBS_BLOCK + threadIdx.x;
  if (p < a.P) {
    const int bid = a.vert_ids[p];
    float bw[NJ];
    blend_weights(a.weights + (size_t)bid * NJ, a.lbs_offsets ? a.lbs_offsets + (size_t)p * NJ : nullptr, bw);
    float Ab[12], Ap[12];
    blend_A(bw, sAb, Ab);
    blend_A(bw, sAp, Ap);
    const float Rb[9] = {Ab[0], Ab[1], Ab[2], Ab[4], Ab[5], Ab[6], Ab[8], Ab[9], Ab[10]};
    float Ri[9];
    inv3(Rb, Ri);
    const float q[3] = {a.query[3 * (size_t)p], a.query[3 * (size_t)p + 1], a.query[3 * (size_t)p + 2]};
    const float q0[3] = {q[0] - Ab[3], q[1] - Ab[7], q[2] - Ab[11]};
    float q1[3], n1[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
    mat3_vec(Ri, q0, q1);
    if (a.normals) {
      nn[0] = a.normals[3 * (size_t)p];
      nn[1] = a.normals[3 * (size_t)p + 1];
      nn[2] = a.normals[3 * (size_t)p + 2];
      mat3_vec(Ri, nn, n1);
    }
    float q2[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
      q2[k] = ((q1[k] - a.off_big[3 * (size_t)bid + k]) + a.off_shape[3 * (size_t)bid + k]) + a.off_pose[3 * (size_t)bid + k];
    const float Rp[9] = {Ap[0], Ap[1], Ap[2], Ap[4], Ap[5], Ap[6], Ap[8], Ap[9], Ap[10]};
    float Rw[9], Rinv[9];
#pragma unroll
    for (int k = 0; k < 9; k++) Rw[k] = a.R[k];
    inv3(Rw, Rinv);

    // ---- incoming gradients, pulled back through the world transform
    float g_src[3] = {0, 0, 0}, g_sn[3] = {0, 0, 0}, g_M1[9];
#pragma unroll
    for (int k = 0; k < 9; k++) g_M1[k] = 0.f;
    if (a.dL_dworld_pts) {  // world = src . Rinv + Th  (row vector)  ->  g_src[m] = sum_k Rinv[m][k] g[k]
      const float g[3] = {a.dL_dworld_pts[3 * (size_t)p], a.dL_dworld_pts[3 * (size_t)p + 1], a.dL_dworld_pts[3 * (size_t)p + 2]};
      mat3_vec(Rinv, g, g_src);
    }
    if (a.dL_dworld_normals && a.normals) {
      const float g[3] = {a.dL_dworld_normals[3 * (size_t)p], a.dL_dworld_normals[3 * (size_t)p + 1], a.dL_dworld_normals[3 * (size_t)p + 2]};
      mat3_vec(Rinv, g, g_sn);
    }
    if (a.dL_dtransforms) {  // transforms = Rw . M1  ->  g_M1 = Rw^T . g
      float g[9];
#pragma unroll
      for (int k = 0; k < 9; k++) g[k] = a.dL_dtransforms[9 * (size_t)p + k];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) g_M1[3 * r + c] = Rw[r] * g[c] + Rw[3 + r] * g[3 + c] + Rw[6 + r] * g[6 + c];
    }
    // src = Rp q2 + tp ; sn = Rp n1 ; M1 = Rp Ri
    float g_Rp[9], g_tp[3], g_q2[3], g_n1[3], g_Ri[9];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      g_tp[r] = g_src[r];
#pragma unroll
      for (int c = 0; c < 3; c++)
        g_Rp[3 * r + c] = g_src[r] * q2[c] + g_sn[r] * n1[c] + (g_M1[3 * r] * Ri[3 * c] + g_M1[3 * r + 1] * Ri[3 * c + 1] + g_M1[3 * r + 2] * Ri[3 * c + 2]);
    }
    mat3T_vec(Rp, g_src, g_q2);
    mat3T_vec(Rp, g_sn, g_n1);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) g_Ri[3 * r + c] = Rp[r] * g_M1[c] + Rp[3 + r] * g_M1[3 + c] + Rp[6 + r] * g_M1[6 + c];
    // q2 = q1 - off_big + off_shape + off_pose
    float g_q1[3] = {g_q2[0], g_q2[1], g_q2[2]};
    if (a.dL_doff_pose)
#pragma unroll
      for (int k = 0; k < 3; k++) atomicAdd(&a.dL_doff_pose[3 * (size_t)bid + k], g_q2[k]);
    // q1 = Ri q0 ; n1 = Ri nn
    float g_q0[3];
    mat3T_vec(Ri, g_q1, g_q0);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) g_Ri[3 * r + c] += g_q1[r] * q0[c] + g_n1[r] * nn[c];
    if (a.dL_dnormals && a.normals) {
      float g_nn[3];
      mat3T_vec(Ri, g_n1, g_nn);
#pragma unroll
      for (int k = 0; k < 3; k++) a.dL_dnormals[3 * (size_t)p + k] = g_nn[k];
    }
    // Ri = inverse(Rb):  g_Rb = -Ri^T g_Ri Ri^T
    float tmp[9], g_Rb[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) tmp[3 * r + c] = Ri[r] * g_Ri[c] + Ri[3 + r] * g_Ri[3 + c] + Ri[6 + r] * g_Ri[6 + c];  // Ri^T g_Ri
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) g_Rb[3 * r + c] = -(tmp[3 * r] * Ri[3 * c] + tmp[3 * r + 1] * Ri[3 * c + 1] + tmp[3 * r + 2] * Ri[3 * c + 2]);
    // q0 = query - tb
    const float g_tb[3] = {-g_q0[0], -g_q0[1], -g_q0[2]};
#pragma unroll
    for (int k = 0; k < 3; k++) a.dL_dquery[3 * (size_t)p + k] = g_q0[k];
    // blended rows: g_Ab[12], g_Ap[12]
    const float g_Ab[12] = {g_Rb[0], g_Rb[1], g_Rb[2], g_tb[0], g_Rb[3], g_Rb[4], g_Rb[5], g_tb[1], g_Rb[6], g_Rb[7], g_Rb[8], g_tb[2]};
    const float g_Ap[12] = {g_Rp[0], g_Rp[1], g_Rp[2], g_tp[0], g_Rp[3], g_Rp[4], g_Rp[5], g_tp[1], g_Rp[6], g_Rp[7], g_Rp[8], g_tp[2]};
    // dA_pose[j][k] += bw[j] g_Ap[k]  (workgroup-level LDS accumulation) ; g_bw[j] = <g_Ab, A_big[j]> + <g_Ap, A_pose[j]>
    float g_bw[NJ];
#pragma unroll 2
    for (int j = 0; j < NJ; j++) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 12; k++) s += g_Ab[k] * sAb[16 * j + k] + g_Ap[k] * sAp[16 * j + k];
      g_bw[j] = s;
    }
    if (a.dL_dA_pose) {
#pragma unroll
      for (int j = 0; j < NJ; j++) s_rows[threadIdx.x * ROW + j] = bw[j];
#pragma unroll
      for (int k = 0; k < 12; k++) s_rows[threadIdx.x * ROW + NJ + k] = g_Ap[k];
    }
    if (a.dL_dlbs_offsets && a.lbs_offsets) {  // softmax backward: g_z[j] = bw[j] (g_bw[j] - sum_i bw[i] g_bw[i])
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; j++) dot += bw[j] * g_bw[j];
#pragma unroll
      for (int j = 0; j < NJ; j++) a.dL_dlbs_offsets[(size_t)p * NJ + j] = bw[j] * (g_bw[j] - dot);
    }
  }
  __syncthreads();
  // dA_pose[j][k] = sum over the workgroup's points of bw[p][j] * g_Ap[p][k]: a [24 x 256] x [256 x 12] product out of LDS,
  // one output entry per thread, then one atomic per entry and workgroup
  if (a.dL_dA_pose)
    for (int e = threadIdx.x; e < NJ * 12; e += LBS_BLOCK) {
      const int j = e / 12, k = e % 12;
      float acc = 0.f;
      for (int q = 0; q < LBS_BLOCK; q++) acc += s_rows[q * ROW + j] * s_rows[q * ROW + NJ + k];
      if (a.partials)
        a.partials[(size_t)blockIdx.x * (NJ * 12) + e] = acc;  // reduced by the caller: 782 workgroups x 288 atomics onto the
      else if (acc != 0.f)                                       // same 288 addresses serialise at the memory side
        atomicAdd(&a.dL_dA_pose[16 * j + k], acc);
    }
}

}  // namespace gsr

extern "C" {

int gsr_lbs_forward(int P, int V, const float *query, const float *normals, const float *smpl_verts, const float *weights,
                    const float *lbs_offsets, const float *A_big, const float *A_pose, const float *off_big,
                    const float *off_shape, const float *off_pose, const float *R, const float *Th, int *vert_ids,
                    float *bweights, float *smpl_pts, float *world_pts, float *transforms, float *translation,
                    float *world_normals, gsr_stream_t stream_) {
  if (P < 0 || V <= 0 || (P > 0 && (!query || !smpl_verts || !weights || !A_big || !A_pose || !off_big || !off_shape ||
                                    !off_pose || !R || !Th || !world_pts))) {
    gsr::set_error("gsr_lbs_forward: bad arguments");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  gsr::LbsArgs a = {P, V, query, normals, smpl_verts, weights, lbs_offsets, A_big, A_pose, off_big, off_shape, off_pose, R, Th,
                    vert_ids, bweights, smpl_pts, world_pts, transforms, translation, world_normals, nullptr};
  hipLaunchKernelGGL(gsr::lbs_forward_kernel<false>, dim3((P + gsr::LBS_BLOCK - 1) / gsr::LBS_BLOCK), dim3(gsr::LBS_BLOCK), 0,
                     stream, a);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_lbs_backward_workgroups(int P) { return P > 0 ? (P + gsr::LBS_BLOCK - 1) / gsr::LBS_BLOCK : 0; }

size_t gsr_lbs_workspace_bytes(int V) { return V > 0 ? gsr::grid_workspace_bytes(V) : 0; }

int gsr_lbs_grid_build(int V, const float *smpl_verts, char *workspace, size_t workspace_bytes, gsr_stream_t stream_) {
  if (V <= 0 || !smpl_verts || !workspace || workspace_bytes < gsr::grid_workspace_bytes(V) ||
      reinterpret_cast<size_t>(workspace) % 16 != 0) {
    gsr::set_error("gsr_lbs_grid_build: bad arguments (workspace of %zu bytes, 16-byte aligned)", gsr::grid_workspace_bytes(V > 0 ? V : 1));
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(gsr::lbs_grid_build_kernel, dim3(1), dim3(gsr::GRID_BLOCK), 0, stream, V, smpl_verts, workspace);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_knn_nearest(int M, const float *query, int N, const float *ref, int *idx, float *dist, char *workspace,
                    size_t workspace_bytes, gsr_stream_t stream_) {
  if (M < 0 || N <= 0 || !ref || (M > 0 && (!query || (!idx && !dist)))) {
    gsr::set_error("gsr_knn_nearest: bad arguments");
    return GSR_EINVAL;
  }
  if (M == 0) return GSR_OK;
  if (!workspace || workspace_bytes < gsr::grid_workspace_bytes(N) || reinterpret_cast<size_t>(workspace) % 16 != 0) {
    gsr::set_error("gsr_knn_nearest: workspace of %zu bytes (16-byte aligned) required, got %zu", gsr::grid_workspace_bytes(N),
                   workspace_bytes);
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(gsr::lbs_grid_build_kernel, dim3(1), dim3(gsr::GRID_BLOCK), 0, stream, N, ref, workspace);
  hipLaunchKernelGGL(gsr::grid_nearest_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, M, query, workspace, idx, dist);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_lbs_forward_grid(int P, int V, const float *query, const float *normals, const float *smpl_verts, const float *weights,
                         const float *lbs_offsets, const float *A_big, const float *A_pose, const float *off_big,
                         const float *off_shape, const float *off_pose, const float *R, const float *Th, int *vert_ids,
                         float *bweights, float *smpl_pts, float *world_pts, float *transforms, float *translation,
                         float *world_normals, char *workspace, size_t workspace_bytes, int grid_is_built, gsr_stream_t stream_) {
  if (P < 0 || V <= 0 || (P > 0 && (!query || !smpl_verts || !weights || !A_big || !A_pose || !off_big || !off_shape ||
                                    !off_pose || !R || !Th || !world_pts))) {
    gsr::set_error("gsr_lbs_forward_grid: bad arguments");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  if (!workspace || workspace_bytes < gsr::grid_workspace_bytes(V) || reinterpret_cast<size_t>(workspace) % 16 != 0) {
    gsr::set_error("gsr_lbs_forward_grid: workspace of %zu bytes (16-byte aligned) required, got %zu",
                   gsr::grid_workspace_bytes(V), workspace_bytes);
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!grid_is_built)
    hipLaunchKernelGGL(gsr::lbs_grid_build_kernel, dim3(1), dim3(gsr::GRID_BLOCK), 0, stream, V, smpl_verts, workspace);
  gsr::LbsArgs a = {P, V, query, normals, smpl_verts, weights, lbs_offsets, A_big, A_pose, off_big, off_shape, off_pose, R, Th,
                    vert_ids, bweights, smpl_pts, world_pts, transforms, translation, world_normals, workspace, nullptr, nullptr,
                    nullptr, nullptr};
  hipLaunchKernelGGL(gsr::lbs_forward_kernel<true>, dim3((P + gsr::LBS_BLOCK - 1) / gsr::LBS_BLOCK), dim3(gsr::LBS_BLOCK), 0,
                     stream, a);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

size_t gsr_lbs_nn_cache_bytes(int P) { return gsr::nn_cache_bytes(P > 0 ? (size_t)P : 1); }

int gsr_lbs_forward_cached(int P, int V, const float *query, const float *normals, const float *smpl_verts, const float *weights,
                           const float *lbs_offsets, const float *A_big, const float *A_pose, const float *off_big,
                           const float *off_shape, const float *off_pose, const float *R, const float *Th, int *vert_ids,
                           float *bweights, float *smpl_pts, float *world_pts, float *transforms, float *translation,
                           float *world_normals, char *workspace, size_t workspace_bytes, char *nn_cache, size_t nn_cache_bytes,
                           int cache_is_valid, gsr_stream_t stream_) {
  if (P < 0 || V <= 0 || (P > 0 && (!query || !smpl_verts || !weights || !A_big || !A_pose || !off_big || !off_shape ||
                                    !off_pose || !R || !Th || !world_pts))) {
    gsr::set_error("gsr_lbs_forward_cached: bad arguments");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  if (!workspace || workspace_bytes < gsr::grid_workspace_bytes(V) || reinterpret_cast<size_t>(workspace) % 16 != 0) {
    gsr::set_error("gsr_lbs_forward_cached: grid workspace of %zu bytes (16-byte aligned, built by gsr_lbs_grid_build) required, got %zu",
                   gsr::grid_workspace_bytes(V), workspace_bytes);
    return GSR_EINVAL;
  }
  if (!nn_cache || nn_cache_bytes < gsr::nn_cache_bytes((size_t)P) || reinterpret_cast<size_t>(nn_cache) % 16 != 0) {
    gsr::set_error("gsr_lbs_forward_cached: cache buffer of %zu bytes (16-byte aligned) required, got %zu", gsr::nn_cache_bytes((size_t)P),
                   nn_cache_bytes);
    return GSR_EINVAL;
  }
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  gsr::NnCacheView c = gsr::nn_cache_view(nn_cache, (size_t)P);
  gsr::LbsArgs a = {P, V, query, normals, smpl_verts, weights, lbs_offsets, A_big, A_pose, off_big, off_shape, off_pose, R, Th,
                    vert_ids, bweights, smpl_pts, world_pts, transforms, translation, world_normals, workspace, nullptr, nullptr,
                    nullptr, nullptr};
  const dim3 grid((P + gsr::LBS_BLOCK - 1) / gsr::LBS_BLOCK), block(gsr::LBS_BLOCK);
  if (!cache_is_valid) {  // full search; the entries are made on the way
    GSR_HIP(gsr::zero_async(c.count, 64, stream));
    a.cache_entry = c.entry, a.cache_ids = c.ids;
    hipLaunchKernelGGL(gsr::lbs_forward_kernel<true>, grid, block, 0, stream, a);
  } else {
    hipLaunchKernelGGL(gsr::nn_cache_update_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, query, workspace, c);
    a.given_ids = c.ids, a.cache_count = c.count;
    hipLaunchKernelGGL(gsr::lbs_forward_kernel<true>, grid, block, 0, stream, a);
  }
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_lbs_backward(int P, int V, const float *query, const float *normals, const int *vert_ids, const float *weights,
                     const float *lbs_offsets, const float *A_big, const float *A_pose, const float *off_big,
                     const float *off_shape, const float *off_pose, const float *R, const float *dL_dworld_pts,
                     const float *dL_dtransforms, const float *dL_dworld_normals, float *dL_dquery, float *dL_dnormals,
                     float *dL_dlbs_offsets, float *dL_dA_pose, float *dL_doff_pose, float *dA_pose_partials,
                     gsr_stream_t stream_) {
  if (P < 0 || V <= 0 || (P > 0 && (!query || !vert_ids || !weights || !A_big || !A_pose || !off_big || !off_shape ||
                                    !off_pose || !R || !dL_dquery))) {
    gsr::set_error("gsr_lbs_backward: bad arguments");
    return GSR_EINVAL;
  }
  if (P == 0) return GSR_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  gsr::LbsBwdArgs a = {P, V, query, normals, weights, lbs_offsets, A_big, A_pose, off_big, off_shape, off_pose, R, vert_ids,
                       dL_dworld_pts, dL_dtransforms, dL_dworld_normals, dL_dquery, dL_dnormals, dL_dlbs_offsets, dL_dA_pose,
                       dL_doff_pose, dL_dA_pose ? dA_pose_partials : nullptr};
  hipLaunchKernelGGL(gsr::lbs_backward_kernel, dim3((P + gsr::LBS_BLOCK - 1) / gsr::LBS_BLOCK), dim3(gsr::LBS_BLOCK), 0, stream, a);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}
}
