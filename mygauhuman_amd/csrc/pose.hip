// pose.hip -- SMPL pose -> 24 joint transforms, forward and backward, ONE single-wave launch each way.
//
// Replaces the per-frame torch chain of the reference (scene/gaussian_model.py: batch_rodrigues_torch :894-912 /
// batch_rodrigues :982-1013, the pose-refinement product rot_mats[1:] @ correct_Rs :822-825, get_rigid_transformation_torch
// :914-944, get_transform_params_torch :947-980): ~75 tiny kernels forward (23 chained 4x4 matmuls among them) and about
// twice that in autograd's backward, i.e. milliseconds of launch overhead for ~20 kFLOP of work.
//   lane j < 24:  R_j = rodrigues(theta_j) (angle = |theta + 1e-8|, :989)  [ @ correct_Rs[j-1] for j >= 1 ]
//   lanes 0..11:  kinematic chain G_i = G_parent(i) * [R_i | J_i - J_parent(i)], joints in order (parents[i] < i)
//   lane j < 24:  A_j = [ G_j.R | G_j.t - G_j.R J_j ]   (rest pose removed)
// Backward = the exact adjoint of those three phases (what autograd computes for the reference chain), giving
// dL/dposes, dL/dcorrect_Rs and dL/djoints from dL/dA and the extra dL/drot_mats of the pose blend shapes.
#include "gsr_common.h"

namespace gsr {

constexpr int PJ = 24;

struct PoseArgs {
  const float *poses, *correct_Rs, *joints;
  int parents[PJ];
  float *rot_mats, *A;                                 // forward outputs
  const float *g_A, *g_rot;                            // backward inputs
  float *d_poses, *d_correct_Rs, *d_joints;            // backward outputs
};

struct Rodrigues {  // intermediates of one joint's axis-angle -> matrix map
  float angle, d[3], s, c, K[9], KK[9];
};

__device__ __forceinline__ void mat3_mm(const float *A, const float *B, float *o) {
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) o[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ void mat3_mm_nt(const float *A, const float *B, float *o) {  // A B^T
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) o[3 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
}
__device__ __forceinline__ void mat3_mm_tn(const float *A, const float *B, float *o) {  // A^T B
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) o[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}

__device__ __forceinline__ void rodrigues_fwd(const float *v, Rodrigues &q, float *R) {
  const float e0 = v[0] + 1e-8f, e1 = v[1] + 1e-8f, e2 = v[2] + 1e-8f;
  q.angle = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
#pragma unroll
  for (int k = 0; k < 3; k++) q.d[k] = v[k] / q.angle;
  q.s = sinf(q.angle);
  q.c = cosf(q.angle);
  const float K[9] = {0.f, -q.d[2], q.d[1], q.d[2], 0.f, -q.d[0], -q.d[1], q.d[0], 0.f};
#pragma unroll
  for (int k = 0; k < 9; k++) q.K[k] = K[k];
  mat3_mm(q.K, q.K, q.KK);
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = ((k % 4 == 0) ? 1.0f : 0.0f) + q.s * q.K[k] + (1.0f - q.c) * q.KK[k];
}

// dL/dv from dL/dR
__device__ __forceinline__ void rodrigues_bwd(const float *v, const Rodrigues &q, const float *dR, float *dv) {
  float gs = 0.f, gc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    gs += dR[k] * q.K[k];
    gc -= dR[k] * q.KK[k];
  }
  // dK = s dR + (1 - c)(dR K^T + K^T dR)
  float t1[9], t2[9], dK[9];
  mat3_mm_nt(dR, q.K, t1);
  mat3_mm_tn(q.K, dR, t2);
#pragma unroll
  for (int k = 0; k < 9; k++) dK[k] = q.s * dR[k] + (1.0f - q.c) * (t1[k] + t2[k]);
  const float dd[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
  float da = gs * q.c - gc * q.s;
  da -= (dd[0] * v[0] + dd[1] * v[1] + dd[2] * v[2]) / (q.angle * q.angle);
#pragma unroll
  for (int k = 0; k < 3; k++) dv[k] = dd[k] / q.angle + da * (v[k] + 1e-8f) / q.angle;
}

// shared by forward and backward: per-joint rotation, local transform and the chain, all in LDS
struct PoseLds {
  float R[PJ][9];      // final joint rotations (after the refinement product)
  float rel[PJ][3];    // J_i - J_parent
  float G[PJ][12];     // chained transform, rows 0..2 of the 4x4
  float dG[PJ][12];
  float dJ[PJ][3];
};

__device__ __forceinline__ void pose_forward_phases(const PoseArgs &a, PoseLds &s, Rodrigues &rq, float *Rod, int lane) {
  if (lane < PJ) {
    float v[3] = {a.poses[3 * lane], a.poses[3 * lane + 1], a.poses[3 * lane + 2]};
    rodrigues_fwd(v, rq, Rod);
    float R[9];
    if (a.correct_Rs && lane >= 1) {
      float C[9];
#pragma unroll
      for (int k = 0; k < 9; k++) C[k] = a.correct_Rs[9 * (lane - 1) + k];
      mat3_mm(Rod, C, R);
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = Rod[k];
    }
    const int p = a.parents[lane];
#pragma unroll
    for (int k = 0; k < 9; k++) s.R[lane][k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) s.rel[lane][k] = a.joints[3 * lane + k] - (lane >= 1 ? a.joints[3 * p + k] : 0.f);
  }
  __syncthreads();
  // chain: element (r, c) of G_i on lane 4r + c
  const int r = lane / 4, c = lane % 4;
  for (int i = 0; i < PJ; i++) {
    if (lane < 12) {
      const float tm_0c = c < 3 ? s.R[i][c] : s.rel[i][0];
      const float tm_1c = c < 3 ? s.R[i][3 + c] : s.rel[i][1];
      const float tm_2c = c < 3 ? s.R[i][6 + c] : s.rel[i][2];
      if (i == 0) {
        s.G[0][lane] = r == 0 ? tm_0c : (r == 1 ? tm_1c : tm_2c);
      } else {
        const float *Gp = s.G[a.parents[i]];
        s.G[i][lane] = Gp[4 * r] * tm_0c + Gp[4 * r + 1] * tm_1c + Gp[4 * r + 2] * tm_2c + (c == 3 ? Gp[4 * r + 3] : 0.f);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(WAVE) void smpl_pose_forward_kernel(const PoseArgs a) {
  __shared__ PoseLds s;
  const int lane = threadIdx.x;
  Rodrigues rq;
  float Rod[9];
  pose_forward_phases(a, s, rq, Rod, lane);
  if (lane < PJ) {
    const float *G = s.G[lane];
    const float J[3] = {a.joints[3 * lane], a.joints[3 * lane + 1], a.joints[3 * lane + 2]};
    float *o = a.A + 16 * lane;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      o[4 * r] = G[4 * r];
      o[4 * r + 1] = G[4 * r + 1];
      o[4 * r + 2] = G[4 * r + 2];
      o[4 * r + 3] = G[4 * r + 3] - (G[4 * r] * J[0] + G[4 * r + 1] * J[1] + G[4 * r + 2] * J[2]);
    }
    o[12] = o[13] = o[14] = 0.f;
    o[15] = 1.f;
    if (a.rot_mats)
#pragma unroll
      for (int k = 0; k < 9; k++) a.rot_mats[9 * lane + k] = s.R[lane][k];
  }
}

__global__ __launch_bounds__(WAVE) void smpl_pose_backward_kernel(const PoseArgs a) {
  __shared__ PoseLds s;
  __shared__ float s_dtm[PJ][12];  // gradient of the local transforms [R_i | rel_i]
  const int lane = threadIdx.x;
  Rodrigues rq;
  float Rod[9];
  pose_forward_phases(a, s, rq, Rod, lane);
  // adjoint of A_j = [G.R | G.t - G.R J]
  if (lane < PJ) {
    const float *g = a.g_A + 16 * lane;
    const float J[3] = {a.joints[3 * lane], a.joints[3 * lane + 1], a.joints[3 * lane + 2]};
    float dJ[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float gt = g[4 * r + 3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        s.dG[lane][4 * r + c] = g[4 * r + c] - gt * J[c];
        dJ[c] -= gt * s.G[lane][4 * r + c];
      }
      s.dG[lane][4 * r + 3] = gt;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) s.dJ[lane][c] = dJ[c];
  }
  __syncthreads();
  // adjoint of the chain, children before parents (parents[i] < i)
  const int r = lane / 4, c = lane % 4;
  for (int i = PJ - 1; i >= 1; i--) {
    const int p = a.parents[i];
    float add = 0.f, dtm = 0.f;
    if (lane < 12) {
      const float *dGi = s.dG[i];
      // dG_p[r][k] += sum_c dG_i[r][c] tm_i[k][c]   (k = c here; tm row 3 = (0,0,0,1))
      if (c < 3)
        add = dGi[4 * r] * s.R[i][3 * c] + dGi[4 * r + 1] * s.R[i][3 * c + 1] + dGi[4 * r + 2] * s.R[i][3 * c + 2] +
              dGi[4 * r + 3] * s.rel[i][c];
      else
        add = dGi[4 * r + 3];
      // dtm_i[k][c'] = sum_r G_p[r][k] dG_i[r][c']   (k = r here as the row index of tm)
      const float *Gp = s.G[p];
      dtm = Gp[r] * dGi[c] + Gp[4 + r] * dGi[4 + c] + Gp[8 + r] * dGi[8 + c];
    }
    __syncthreads();
    if (lane < 12) {
      s.dG[p][lane] += add;
      s_dtm[i][lane] = dtm;
    }
    __syncthreads();
  }
  if (lane < 12) s_dtm[0][lane] = s.dG[0][lane];
  __syncthreads();
  // joints: rel_i = J_i - J_parent(i); sequential scatter by one lane (24 x 3 adds)
  if (lane == 0) {
    for (int i = 0; i < PJ; i++)
      for (int k = 0; k < 3; k++) {
        const float g = s_dtm[i][4 * k + 3];
        s.dJ[i][k] += g;
        if (i >= 1) s.dJ[a.parents[i]][k] -= g;
      }
  }
  __syncthreads();
  if (lane < PJ) {
    float dR[9];
#pragma unroll
    for (int k = 0; k < 9; k++) dR[k] = s_dtm[lane][4 * (k / 3) + (k % 3)] + (a.g_rot ? a.g_rot[9 * lane + k] : 0.f);
    float dRod[9];
    if (a.correct_Rs && lane >= 1) {
      float C[9], dC[9];
#pragma unroll
      for (int k = 0; k < 9; k++) C[k] = a.correct_Rs[9 * (lane - 1) + k];
      mat3_mm_nt(dR, C, dRod);   // R = Rod C  ->  dRod = dR C^T, dC = Rod^T dR
      mat3_mm_tn(Rod, dR, dC);
      if (a.d_correct_Rs)
#pragma unroll
        for (int k = 0; k < 9; k++) a.d_correct_Rs[9 * (lane - 1) + k] = dC[k];
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) dRod[k] = dR[k];
    }
    const float v[3] = {a.poses[3 * lane], a.poses[3 * lane + 1], a.poses[3 * lane + 2]};
    float dv[3];
    rodrigues_bwd(v, rq, dRod, dv);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (a.d_poses) a.d_poses[3 * lane + k] = dv[k];
      if (a.d_joints) a.d_joints[3 * lane + k] = s.dJ[lane][k];
    }
  }
}

static int check_parents(const char *who, const int *parents) {
  if (!parents) {
    set_error("%s: parents (host array of 24 ints) is required", who);
    return GSR_EINVAL;
  }
  for (int i = 1; i < PJ; i++)
    if (parents[i] < 0 || parents[i] >= i) {
      set_error("%s: parents[%d] = %d; every joint's parent must precede it", who, i, parents[i]);
      return GSR_EINVAL;
    }
  return GSR_OK;
}

}  // namespace gsr

extern "C" {

int gsr_smpl_pose_forward(const float *poses, const float *correct_Rs, const float *joints, const int *parents_host,
                          float *rot_mats, float *A, gsr_stream_t stream_) {
  using namespace gsr;
  if (!poses || !joints || !A) {
    set_error("gsr_smpl_pose_forward: null argument");
    return GSR_EINVAL;
  }
  int rc = check_parents("gsr_smpl_pose_forward", parents_host);
  if (rc != GSR_OK) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  PoseArgs a = {};
  a.poses = poses, a.correct_Rs = correct_Rs, a.joints = joints, a.rot_mats = rot_mats, a.A = A;
  for (int i = 0; i < PJ; i++) a.parents[i] = i == 0 ? 0 : parents_host[i];
  hipLaunchKernelGGL(smpl_pose_forward_kernel, dim3(1), dim3(WAVE), 0, stream, a);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

int gsr_smpl_pose_backward(const float *poses, const float *correct_Rs, const float *joints, const int *parents_host,
                           const float *dL_dA, const float *dL_drot_mats, float *dL_dposes, float *dL_dcorrect_Rs,
                           float *dL_djoints, gsr_stream_t stream_) {
  using namespace gsr;
  if (!poses || !joints || !dL_dA) {
    set_error("gsr_smpl_pose_backward: null argument");
    return GSR_EINVAL;
  }
  int rc = check_parents("gsr_smpl_pose_backward", parents_host);
  if (rc != GSR_OK) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  PoseArgs a = {};
  a.poses = poses, a.correct_Rs = correct_Rs, a.joints = joints;
  a.g_A = dL_dA, a.g_rot = dL_drot_mats;
  a.d_poses = dL_dposes, a.d_correct_Rs = dL_dcorrect_Rs, a.d_joints = dL_djoints;
  for (int i = 0; i < PJ; i++) a.parents[i] = i == 0 ? 0 : parents_host[i];
  hipLaunchKernelGGL(smpl_pose_backward_kernel, dim3(1), dim3(WAVE), 0, stream, a);
  GSR_LAUNCH_CHECK(stream, 0);
  return GSR_OK;
}

}  // extern "C"
