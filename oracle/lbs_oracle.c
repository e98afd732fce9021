/*
 * lbs_oracle.c -- CPU ORACLE for the SMPL linear-blend-skinning part of the
 * hot path.  TEST INFRASTRUCTURE ONLY (see gsr_oracle.c header for the rules).
 *
 * Restates, in plain C / fp32:
 *   batch_rodrigues                  scene/gaussian_model.py:982-1013, smplx/lbs.py:299-333
 *   get_rigid_transformation_torch   scene/gaussian_model.py:914-944
 *   get_transform_params_torch       scene/gaussian_model.py:947-980
 *   coarse_deform_c2source           scene/gaussian_model.py:768-872
 *   smplx lbs()                      smplx/lbs.py:156-252 (+ :255-296, :349-405)
 *   geom_transform_points            utils/graphics_utils.py:22-29
 *
 * Pinned by tests/golden/lbs_*.npz (generated from the imported reference
 * smplx.lbs / graphics_utils, see tests/golden/make_golden.py).  The
 * coarse_deform_c2source restatement is pinned only through those shared
 * helpers (scene.gaussian_model is not importable: knn_cuda, cv2 ... absent).
 *
 * Layouts (all float32, C order): v_template [V][3]; shapedirs [V][3][NB];
 * posedirs_vk [V*3][NP] (gaussian_model.py layout) or posedirs_kv [NP][V*3]
 * (smplx layout); J_regressor [J][V]; weights [V][J]; parents int32[J];
 * 4x4 / 3x3 matrices row-major.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NJ 24

void oracle_rodrigues(int N, const float *rv, float *R) {
  for (int i = 0; i < N; i++) {
    float a0 = rv[3 * i] + 1e-8f, a1 = rv[3 * i + 1] + 1e-8f, a2 = rv[3 * i + 2] + 1e-8f;
    float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    float rx = rv[3 * i] / angle, ry = rv[3 * i + 1] / angle, rz = rv[3 * i + 2] / angle;
    float c = cosf(angle), s = sinf(angle);
    float K[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0}, KK[9];
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) KK[3 * r + cc] = K[3 * r] * K[cc] + K[3 * r + 1] * K[3 + cc] + K[3 * r + 2] * K[6 + cc];
    for (int k = 0; k < 9; k++) R[9 * i + k] = ((k % 4 == 0) ? 1.f : 0.f) + s * K[k] + (1 - c) * KK[k];
  }
}

static void mat4_mul(const float *a, const float *b, float *o) {
  float t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      float s = 0;
      for (int k = 0; k < 4; k++) s += a[4 * r + k] * b[4 * k + c];
      t[4 * r + c] = s;
    }
  memcpy(o, t, sizeof(t));
}

static void inv3(const float *m, float *o) {
  float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  float id = 1.0f / det;
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
void oracle_inv3(const float *m, float *o) { inv3(m, o); }

/* v_shaped = v_template + shapedirs . betas ; joints = J_regressor . v_shaped */
static void shaped_and_joints(int V, int NB, const float *v_template, const float *shapedirs, const float *J_regressor,
                              const float *betas, float *v_shaped, float *joints) {
  for (int v = 0; v < V; v++)
    for (int k = 0; k < 3; k++) {
      float s = 0;
      for (int l = 0; l < NB; l++) s += shapedirs[((size_t)v * 3 + k) * NB + l] * betas[l];
      v_shaped[3 * v + k] = v_template[3 * v + k] + s;
    }
  for (int j = 0; j < NJ; j++)
    for (int k = 0; k < 3; k++) {
      double s = 0; /* 6890-term dot product: accumulate wide, round once */
      for (int v = 0; v < V; v++) s += (double)J_regressor[(size_t)j * V + v] * v_shaped[3 * v + k];
      joints[3 * j + k] = (float)s;
    }
}

/* kinematic chain + rest-pose removal (gaussian_model.py:914-944, lbs.py:349-405).
 * posed_joints (may be NULL) = translation column of the chained transforms
 * before the rest-pose subtraction (lbs.py:398). */
static void rigid_chain(const float *rot_mats, const float *joints, const int *parents, float *A, float *posed_joints) {
  float tm[NJ][16], chain[NJ][16];
  for (int j = 0; j < NJ; j++) {
    float rel[3];
    for (int k = 0; k < 3; k++) rel[k] = joints[3 * j + k] - (j > 0 ? joints[3 * parents[j] + k] : 0.f);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) tm[j][4 * r + c] = rot_mats[9 * j + 3 * r + c];
      tm[j][4 * r + 3] = rel[r];
    }
    tm[j][12] = tm[j][13] = tm[j][14] = 0.f;
    tm[j][15] = 1.f;
  }
  memcpy(chain[0], tm[0], sizeof(tm[0]));
  for (int j = 1; j < NJ; j++) mat4_mul(chain[parents[j]], tm[j], chain[j]);
  for (int j = 0; j < NJ; j++) {
    if (posed_joints)
      for (int k = 0; k < 3; k++) posed_joints[3 * j + k] = chain[j][4 * k + 3];
    memcpy(A + 16 * j, chain[j], sizeof(chain[j]));
    for (int r = 0; r < 4; r++) {
      float rel = chain[j][4 * r] * joints[3 * j] + chain[j][4 * r + 1] * joints[3 * j + 1] + chain[j][4 * r + 2] * joints[3 * j + 2];
      A[16 * j + 4 * r + 3] = chain[j][4 * r + 3] - rel;
    }
  }
}

/* get_transform_params_torch with rot_mats given (gaussian_model.py:947-980) */
void oracle_joint_transforms(int V, int NB, const float *v_template, const float *shapedirs, const float *J_regressor,
                             const int *parents, const float *betas, const float *rot_mats, float *A, float *joints) {
  float *v_shaped = (float *)malloc(sizeof(float) * 3 * (size_t)V);
  shaped_and_joints(V, NB, v_template, shapedirs, J_regressor, betas, v_shaped, joints);
  rigid_chain(rot_mats, joints, parents, A, NULL);
  free(v_shaped);
}

/* pose blend-shape offsets: (R[1:] - I).flatten() [NP=207] . posedirs -> [V][3] */
void oracle_pose_offsets(int V, const float *posedirs_vk, const float *rot_mats, float *out) {
  float pf[(NJ - 1) * 9];
  for (int j = 1; j < NJ; j++)
    for (int k = 0; k < 9; k++) pf[(j - 1) * 9 + k] = rot_mats[9 * j + k] - ((k % 4 == 0) ? 1.f : 0.f);
  const int NP = (NJ - 1) * 9;
  for (size_t e = 0; e < (size_t)V * 3; e++) {
    float s = 0;
    for (int k = 0; k < NP; k++) s += pf[k] * posedirs_vk[e * NP + k];
    out[e] = s;
  }
}
void oracle_shape_offsets(int V, int NB, const float *shapedirs, const float *betas, float *out) {
  for (size_t e = 0; e < (size_t)V * 3; e++) {
    float s = 0;
    for (int l = 0; l < NB; l++) s += shapedirs[e * NB + l] * betas[l];
    out[e] = s;
  }
}

/* Per-point part of coarse_deform_c2source (gaussian_model.py:776-872), B = 1.
 * off_big / off_shape / off_pose are the per-vertex [V][3] tables
 * PoseOff(theta_big), ShapeOff(beta), PoseOff(theta, dR); R is params['R'] and
 * Th params['Th']. lbs_off may be NULL (no learned weight offsets). */
void oracle_lbs_deform(int P, const float *query, const float *normals, const int *vert_ids, const float *weights,
                       const float *lbs_off, const float *A_big, const float *A_pose, const float *off_big,
                       const float *off_shape, const float *off_pose, const float *R, const float *Th,
                       float *smpl_src, float *world_src, float *bweights_out, float *transforms, float *translation,
                       float *world_normals) {
  float Rinv[9];
  inv3(R, Rinv);
  for (int p = 0; p < P; p++) {
    int v = vert_ids[p];
    float bw[NJ];
    for (int j = 0; j < NJ; j++) bw[j] = weights[(size_t)v * NJ + j];
    if (lbs_off) { /* softmax(log(w + 1e-9) + off) */
      float mx = -INFINITY, e[NJ], sum = 0;
      for (int j = 0; j < NJ; j++) {
        e[j] = logf(bw[j] + 1e-9f) + lbs_off[(size_t)p * NJ + j];
        mx = fmaxf(mx, e[j]);
      }
      for (int j = 0; j < NJ; j++) {
        e[j] = expf(e[j] - mx);
        sum += e[j];
      }
      for (int j = 0; j < NJ; j++) bw[j] = e[j] / sum;
    }
    if (bweights_out) memcpy(bweights_out + (size_t)p * NJ, bw, sizeof(bw));
    float Ab[16], Ap[16];
    for (int k = 0; k < 16; k++) {
      float s0 = 0, s1 = 0;
      for (int j = 0; j < NJ; j++) {
        s0 += bw[j] * A_big[16 * j + k];
        s1 += bw[j] * A_pose[16 * j + k];
      }
      Ab[k] = s0;
      Ap[k] = s1;
    }
    float Rb[9] = {Ab[0], Ab[1], Ab[2], Ab[4], Ab[5], Ab[6], Ab[8], Ab[9], Ab[10]}, Ri[9];
    inv3(Rb, Ri);
    float q0[3] = {query[3 * p] - Ab[3], query[3 * p + 1] - Ab[7], query[3 * p + 2] - Ab[11]};
    float q[3], n[3], tr[3], t0[3] = {-Ab[3], -Ab[7], -Ab[11]};
    for (int r = 0; r < 3; r++) {
      q[r] = Ri[3 * r] * q0[0] + Ri[3 * r + 1] * q0[1] + Ri[3 * r + 2] * q0[2];
      n[r] = normals ? Ri[3 * r] * normals[3 * p] + Ri[3 * r + 1] * normals[3 * p + 1] + Ri[3 * r + 2] * normals[3 * p + 2] : 0.f;
      tr[r] = Ri[3 * r] * t0[0] + Ri[3 * r + 1] * t0[1] + Ri[3 * r + 2] * t0[2];
    }
    for (int k = 0; k < 3; k++) {
      float ob = off_big[3 * v + k], os = off_shape[3 * v + k], op = off_pose[3 * v + k];
      q[k] = q[k] - ob;
      q[k] = q[k] + os;
      q[k] = q[k] + op;
      tr[k] = tr[k] - ob;
      tr[k] = tr[k] + os;
      tr[k] = tr[k] + op;
    }
    float Rp[9] = {Ap[0], Ap[1], Ap[2], Ap[4], Ap[5], Ap[6], Ap[8], Ap[9], Ap[10]};
    float tp[3] = {Ap[3], Ap[7], Ap[11]};
    float can[3], sn[3], tr2[3], M1[9];
    for (int r = 0; r < 3; r++) {
      can[r] = Rp[3 * r] * q[0] + Rp[3 * r + 1] * q[1] + Rp[3 * r + 2] * q[2];
      sn[r] = Rp[3 * r] * n[0] + Rp[3 * r + 1] * n[1] + Rp[3 * r + 2] * n[2];
      tr2[r] = Rp[3 * r] * tr[0] + Rp[3 * r + 1] * tr[1] + Rp[3 * r + 2] * tr[2] + tp[r];
      for (int c = 0; c < 3; c++) M1[3 * r + c] = Rp[3 * r] * Ri[c] + Rp[3 * r + 1] * Ri[3 + c] + Rp[3 * r + 2] * Ri[6 + c];
    }
    float src[3] = {can[0] + tp[0], can[1] + tp[1], can[2] + tp[2]};
    for (int k = 0; k < 3; k++) {
      if (smpl_src) smpl_src[3 * p + k] = src[k];
      /* row-vector times R_inv (gaussian_model.py:864-870) */
      world_src[3 * p + k] = src[0] * Rinv[k] + src[1] * Rinv[3 + k] + src[2] * Rinv[6 + k] + Th[k];
      if (world_normals) world_normals[3 * p + k] = sn[0] * Rinv[k] + sn[1] * Rinv[3 + k] + sn[2] * Rinv[6 + k];
      if (translation) translation[3 * p + k] = tr2[0] * Rinv[k] + tr2[1] * Rinv[3 + k] + tr2[2] * Rinv[6 + k] + Th[k];
    }
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        transforms[9 * p + 3 * r + c] = R[3 * r] * M1[c] + R[3 * r + 1] * M1[3 + c] + R[3 * r + 2] * M1[6 + c];
  }
}

/* smplx/lbs.py:156-252, B = 1.  posedirs_kv is [NP][V*3] (smplx layout).
 * Outputs verts [V][3], J_transformed [J][3], A [J][16], T [V][16]. */
void oracle_smpl_lbs(int V, int NB, const float *betas, const float *pose /*[J*3]*/, const float *v_template,
                     const float *shapedirs, const float *posedirs_kv, const float *J_regressor, const int *parents,
                     const float *lbs_weights, float *verts, float *J_transformed, float *A, float *T) {
  float *v_shaped = (float *)malloc(sizeof(float) * 3 * (size_t)V);
  float joints[NJ * 3], rot[NJ * 9];
  shaped_and_joints(V, NB, v_template, shapedirs, J_regressor, betas, v_shaped, joints);
  oracle_rodrigues(NJ, pose, rot);
  float pf[(NJ - 1) * 9];
  for (int j = 1; j < NJ; j++)
    for (int k = 0; k < 9; k++) pf[(j - 1) * 9 + k] = rot[9 * j + k] - ((k % 4 == 0) ? 1.f : 0.f);
  const int NP = (NJ - 1) * 9;
  rigid_chain(rot, joints, parents, A, J_transformed);
  for (int v = 0; v < V; v++) {
    float vp[3];
    for (int k = 0; k < 3; k++) {
      float s = 0;
      for (int l = 0; l < NP; l++) s += pf[l] * posedirs_kv[(size_t)l * V * 3 + 3 * v + k];
      vp[k] = s + v_shaped[3 * v + k];
    }
    float Tv[16];
    for (int k = 0; k < 16; k++) {
      float s = 0;
      for (int j = 0; j < NJ; j++) s += lbs_weights[(size_t)v * NJ + j] * A[16 * j + k];
      Tv[k] = s;
    }
    if (T) memcpy(T + 16 * (size_t)v, Tv, sizeof(Tv));
    for (int r = 0; r < 3; r++) verts[3 * v + r] = Tv[4 * r] * vp[0] + Tv[4 * r + 1] * vp[1] + Tv[4 * r + 2] * vp[2] + Tv[4 * r + 3];
  }
  free(v_shaped);
}

/* utils/graphics_utils.py:22-29: points_hom @ M, divide by (w + 1e-7) */
void oracle_project(int P, const float *pts, const float *M, float *out) {
  for (int i = 0; i < P; i++) {
    const float *p = pts + 3 * i;
    float o[4];
    for (int c = 0; c < 4; c++) o[c] = p[0] * M[c] + p[1] * M[4 + c] + p[2] * M[8 + c] + M[12 + c];
    float d = o[3] + 0.0000001f;
    out[3 * i] = o[0] / d;
    out[3 * i + 1] = o[1] / d;
    out[3 * i + 2] = o[2] / d;
  }
}
