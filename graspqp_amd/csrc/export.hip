// Export-time kinematics (reference scripts/fit.py:224-300 `export_poses`, every 500 iterations and at the end):
//
//   gq_link_jacobian      explicit geometric Jacobian of every mesh link, (B,L,6,J) = [J_v ; J_w] in the hand frame at the
//                         link-frame origin -- HandModel.jacobian (hand_model.py:772-777 -> the pytorch_kinematics fork's
//                         tree Chain.jacobian)
//   gq_contact_jacobian   linear Jacobian of the selected contact candidates, (B,n,3,J) = J_v + J_w x r with r = the
//                         candidate's offset from its link origin (hand_model.py:1176-1196); for a revolute joint j on the
//                         path to the contact this is a_j x (c - p_j), for a prismatic one a_j, otherwise 0
//   gq_joint_velocities   theta = pinv(J_flat) d with the reference's damped pseudo-inverse (hand_model.py:46-54,
//                         lambda = 1e-3): (J'J + lambda I) theta = J'd (the left form; the right form the reference takes
//                         for 3n < n_dofs is the same matrix by the push-through identity), residuals and end-effector
//                         velocities (hand_model.py:1198-1218)
//   gq_root_pose_wxyz     [t, unit quaternion (w,x,y,z)] of hand_pose[:, :9] (fit.py:260-263: Gram-Schmidt, then
//                         roma.rotmat_to_unitquat and the xyzw -> wxyz shuffle)
//
// These run a handful of times per run; they are written for clarity and exactness (the normal equations are formed and
// factored in fp64), not for throughput.  The same per-joint quantities (axis a_j, origin p_j in the hand frame) drive
// the analytic backward pass of kin.hip, which is the transposed form of this Jacobian.
#include "kin_dev.h"

struct GqJacArgs {
  gqHand h;
  const float* node_W;   // (B,J,12) from gq_fk_forward's workspace
  const float* link_T;   // (B,L,12)
  const int64_t* idx;    // (B,n) or null (link form)
  int B, n;
  float* out;            // (B,n,3,J) or (B,L,6,J)
};

// one wavefront per row; lane j = joint node j (J <= 64)
template <bool LINKS>
__global__ __launch_bounds__(GQ_WAVE) void gq_jacobian_kernel(GqJacArgs g) {
  __shared__ int s_parent[GQ_WAVE];
  __shared__ float s_col[GQ_WAVE * 6];  // coupled hands: tree-joint columns, folded into actuated columns (J_act = J_tree C)
  const gqHand& h = g.h;
  const int row = blockIdx.x, lane = gq_lane();
  gq3 a = gq_mk(0, 0, 0), p = gq_mk(0, 0, 0);
  int type = 0;
  if (lane < h.J) {
    const GqT W = gq_t_load(g.node_W + ((size_t)row * h.J + lane) * 12);
    a = gq_t_rot(W, gq_mk(h.node_axis[lane * 3], h.node_axis[lane * 3 + 1], h.node_axis[lane * 3 + 2]));
    p = gq_t_pos(W);
    type = h.node_type[lane];
    s_parent[lane] = h.node_parent[lane];
  }
  gq_wave_sync();
  const int count = LINKS ? h.L : g.n;
  for (int c = 0; c < count; ++c) {  // wave-uniform loop
    int l;
    gq3 x;  // the point whose velocity the Jacobian gives, hand frame
    if (LINKS) {
      l = c;
      x = gq_t_pos(gq_t_load(g.link_T + ((size_t)row * h.L + l) * 12));
    } else {
      const int ci = (int)g.idx[(size_t)row * g.n + c];
      l = h.cand_link[ci];
      x = gq_t_apply(gq_t_load(g.link_T + ((size_t)row * h.L + l) * 12),
                     gq_mk(h.cand_pos[ci * 3], h.cand_pos[ci * 3 + 1], h.cand_pos[ci * 3 + 2]));
    }
    // is joint `lane` on the path from the base to link l?  walk up from the link's node
    bool on_path = false;
    for (int nd = h.link_node[l]; nd >= 0; nd = s_parent[nd]) on_path |= (nd == lane);
    gq3 jv = gq_mk(0, 0, 0), jw = gq_mk(0, 0, 0);
    if (on_path && lane < h.J) {
      if (type == 1) {
        jv = gq_cross(a, x - p);
        jw = a;
      } else {
        jv = a;
      }
    }
    if (h.coup) {  // the reference's jacobian_fnc (hands/ability_hand.py:33-40, panda.py:17-26): columns combined by C
      gq_wave_sync();
      s_col[lane * 6 + 0] = jv.x; s_col[lane * 6 + 1] = jv.y; s_col[lane * 6 + 2] = jv.z;
      s_col[lane * 6 + 3] = jw.x; s_col[lane * 6 + 4] = jw.y; s_col[lane * 6 + 5] = jw.z;
      gq_wave_sync();
      float v[6] = {0, 0, 0, 0, 0, 0};
      if (lane < h.JA)
        for (int j = 0; j < h.J; ++j) {
          const float cj = h.coup[j * h.JA + lane];
#pragma unroll
          for (int k = 0; k < 6; ++k) v[k] = fmaf(cj, s_col[j * 6 + k], v[k]);
        }
      jv = gq_mk(v[0], v[1], v[2]);
      jw = gq_mk(v[3], v[4], v[5]);
    }
    if (lane < h.JA) {
      const int JA = h.JA;
      if (LINKS) {
        float* o = g.out + (((size_t)row * h.L + c) * 6) * JA + lane;
        o[0] = jv.x; o[JA] = jv.y; o[2 * JA] = jv.z;
        o[3 * JA] = jw.x; o[4 * JA] = jw.y; o[5 * JA] = jw.z;
      } else {
        float* o = g.out + (((size_t)row * g.n + c) * 3) * JA + lane;
        o[0] = jv.x; o[JA] = jv.y; o[2 * JA] = jv.z;
      }
    }
  }
}

// Backward of gq_contact_jacobian w.r.t. the joint angles: g_theta[k] = sum_{contact i, joint j} G_ij . dJ_ij/dtheta_k with the
// kinematic Hessian of the tree in closed form.  J_ij = a_j x (x_i - p_j) (revolute j on the path base -> contact i) or a_j
// (prismatic); for another joint k on that path
//   k above j (nearer the base):  everything below k turns about a_k (revolute k): dJ_ij = a_k x J_ij; a prismatic k shifts
//                                 p_j and x_i alike: 0
//   k == j, revolute:             only x_i turns: dJ_ij = a_j x (a_j x (x_i - p_j))
//   k below j (nearer the contact): only x_i moves, by a_k x (x_i - p_k) (revolute k) or a_k (prismatic k): dJ_ij = a_j x dx_i
//                                 for a revolute j, 0 for a prismatic one
// (what autograd through the pytorch_kinematics Jacobian computes in the reference, core/energy.py:80-87 via
// hand_model.py:1155-1218).  One wavefront per row, lane k = tree joint k; coupled hands: G's actuated columns are spread to the
// tree joints with C and the tree gradient is folded back with C' (J_act = J_tree C).
struct GqJacBwdArgs {
  gqHand h;
  const float* node_W;   // (B,J,12)
  const float* link_T;   // (B,L,12)
  const int64_t* idx;    // (B,n)
  const float* G;        // (B,n,3,JA)
  int B, n;
  float* g_theta;        // (B,JA)
};

__global__ __launch_bounds__(GQ_WAVE) void gq_contact_jacobian_bwd_kernel(GqJacBwdArgs g) {
  __shared__ int s_parent[GQ_WAVE], s_type[GQ_WAVE];
  __shared__ float s_a[GQ_WAVE * 3], s_p[GQ_WAVE * 3], s_G[GQ_WAVE * 3], s_acc[GQ_WAVE];
  const gqHand& h = g.h;
  const int row = blockIdx.x, lane = gq_lane();
  gq3 ak = gq_mk(0, 0, 0), pk = gq_mk(0, 0, 0);
  int tk = 0;
  if (lane < h.J) {
    const GqT W = gq_t_load(g.node_W + ((size_t)row * h.J + lane) * 12);
    ak = gq_t_rot(W, gq_mk(h.node_axis[lane * 3], h.node_axis[lane * 3 + 1], h.node_axis[lane * 3 + 2]));
    pk = gq_t_pos(W);
    tk = h.node_type[lane];
    s_parent[lane] = h.node_parent[lane];
    s_type[lane] = tk;
    s_a[lane * 3] = ak.x; s_a[lane * 3 + 1] = ak.y; s_a[lane * 3 + 2] = ak.z;
    s_p[lane * 3] = pk.x; s_p[lane * 3 + 1] = pk.y; s_p[lane * 3 + 2] = pk.z;
  }
  gq_wave_sync();
  float acc = 0.0f;
  for (int c = 0; c < g.n; ++c) {  // wave-uniform loop
    const int ci = (int)g.idx[(size_t)row * g.n + c];
    const int l = h.cand_link[ci];
    const gq3 x = gq_t_apply(gq_t_load(g.link_T + ((size_t)row * h.L + l) * 12),
                             gq_mk(h.cand_pos[ci * 3], h.cand_pos[ci * 3 + 1], h.cand_pos[ci * 3 + 2]));
    // G of this contact per TREE joint (lane j)
    gq3 Gj = gq_mk(0, 0, 0);
    if (lane < h.J) {
      const float* Gc = g.G + ((size_t)row * g.n + c) * 3 * h.JA;
      if (h.coup) {
        for (int a = 0; a < h.JA; ++a) {
          const float cj = h.coup[lane * h.JA + a];
          Gj.x = fmaf(cj, Gc[a], Gj.x);
          Gj.y = fmaf(cj, Gc[h.JA + a], Gj.y);
          Gj.z = fmaf(cj, Gc[2 * h.JA + a], Gj.z);
        }
      } else {
        Gj = gq_mk(Gc[lane], Gc[h.JA + lane], Gc[2 * h.JA + lane]);
      }
    }
    gq_wave_sync();
    s_G[lane * 3] = Gj.x; s_G[lane * 3 + 1] = Gj.y; s_G[lane * 3 + 2] = Gj.z;
    gq_wave_sync();
    bool on_path = false;
    for (int nd = h.link_node[l]; nd >= 0; nd = s_parent[nd]) on_path |= (nd == lane);
    if (on_path && lane < h.J) {
      bool passed = false;  // walking from the contact's link towards the base: have we passed joint `lane` yet?
      for (int j = h.link_node[l]; j >= 0; j = s_parent[j]) {
        const gq3 aj = gq_mk(s_a[j * 3], s_a[j * 3 + 1], s_a[j * 3 + 2]);
        const gq3 pj = gq_mk(s_p[j * 3], s_p[j * 3 + 1], s_p[j * 3 + 2]);
        const gq3 Gv = gq_mk(s_G[j * 3], s_G[j * 3 + 1], s_G[j * 3 + 2]);
        const bool jrev = s_type[j] == 1;
        gq3 dJ = gq_mk(0, 0, 0);
        if (j == lane) {
          passed = true;
          if (jrev) dJ = gq_cross(aj, gq_cross(aj, x - pj));
        } else if (!passed) {  // j lies between joint `lane` and the contact: `lane` is above j
          if (tk == 1) dJ = gq_cross(ak, jrev ? gq_cross(aj, x - pj) : aj);
        } else {               // j lies above joint `lane`
          if (jrev) dJ = gq_cross(aj, tk == 1 ? gq_cross(ak, x - pk) : ak);
        }
        acc += gq_dot(Gv, dJ);
      }
    }
  }
  if (h.coup) {
    gq_wave_sync();
    s_acc[lane] = lane < h.J ? acc : 0.0f;
    gq_wave_sync();
    if (lane < h.JA) {
      float v = 0.0f;
      for (int j = 0; j < h.J; ++j) v = fmaf(h.coup[j * h.JA + lane], s_acc[j], v);
      g.g_theta[(size_t)row * h.JA + lane] = v;
    }
  } else if (lane < h.JA) {
    g.g_theta[(size_t)row * h.JA + lane] = acc;
  }
}

// theta = (J'J + lambda I)^-1 J' d  for one row per wavefront; A and its Cholesky factor in fp64 (LDS)
#define GQ_JV_MAXJ 64
struct GqJvArgs {
  const float* Jc;    // (B,m,J)
  const float* dir;   // (B,m) world (with Rg) or hand frame
  const float* Rg;    // (B,9) or null
  int B, m, J;
  float damping;
  float* theta;       // (B,J)
  float* resid;       // (B,m) or null
  float* ee;          // (B,m) or null
};

__global__ __launch_bounds__(GQ_WAVE) void gq_joint_vel_kernel(GqJvArgs g) {
  extern __shared__ double gq_jv_lds[];
  const int J = g.J, m = g.m;
  double* A = gq_jv_lds;                 // J x J, row-major
  double* bvec = A + (size_t)J * J;      // J
  float* srow = reinterpret_cast<float*>(bvec + J);  // J floats: one row of Jc
  const int row = blockIdx.x, lane = gq_lane();
  const float* Jc = g.Jc + (size_t)row * m * J;
  const float* R = g.Rg ? g.Rg + (size_t)row * 9 : nullptr;
  for (int i = lane; i < J * J; i += GQ_WAVE) A[i] = 0.0;
  double bj = 0.0;
  gq_wave_sync();
  for (int r = 0; r < m; ++r) {
    // direction component r in the hand frame: d_h = R^T d_world per contact triple (hand_model.py:1166)
    float dr;
    {
      const int c3 = (r / 3) * 3, k = r % 3;
      const float* d = g.dir + (size_t)row * m + c3;
      dr = R ? (R[k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2]) : d[k];
    }
    const float v = lane < J ? Jc[(size_t)r * J + lane] : 0.0f;
    if (lane < J) srow[lane] = v;
    gq_wave_sync();
    if (lane < J) {
      bj += (double)v * (double)dr;
      for (int i = 0; i < J; ++i) A[(size_t)i * J + lane] += (double)srow[i] * (double)v;
    }
    gq_wave_sync();
  }
  if (lane < J) {
    A[(size_t)lane * J + lane] += (double)g.damping;
    bvec[lane] = bj;
  }
  gq_wave_sync();
  // Cholesky A = L L' (lower, in place), one column at a time; lane i owns row i
  for (int k = 0; k < J; ++k) {
    const double dkk = sqrt(A[(size_t)k * J + k]);
    gq_wave_sync();
    if (lane == k) A[(size_t)k * J + k] = dkk;
    if (lane > k && lane < J) A[(size_t)lane * J + k] /= dkk;
    gq_wave_sync();
    if (lane > k && lane < J) {
      const double lik = A[(size_t)lane * J + k];
      for (int j = k + 1; j <= lane; ++j) A[(size_t)lane * J + j] -= lik * A[(size_t)j * J + k];
    }
    gq_wave_sync();
  }
  // forward / backward substitution by lane 0 (J <= 64: a few thousand flops)
  if (lane == 0) {
    for (int i = 0; i < J; ++i) {
      double s = bvec[i];
      for (int k = 0; k < i; ++k) s -= A[(size_t)i * J + k] * bvec[k];
      bvec[i] = s / A[(size_t)i * J + i];
    }
    for (int i = J - 1; i >= 0; --i) {
      double s = bvec[i];
      for (int k = i + 1; k < J; ++k) s -= A[(size_t)k * J + i] * bvec[k];
      bvec[i] = s / A[(size_t)i * J + i];
    }
  }
  gq_wave_sync();
  if (lane < J) g.theta[(size_t)row * J + lane] = (float)bvec[lane];
  if (g.resid || g.ee) {
    // ee = J theta (hand frame); residual = (ee - d_h)^2; ee_vel is reported in the world frame (R ee per triple)
    for (int c3 = lane * 3; c3 < m; c3 += GQ_WAVE * 3) {
      double e[3] = {0, 0, 0};
      for (int k = 0; k < 3; ++k)
        for (int j = 0; j < J; ++j) e[k] += (double)Jc[(size_t)(c3 + k) * J + j] * bvec[j];
      const float* d = g.dir + (size_t)row * m + c3;
      for (int k = 0; k < 3; ++k) {
        const float dr = R ? (R[k] * d[0] + R[3 + k] * d[1] + R[6 + k] * d[2]) : d[k];
        const double q = e[k] - (double)dr;
        if (g.resid) g.resid[(size_t)row * m + c3 + k] = (float)(q * q);
      }
      if (g.ee) {
        float* o = g.ee + (size_t)row * m + c3;
        if (R) {
          for (int k = 0; k < 3; ++k) o[k] = (float)(R[3 * k] * e[0] + R[3 * k + 1] * e[1] + R[3 * k + 2] * e[2]);
        } else {
          for (int k = 0; k < 3; ++k) o[k] = (float)e[k];
        }
      }
    }
  }
}

// [t, q_wxyz]: Gram-Schmidt (utils/transforms.py:5-13), then the largest-of-(R00, R11, R22, trace) branch of
// roma.rotmat_to_unitquat (= scipy Rotation.from_matrix), normalised, NOT sign-canonicalised -- fit.py:260-263
__global__ __launch_bounds__(256) void gq_root_pose_kernel(const float* __restrict__ hand_pose, int B, int D,
                                                           float* __restrict__ out) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B) return;
  const float* hp = hand_pose + (size_t)row * D;
  float R[9];
  gq_rot6d(hp + 3, R);
  const float tr = R[0] + R[4] + R[8];
  const float dec[4] = {R[0], R[4], R[8], tr};
  int ch = 0;
  for (int i = 1; i < 4; ++i)
    if (dec[i] > dec[ch]) ch = i;  // first maximum, like argmax
  float q[4];  // x y z w
  if (ch != 3) {
    const int i = ch, j = (i + 1) % 3, k = (j + 1) % 3;
    q[i] = 1.0f - tr + 2.0f * R[i * 3 + i];
    q[j] = R[j * 3 + i] + R[i * 3 + j];
    q[k] = R[k * 3 + i] + R[i * 3 + k];
    q[3] = R[k * 3 + j] - R[j * 3 + k];
  } else {
    q[0] = R[2 * 3 + 1] - R[1 * 3 + 2];
    q[1] = R[0 * 3 + 2] - R[2 * 3 + 0];
    q[2] = R[1 * 3 + 0] - R[0 * 3 + 1];
    q[3] = 1.0f + tr;
  }
  const float inv = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float* o = out + (size_t)row * 7;
  o[0] = hp[0]; o[1] = hp[1]; o[2] = hp[2];
  o[3] = q[3] * inv; o[4] = q[0] * inv; o[5] = q[1] * inv; o[6] = q[2] * inv;
}

extern "C" {

int gq_link_jacobian(const gqHand* h, int64_t batch, const float* link_T, float* jac, const void* workspace,
                     size_t workspace_bytes, void* stream) {
  GQ_REQUIRE(h && link_T && jac && workspace, "link_jacobian: null pointer");
  GQ_REQUIRE(batch > 0 && h->J <= GQ_WAVE, "link_jacobian: bad sizes (B=%lld, J=%d > 64 unsupported)", (long long)batch, h->J);
  GQ_REQUIRE(workspace_bytes >= (size_t)batch * h->J * 12 * sizeof(float), "link_jacobian: not an FK workspace of this batch");
  GqJacArgs a{};
  a.h = *h;
  a.node_W = (const float*)workspace;
  a.link_T = link_T;
  a.B = (int)batch;
  a.out = jac;
  hipLaunchKernelGGL(gq_jacobian_kernel<true>, dim3((unsigned)batch), dim3(GQ_WAVE), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_contact_jacobian(const gqHand* h, const int64_t* contact_idx, int64_t batch, int n_contact, const float* link_T,
                        float* jac, const void* workspace, size_t workspace_bytes, void* stream) {
  GQ_REQUIRE(h && contact_idx && link_T && jac && workspace, "contact_jacobian: null pointer");
  GQ_REQUIRE(batch > 0 && n_contact > 0 && h->J <= GQ_WAVE, "contact_jacobian: bad sizes (B=%lld, n=%d, J=%d)",
             (long long)batch, n_contact, h->J);
  GQ_REQUIRE(workspace_bytes >= (size_t)batch * h->J * 12 * sizeof(float), "contact_jacobian: not an FK workspace of this batch");
  GqJacArgs a{};
  a.h = *h;
  a.node_W = (const float*)workspace;
  a.link_T = link_T;
  a.idx = contact_idx;
  a.B = (int)batch;
  a.n = n_contact;
  a.out = jac;
  hipLaunchKernelGGL(gq_jacobian_kernel<false>, dim3((unsigned)batch), dim3(GQ_WAVE), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_contact_jacobian_backward(const gqHand* h, const int64_t* contact_idx, int64_t batch, int n_contact, const float* link_T,
                                 const float* grad_jac, float* grad_theta, const void* workspace, size_t workspace_bytes,
                                 void* stream) {
  GQ_REQUIRE(h && contact_idx && link_T && grad_jac && grad_theta && workspace, "contact_jacobian_backward: null pointer");
  GQ_REQUIRE(batch > 0 && n_contact > 0 && h->J <= GQ_WAVE, "contact_jacobian_backward: bad sizes (B=%lld, n=%d, J=%d)",
             (long long)batch, n_contact, h->J);
  GQ_REQUIRE(workspace_bytes >= (size_t)batch * h->J * 12 * sizeof(float),
             "contact_jacobian_backward: not an FK workspace of this batch");
  GqJacBwdArgs a{};
  a.h = *h;
  a.node_W = (const float*)workspace;
  a.link_T = link_T;
  a.idx = contact_idx;
  a.G = grad_jac;
  a.B = (int)batch;
  a.n = n_contact;
  a.g_theta = grad_theta;
  hipLaunchKernelGGL(gq_contact_jacobian_bwd_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_joint_velocities(const float* jac, const float* directions, const float* Rg, int64_t batch, int m, int n_dofs,
                        float damping, float* theta, float* residual, float* ee_vel, void* stream) {
  GQ_REQUIRE(jac && directions && theta, "joint_velocities: null pointer");
  GQ_REQUIRE(batch > 0 && m > 0 && m % 3 == 0 && n_dofs > 0 && n_dofs <= GQ_JV_MAXJ && damping > 0.0f,
             "joint_velocities: bad sizes (B=%lld, m=%d, n_dofs=%d, damping=%g)", (long long)batch, m, n_dofs, (double)damping);
  GqJvArgs a{};
  a.Jc = jac;
  a.dir = directions;
  a.Rg = Rg;
  a.B = (int)batch;
  a.m = m;
  a.J = n_dofs;
  a.damping = damping;
  a.theta = theta;
  a.resid = residual;
  a.ee = ee_vel;
  const size_t lds = ((size_t)n_dofs * n_dofs + n_dofs) * sizeof(double) + (size_t)n_dofs * sizeof(float);
  hipLaunchKernelGGL(gq_joint_vel_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), lds, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_root_pose_wxyz(const float* hand_pose, int64_t batch, int pose_dim, float* root_pose, void* stream) {
  GQ_REQUIRE(hand_pose && root_pose && batch > 0 && pose_dim >= 9, "root_pose_wxyz: bad arguments");
  hipLaunchKernelGGL(gq_root_pose_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     hand_pose, (int)batch, pose_dim, root_pose);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
