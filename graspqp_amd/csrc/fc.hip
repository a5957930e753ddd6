// Force-closure energy of GraspQP as fused kernels around the batched box-QP:
//   friction-cone edges + grasp matrix F (reference metrics/ops/span.py:263-295, 341-346),
//   E_fc = 2 (1/2 |F x|^2 + 0.01) exp(-gain (prod sigma_i(F))^(1/6))  (span.py:402, registry.py:82-87),
// and its gradient with respect to the contact points (contact normals come from the object SDF and are
// constants for autograd, object_model.py:246).
#include "fc_dev.h"

int gq_lsq_boxqp_iterate_(const float* A, float lower_s, float upper_s, int64_t batch, int m, int nz, float ridge,
                          float eps, int max_iter, int32_t* n_iter, void* workspace, size_t workspace_bytes,
                          void* stream, const float** resid, const float** snap, const int** kstar);
int gq_lsq_boxqp_backward_scaled_(const float* A, const float* lam, const float* slack, const float* grad_x,
                                  int64_t batch, int m, int nz, float ridge, float* dx, float* dlam,
                                  const float* scale_ge, const float* scale_svd, float svd_gain, float values_gain,
                                  void* stream);

__global__ void gq_grasp_matrix_kernel(const float* __restrict__ cpts, const float* __restrict__ cnrm,
                                       const float* __restrict__ cog, int B, int n, int k, float mu, float tw,
                                       float* __restrict__ F) {
  const int nz = n * k;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)B * nz) return;
  const int row = (int)(t / nz), i = (int)(t % nz);
  const GqCone c = gq_cone_column(cpts + (size_t)row * n * 3, cnrm + (size_t)row * n * 3, cog + (size_t)row * 3, i / k,
                                  i % k, k, mu, tw);
  float* o = F + (size_t)row * 6 * nz + i;
  o[0] = c.f.x;
  o[nz] = c.f.y;
  o[2 * nz] = c.f.z;
  o[3 * nz] = c.tau.x;
  o[4 * nz] = c.tau.y;
  o[5 * nz] = c.tau.z;
}

struct GqFcArgs {
  const float* F;     // (B,6,nz)
  float* x;           // (B,nz)   written here from the best PDIPM snapshot (select fused in)
  float* lam;         // (B,2nz)
  float* slack;       // (B,2nz)
  const float* resid; // (B,max_iter)
  const float* snap;  // (B,max_iter,5,nz)
  const int* kstar;
  int max_iter;
  int B, n, k;
  float svd_gain, values_gain, eps_add;
  float* e_fc;   // (B)
  float* val;    // (B)
  float* svd;    // (B)
  float* Ftr;    // (B,nz)  f_i . (F x)
  float* x_sum;  // (B,n) or null
};

__global__ __launch_bounds__(GQ_WAVE) void gq_fc_energy_kernel(GqFcArgs g) {
  const int row = blockIdx.x, lane = gq_lane();
  const int nz = g.n * g.k;
  // best iterate of this row among iterations 0..k* (qpth returns the per-row best, not the last)
  int bi = 0;
  {
    const int ks = g.kstar[0];
    float bst = 0.0f;
    for (int it = 0; it <= ks; ++it) {
      const float rs = g.resid[(size_t)row * g.max_iter + it];
      if (it == 0 || rs < bst) {
        bst = rs;
        bi = it;
      }
    }
  }
  const float* sn = g.snap + (((size_t)row * g.max_iter + bi) * 5) * nz;
  double gr[21];
#pragma unroll
  for (int i = 0; i < 21; ++i) gr[i] = 0.0;
  float r[6] = {0, 0, 0, 0, 0, 0};
  // nz may exceed 64: lanes stride over the columns
  for (int i = lane; i < nz; i += GQ_WAVE) {
    float f[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) f[q] = g.F[((size_t)row * 6 + q) * nz + i];
    const float xi = sn[i];
    g.x[(size_t)row * nz + i] = xi;
    g.lam[(size_t)row * 2 * nz + i] = sn[nz + i];
    g.lam[(size_t)row * 2 * nz + nz + i] = sn[2 * nz + i];
    g.slack[(size_t)row * 2 * nz + i] = sn[3 * nz + i];
    g.slack[(size_t)row * 2 * nz + nz + i] = sn[4 * nz + i];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      r[a] = fmaf(f[a], xi, r[a]);
#pragma unroll
      for (int b = 0; b <= a; ++b) gr[a * (a + 1) / 2 + b] += (double)f[a] * (double)f[b];
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) r[a] = gq_dpp_sum(r[a]);
#pragma unroll
  for (int i = 0; i < 21; ++i) gr[i] = gq_dpp_sum_d(gr[i]);
  double Lm[21], inv[6];
  const bool ok = gq_chol6(gr, Lm, inv);
  double lp = 1.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) lp *= Lm[i * (i + 1) / 2 + i];
  const float svd = ok ? powf((float)lp, 1.0f / 6.0f) : 0.0f;  // (prod sigma)^(1/6) = det(F F')^(1/12)
  float val = 0.0f;
#pragma unroll
  for (int a = 0; a < 6; ++a) val = fmaf(r[a], r[a], val);
  val *= 0.5f;
  for (int i = lane; i < nz; i += GQ_WAVE) {
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 6; ++q) acc = fmaf(g.F[((size_t)row * 6 + q) * nz + i], r[q], acc);
    g.Ftr[(size_t)row * nz + i] = acc;
  }
  if (g.x_sum) {
    for (int c = lane; c < g.n; c += GQ_WAVE) {
      float s = 0.0f;
      for (int e = 0; e < g.k; ++e) s += sn[c * g.k + e];
      g.x_sum[(size_t)row * g.n + c] = s;
    }
  }
  if (lane == 0) {
    g.val[row] = val;
    g.svd[row] = svd;
    g.e_fc[row] = g.values_gain * (val + g.eps_add) * expf(-g.svd_gain * svd);
  }
}

struct GqFcBwdArgs {
  const float* F;
  const float* x;
  const float* dx;  // (B,nz) from the QP backward
  const float* cpts;
  const float* cnrm;
  const float* cog;
  const float* g_e;
  const float* val;
  const float* svd;
  int B, n, k;
  float mu, tw, svd_gain, values_gain, eps_add;
  int accumulate;  // add into g_cpts instead of overwriting
  float* g_cpts;  // (B,n,3)
};

__global__ __launch_bounds__(GQ_WAVE) void gq_fc_grad_kernel(GqFcBwdArgs g) {
  extern __shared__ float sh[];  // nz*3 per-column contributions
  const int row = blockIdx.x, lane = gq_lane();
  const int nz = g.n * g.k;
  double gr[21];
#pragma unroll
  for (int i = 0; i < 21; ++i) gr[i] = 0.0;
  float r[6] = {0, 0, 0, 0, 0, 0}, fd[6] = {0, 0, 0, 0, 0, 0};
  for (int i = lane; i < nz; i += GQ_WAVE) {
    float f[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) f[q] = g.F[((size_t)row * 6 + q) * nz + i];
    const float xi = g.x[(size_t)row * nz + i], di = g.dx[(size_t)row * nz + i];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      r[a] = fmaf(f[a], xi, r[a]);
      fd[a] = fmaf(f[a], di, fd[a]);
#pragma unroll
      for (int b = 0; b <= a; ++b) gr[a * (a + 1) / 2 + b] += (double)f[a] * (double)f[b];
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    r[a] = gq_dpp_sum(r[a]);
    fd[a] = gq_dpp_sum(fd[a]);
  }
#pragma unroll
  for (int i = 0; i < 21; ++i) gr[i] = gq_dpp_sum_d(gr[i]);
  double Lm[21], inv[6];
  const bool ok = gq_chol6(gr, Lm, inv);
  const float ge = g.g_e[row], svd = g.svd[row], val = g.val[row];
  const float ex = expf(-g.svd_gain * svd);
  const float gval = ge * g.values_gain * ex;
  const float gsvd = ok ? ge * g.values_gain * (val + g.eps_add) * ex * (-g.svd_gain) : 0.0f;
  const float* cp = g.cpts + (size_t)row * g.n * 3;
  const float* cn = g.cnrm + (size_t)row * g.n * 3;
  const float* cog = g.cog + (size_t)row * 3;
  for (int i = lane; i < nz; i += GQ_WAVE) {
    const GqCone c = gq_cone_column(cp, cn, cog, i / g.k, i % g.k, g.k, g.mu, g.tw);
    const float xi = g.x[(size_t)row * nz + i], di = g.dx[(size_t)row * nz + i];
    // d(prod sigma^(1/6))/dF = svd/6 * (F F')^-1 F : solve L L' w = f_i
    double w[6] = {c.f.x, c.f.y, c.f.z, c.tau.x, c.tau.y, c.tau.z};
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int t = 0; t < a; ++t) w[a] -= Lm[a * (a + 1) / 2 + t] * w[t];
      w[a] *= inv[a];
    }
#pragma unroll
    for (int a = 5; a >= 0; --a) {
#pragma unroll
      for (int t = a + 1; t < 6; ++t) w[a] -= Lm[t * (t + 1) / 2 + a] * w[t];
      w[a] *= inv[a];
    }
    // gradient wrt the torque rows of column i (rows 3..5)
    gq3 gt;
    const float s6 = gsvd * svd / 6.0f;
    gt.x = (gval * r[3] + fd[3]) * xi + r[3] * di + s6 * (float)w[3];
    gt.y = (gval * r[4] + fd[4]) * xi + r[4] * di + s6 * (float)w[4];
    gt.z = (gval * r[5] + fd[5]) * xi + r[5] * di + s6 * (float)w[5];
    // tau = tw (r x f)  ->  d/dr = tw (f x g_tau)
    const gq3 gp = g.tw * gq_cross(c.f, gt);
    sh[i * 3] = gp.x;
    sh[i * 3 + 1] = gp.y;
    sh[i * 3 + 2] = gp.z;
  }
  __syncthreads();
  for (int c = lane; c < g.n; c += GQ_WAVE) {
    float sx = 0, sy = 0, sz = 0;
    for (int e = 0; e < g.k; ++e) {
      sx += sh[(c * g.k + e) * 3];
      sy += sh[(c * g.k + e) * 3 + 1];
      sz += sh[(c * g.k + e) * 3 + 2];
    }
    float* o = g.g_cpts + ((size_t)row * g.n + c) * 3;
    if (g.accumulate) {
      o[0] += sx;
      o[1] += sy;
      o[2] += sz;
    } else {
      o[0] = sx;
      o[1] = sy;
      o[2] = sz;
    }
  }
}


extern "C" {

// workspace = F (B,6,nz) + x,lam,slack + Ftr + dl_dx + dx + dlam + val + svd + QP workspace
int gq_fc_workspace_bytes(int64_t batch, int n_contact, int n_cone, int max_iter, size_t* bytes) {
  GQ_REQUIRE(bytes && batch >= 0 && n_contact > 0 && n_cone > 0, "fc_workspace_bytes: bad arguments");
  const size_t nz = (size_t)n_contact * n_cone, B = (size_t)batch;
  size_t qp = 0;
  int rc = gq_boxqp_workspace_bytes(batch, (int)nz, max_iter, &qp);
  if (rc) return rc;
  *bytes = gq_al(B * 6 * nz * 4) + gq_al(B * nz * 4) * 4 + gq_al(B * 2 * nz * 4) * 3 + gq_al(B * 4) * 2 + qp + 512;
  return GQ_OK;
}

// E_fc forward (energy_fnc of reference energy.py:35-42 for energy_type "graspqp").
int gq_fc_forward(const float* contact_pts, const float* contact_normals, const float* cog, int64_t batch,
                  int n_contact, int n_cone, float friction, float torque_weight, float max_limit, float svd_gain,
                  float values_gain, float eps, int max_iter, float* e_fc, float* x_sum, int32_t* n_iter,
                  void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GQ_REQUIRE(contact_pts && contact_normals && cog && e_fc && workspace, "fc_forward: null pointer");
  GQ_REQUIRE(batch > 0 && n_contact > 0 && n_cone > 0, "fc_forward: bad sizes");
  const int nz = n_contact * n_cone;
  size_t need = 0;
  int rc = gq_fc_workspace_bytes(batch, n_contact, n_cone, max_iter, &need);
  if (rc) return rc;
  GQ_REQUIRE(workspace_bytes >= need, "fc_forward: workspace too small (%zu < %zu)", workspace_bytes, need);
  GqFcWs w = gq_fc_carve(workspace, (size_t)batch, (size_t)nz, workspace_bytes);
  const int64_t tot = batch * nz;
  hipLaunchKernelGGL(gq_grasp_matrix_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, contact_pts,
                     contact_normals, cog, (int)batch, n_contact, n_cone, friction, torque_weight, w.F);
  GQ_LAUNCH_CHECK();
  // bounds 1 <= x <= max_limit + 1, b = 0, ridge 1e-4 (span.py:348-349, qp_solver.py:101-112)
  const float *resid = nullptr, *snap = nullptr;
  const int* kstar = nullptr;
  rc = gq_lsq_boxqp_iterate_(w.F, 1.0f, max_limit + 1.0f, batch, 6, nz, 1e-4f, eps, max_iter, n_iter, w.qp, w.qp_bytes,
                             stream, &resid, &snap, &kstar);
  if (rc) return rc;
  GqFcArgs a{};
  a.F = w.F;
  a.x = w.x;
  a.lam = w.lam;
  a.slack = w.slack;
  a.resid = resid;
  a.snap = snap;
  a.kstar = kstar;
  a.max_iter = max_iter;
  a.B = (int)batch;
  a.n = n_contact;
  a.k = n_cone;
  a.svd_gain = svd_gain;
  a.values_gain = values_gain;
  a.eps_add = 1e-2f;
  a.e_fc = e_fc;
  a.val = w.val;
  a.svd = w.svd;
  a.Ftr = w.Ftr;
  a.x_sum = x_sum;
  hipLaunchKernelGGL(gq_fc_energy_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), 0, st, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Gradient of E_fc wrt contact_pts for upstream grad_e (B,); must follow gq_fc_forward on the same workspace.
int gq_fc_backward(const float* contact_pts, const float* contact_normals, const float* cog, const float* grad_e,
                   int64_t batch, int n_contact, int n_cone, float friction, float torque_weight, float svd_gain,
                   float values_gain, int accumulate, float* grad_contact_pts, void* workspace, size_t workspace_bytes,
                   void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GQ_REQUIRE(contact_pts && contact_normals && cog && grad_e && grad_contact_pts && workspace, "fc_backward: null");
  GQ_REQUIRE(batch > 0 && n_contact > 0 && n_cone > 0, "fc_backward: bad sizes");
  const int nz = n_contact * n_cone;
  GqFcWs w = gq_fc_carve(workspace, (size_t)batch, (size_t)nz, workspace_bytes);
  // dl/dx = g_val * F'(F x) with g_val = g_e * values_gain * exp(-svd_gain * svd): the scale is applied inside the
  // QP backward kernel (no separate elementwise launch)
  int rc = gq_lsq_boxqp_backward_scaled_(w.F, w.lam, w.slack, w.Ftr, batch, 6, nz, 1e-4f, w.dx, w.dlam, grad_e, w.svd,
                                         svd_gain, values_gain, stream);
  if (rc) return rc;
  GqFcBwdArgs a{};
  a.F = w.F;
  a.x = w.x;
  a.dx = w.dx;
  a.cpts = contact_pts;
  a.cnrm = contact_normals;
  a.cog = cog;
  a.g_e = grad_e;
  a.val = w.val;
  a.svd = w.svd;
  a.B = (int)batch;
  a.n = n_contact;
  a.k = n_cone;
  a.mu = friction;
  a.tw = torque_weight;
  a.svd_gain = svd_gain;
  a.values_gain = values_gain;
  a.eps_add = 1e-2f;
  a.accumulate = accumulate;
  a.g_cpts = grad_contact_pts;
  hipLaunchKernelGGL(gq_fc_grad_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), (size_t)nz * 3 * sizeof(float), st, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// expose intermediate results of the last gq_fc_forward on `workspace` (parity tests / SQPLsqSolver-level users)
int gq_fc_peek(void* workspace, size_t workspace_bytes, int64_t batch, int n_contact, int n_cone, const float** F,
               const float** x, const float** val, const float** svd) {
  GQ_REQUIRE(workspace, "fc_peek: null");
  GqFcWs w = gq_fc_carve(workspace, (size_t)batch, (size_t)n_contact * n_cone, workspace_bytes);
  if (F) *F = w.F;
  if (x) *x = w.x;
  if (val) *val = w.val;
  if (svd) *svd = w.svd;
  return GQ_OK;
}

}  // extern "C"
