// Kernel templates of the batched box-QP (see qp.hip for the overview); instantiated per NZ in qp_nz*.hip so the
// heavily unrolled bodies compile in parallel.
#pragma once
#include "qp_core.h"

struct GqQpArgs {
  const float* A;      // mode 0: (B, m, nz)
  const float* b;      // mode 0: (B, m) or null (= 0)
  const float* Q;      // mode 1: (B, nz, nz)
  const float* p;      // mode 1: (B, nz) or null
  const float* lower;  // (B, nz) or null -> lower_s
  const float* upper;  // (B, nz) or null -> upper_s
  float lower_s, upper_s, ridge;
  int B, m, nz, max_iter;
  float* resid;  // (B, max_iter)
  float* mu;     // (B, max_iter)
  float* snap;   // (B, max_iter, 5, nz): x, lam_u, lam_l, slack_u, slack_l
};

template <int NZ, int MODE>
__global__ __launch_bounds__(GQ_WAVE) void gq_qp_iter_kernel(GqQpArgs g) {
  const int row = blockIdx.x;
  const int lane = gq_lane();
  const int nz = g.nz;
  const bool live = lane < nz;
  float q[NZ];
  float p = 0.0f;
  if (MODE == 0) {
    float col[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) col[r] = (live && r < g.m) ? g.A[((size_t)row * g.m + r) * nz + lane] : 0.0f;
    gq_build_q_from_cols<NZ>(q, col, g.m, nz, lane, g.ridge);
    if (g.b != nullptr) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r < g.m) p = fmaf(-col[r], g.b[(size_t)row * g.m + r], p);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const bool ok = live && (k < nz);
      q[k] = ok ? g.Q[((size_t)row * nz + lane) * nz + k] : ((lane == k) ? 1.0f : 0.0f);
    }
    if (g.p != nullptr && live) p = g.p[(size_t)row * nz + lane];
  }
  const float up = live ? (g.upper ? g.upper[(size_t)row * nz + lane] : g.upper_s) : 1.0f;
  const float lo = live ? (g.lower ? g.lower[(size_t)row * nz + lane] : g.lower_s) : -1.0f;
  const float hu = up, hl = -lo;
  const float m2 = 2.0f * (float)nz;

  float a[NZ];
  float dinv = 1.0f;
  // ---- initial point: solve_kkt(d = 1, rx = p, rs = 0, rz = -h) --------------------------------------
  GqChol<NZ>::form(a, q, 2.0f, lane);
  GqChol<NZ>::factor(a, dinv, lane);
  float x, su, sl, zu, zl;
  gq_kkt_solve<NZ>(a, dinv, lane, 1.0f, 1.0f, p, 0.0f, 0.0f, -hu, -hl, x, su, sl, zu, zl);
  {
    float ms = gq_dpp_nanmin(live ? gq_nanmin(su, sl) : GQ_INF);
    if (ms < 0.0f) {
      su = su - ms + 1.0f;
      sl = sl - ms + 1.0f;
    }
    float mz = gq_dpp_nanmin(live ? gq_nanmin(zu, zl) : GQ_INF);
    if (mz < 0.0f) {
      zu = zu - mz + 1.0f;
      zl = zl - mz + 1.0f;
    }
  }
  if (!live) {
    x = 0.0f;
    su = sl = zu = zl = 1.0f;
  }

  float best = 0.0f;
  for (int it = 0; it < g.max_iter; ++it) {
    const float Qx = GqChol<NZ>::matvec(q, x);
    const float rx = (zu - zl) + Qx + p;
    const float rzu = x + su - hu;
    const float rzl = -x + sl - hl;
    const float sz = gq_dpp_sum(live ? (su * zu + sl * zl) : 0.0f);
    const float mu = fabsf(sz / m2);
    const float nrz = sqrtf(gq_dpp_sum(live ? (rzu * rzu + rzl * rzl) : 0.0f));
    const float nrx = sqrtf(gq_dpp_sum(live ? rx * rx : 0.0f));
    const float resid = nrz + nrx + m2 * mu;
    const bool record = (it == 0) || (resid < best);  // false for NaN: a NaN iterate never becomes best
    if (record) {
      best = resid;
      if (live) {
        float* s = g.snap + (((size_t)row * g.max_iter + it) * 5) * nz + lane;
        s[0] = x;
        s[nz] = zu;
        s[2 * nz] = zl;
        s[3 * nz] = su;
        s[4 * nz] = sl;
      }
    }
    if (lane == 0) {
      g.resid[(size_t)row * g.max_iter + it] = resid;
      g.mu[(size_t)row * g.max_iter + it] = mu;
    }
    if (it == g.max_iter - 1) break;  // qpth returns `best` after the loop; the last update is never used

    const float du = zu / su, dl = zl / sl;
    GqChol<NZ>::form(a, q, live ? (du + dl) : 2.0f, lane);
    GqChol<NZ>::factor(a, dinv, lane);
    // affine scaling direction
    float dxa, dsua, dsla, dzua, dzla;
    gq_kkt_solve<NZ>(a, dinv, lane, du, dl, rx, zu, zl, rzu, rzl, dxa, dsua, dsla, dzua, dzla);
    float st = gq_nanmin(gq_nanmin(gq_step_ratio(zu, dzua), gq_step_ratio(zl, dzla)),
                         gq_nanmin(gq_step_ratio(su, dsua), gq_step_ratio(sl, dsla)));
    float alpha = gq_nanmin(gq_dpp_nanmin(live ? st : GQ_INF), 1.0f);
    const float t3 = gq_dpp_sum(
        live ? ((su + alpha * dsua) * (zu + alpha * dzua) + (sl + alpha * dsla) * (zl + alpha * dzla)) : 0.0f);
    float sig = t3 / sz;
    sig = sig * sig * sig;
    // centering-corrector direction: rx = 0, rs = (-mu*sig + ds_aff*dz_aff)/s, rz = 0
    const float rs2u = (-mu * sig + dsua * dzua) / su;
    const float rs2l = (-mu * sig + dsla * dzla) / sl;
    float dxc, dsuc, dslc, dzuc, dzlc;
    gq_kkt_solve<NZ>(a, dinv, lane, du, dl, 0.0f, rs2u, rs2l, 0.0f, 0.0f, dxc, dsuc, dslc, dzuc, dzlc);
    const float dx = dxa + dxc, dsu = dsua + dsuc, dsl = dsla + dslc, dzu = dzua + dzuc, dzl = dzla + dzlc;
    st = gq_nanmin(gq_nanmin(gq_step_ratio(zu, dzu), gq_step_ratio(zl, dzl)),
                   gq_nanmin(gq_step_ratio(su, dsu), gq_step_ratio(sl, dsl)));
    alpha = gq_nanmin(0.999f * gq_dpp_nanmin(live ? st : GQ_INF), 1.0f);
    if (live) {
      x += alpha * dx;
      su += alpha * dsu;
      sl += alpha * dsl;
      zu += alpha * dzu;
      zl += alpha * dzl;
    }
  }
}

// ---- backward: (dx, _, dlam) = solve_kkt(d, grad_x, 0, 0), d = clamp(lam,1e-8)/clamp(slack,1e-8) ---------------
struct GqQpBwdArgs {
  const float* A;  // mode 0
  const float* Q;  // mode 1
  const float* lam;
  const float* slack;
  const float* grad_x;
  float ridge;
  int B, m, nz;
  float* dx;    // (B, nz)  = grad wrt p
  float* dlam;  // (B, 2nz) ; grad wrt h = -dlam
  // optional per-row scale of grad_x (fc energy: grad_x = Ftr, scale = g_e * values_gain * exp(-svd_gain * svd))
  const float* scale_ge;
  const float* scale_svd;
  float svd_gain, values_gain;
};

template <int NZ, int MODE>
__global__ __launch_bounds__(GQ_WAVE) void gq_qp_bwd_kernel(GqQpBwdArgs g) {
  const int row = blockIdx.x;
  const int lane = gq_lane();
  const int nz = g.nz;
  const bool live = lane < nz;
  float q[NZ];
  if (MODE == 0) {
    float col[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) col[r] = (live && r < g.m) ? g.A[((size_t)row * g.m + r) * nz + lane] : 0.0f;
    gq_build_q_from_cols<NZ>(q, col, g.m, nz, lane, g.ridge);
  } else {
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const bool ok = live && (k < nz);
      q[k] = ok ? g.Q[((size_t)row * nz + lane) * nz + k] : ((lane == k) ? 1.0f : 0.0f);
    }
  }
  float du = 1.0f, dl = 1.0f, gx = 0.0f;
  if (live) {
    const float* lm = g.lam + (size_t)row * 2 * nz;
    const float* sk = g.slack + (size_t)row * 2 * nz;
    du = fmaxf(lm[lane], 1e-8f) / fmaxf(sk[lane], 1e-8f);
    dl = fmaxf(lm[nz + lane], 1e-8f) / fmaxf(sk[nz + lane], 1e-8f);
    gx = g.grad_x[(size_t)row * nz + lane];
  }
  float a[NZ];
  float dinv = 1.0f;
  GqChol<NZ>::form(a, q, live ? (du + dl) : 2.0f, lane);
  GqChol<NZ>::factor(a, dinv, lane);
  float dx, dsu, dsl, dzu, dzl;
  gq_kkt_solve<NZ>(a, dinv, lane, du, dl, gx, 0.0f, 0.0f, 0.0f, 0.0f, dx, dsu, dsl, dzu, dzl);
  if (live) {
    g.dx[(size_t)row * nz + lane] = dx;
    g.dlam[(size_t)row * 2 * nz + lane] = dzu;
    g.dlam[(size_t)row * 2 * nz + nz + lane] = dzl;
  }
}


// per-NZ launchers (defined in qp_nz16/32/48/64.hip)
#define GQ_DECL_QP_NZ(NZ)                                                   \
  int gq_qp_launch_iter_##NZ(const GqQpArgs& a, int mode, hipStream_t st);  \
  int gq_qp_launch_bwd_##NZ(const GqQpBwdArgs& a, int mode, hipStream_t st);
GQ_DECL_QP_NZ(16)
GQ_DECL_QP_NZ(32)
GQ_DECL_QP_NZ(48)
GQ_DECL_QP_NZ(64)

// only the dense-Q form (mode 1) uses these register-Cholesky kernels; the A'A form goes through qp_lr.hip
#define GQ_DEFINE_QP_NZ(NZ)                                                                                   \
  int gq_qp_launch_iter_##NZ(const GqQpArgs& a, int mode, hipStream_t st) {                                    \
    hipLaunchKernelGGL((gq_qp_iter_kernel<NZ, 1>), dim3(a.B), dim3(GQ_WAVE), 0, st, a);                        \
    GQ_LAUNCH_CHECK();                                                                                         \
    return GQ_OK;                                                                                              \
  }                                                                                                            \
  int gq_qp_launch_bwd_##NZ(const GqQpBwdArgs& a, int mode, hipStream_t st) {                                  \
    hipLaunchKernelGGL((gq_qp_bwd_kernel<NZ, 1>), dim3(a.B), dim3(GQ_WAVE), 0, st, a);                         \
    GQ_LAUNCH_CHECK();                                                                                         \
    return GQ_OK;                                                                                              \
  }
int gq_qp_lr_launch_iter(const GqQpArgs& a, hipStream_t st);
int gq_qp_lr_launch_bwd(const GqQpBwdArgs& a, hipStream_t st);
