// Box-QP whose Hessian is low rank plus ridge: Q = A'A + ridge*I with A (m x nz), m <= 8 -- the force-closure QP
// (m = 6 wrench rows, nz = contacts x cone edges).  One problem per wavefront, lane k owns column(s) k, k+64.
//
// Every interior-point iteration has to solve (Q + diag(d_u + d_l)) dx = rhs.  With Lam = ridge + d_u + d_l
// (diagonal) the matrix is Lam + A'A, so by the Woodbury identity
//     dx = Lam^-1 (rhs - A' y),    (I_m + A Lam^-1 A') y = A Lam^-1 rhs,
// i.e. an m x m SPD system instead of an nz x nz one.  Lam spans ~1e-4 .. 1e8 late in the iteration, so the small
// system and the final difference are formed in fp64 (cond(I + A Lam^-1 A') ~ 1e5; in fp32 the subtraction
// rhs - A'y would lose everything for the free variables); everything else stays fp32.  Cross-lane sums use the
// DPP row_shr/row_bcast network (common.h), no LDS.  The PDIPM control flow is the one of qp_kernels.h
// (qpth 0.0.18 semantics, oracle/ref_cpu/qp.py::pdipm_forward_box).
#pragma once
#include "qp_kernels.h"
#include "wave.h"

template <int M>
struct GqSmall {
  static constexpr int T = M * (M + 1) / 2;
  // in-place Cholesky of the packed lower triangle (idx(i,j) = i(i+1)/2 + j); the diagonal slots receive 1 / L_ii
  static __device__ __forceinline__ void factor(double (&G)[T]) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        double s = G[i * (i + 1) / 2 + j];
#pragma unroll
        for (int t = 0; t < j; ++t) s -= G[i * (i + 1) / 2 + t] * G[j * (j + 1) / 2 + t];
        if (i == j) G[i * (i + 1) / 2 + j] = gq_rsq_d(s);
        else G[i * (i + 1) / 2 + j] = s * G[j * (j + 1) / 2 + j];
      }
    }
  }
  static __device__ __forceinline__ void solve(const double (&L)[T], double (&v)[M]) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
      for (int t = 0; t < i; ++t) v[i] -= L[i * (i + 1) / 2 + t] * v[t];
      v[i] *= L[i * (i + 1) / 2 + i];
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
#pragma unroll
      for (int t = i + 1; t < M; ++t) v[i] -= L[t * (t + 1) / 2 + i] * v[t];
      v[i] *= L[i * (i + 1) / 2 + i];
    }
  }
};

template <int M, int NC>
struct GqLr {
  static constexpr int T = M * (M + 1) / 2;
  float a[NC][M];   // my columns of A
  double il[NC];    // 1 / Lam of my columns (0 for dead columns)
  double L[T];      // Cholesky factor of I + A Lam^-1 A'
  float ridge;

  __device__ __forceinline__ void factor(const float (&lam)[NC], const bool (&live)[NC]) {
    double part[T];
#pragma unroll
    for (int i = 0; i < T; ++i) part[i] = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      il[c] = live[c] ? gq_rcp_d((double)lam[c]) : 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const double ai = (double)a[c][i] * il[c];
#pragma unroll
        for (int j = 0; j <= i; ++j) part[i * (i + 1) / 2 + j] += ai * (double)a[c][j];
      }
    }
    gq_wave_sums_d<T>(part);
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = part[i * (i + 1) / 2 + j] + (i == j ? 1.0 : 0.0);
    GqSmall<M>::factor(L);
  }
  // dx = (Lam + A'A)^-1 rhs
  // y_out (optional): the Woodbury vector y, which IS A dx:  A dx = A Lam^-1 rhs - (M - I) y = v - v + y  (M y = v)
  __device__ __forceinline__ void solve(const float (&rhs)[NC], float (&dx)[NC], float* y_out = nullptr) const {
    double v[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < NC; ++c) s += (double)a[c][i] * ((double)rhs[c] * il[c]);
      v[i] = s;
    }
    gq_wave_sums_d<M>(v);
    GqSmall<M>::solve(L, v);
    if (y_out) {
#pragma unroll
      for (int i = 0; i < M; ++i) y_out[i] = (float)v[i];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double num = (double)rhs[c];
#pragma unroll
      for (int i = 0; i < M; ++i) num -= (double)a[c][i] * v[i];
      dx[c] = (float)(num * il[c]);
    }
  }
  // Q x = A'(A x) + ridge x;  ax_known: A x carried along by the caller (A x_new = A x + alpha A dx, A dx = y) instead of
  // being re-reduced over the wavefront
  __device__ __forceinline__ void matvec(const float (&x)[NC], float (&out)[NC], const float* ax_known = nullptr) const {
    float ax[M];
    if (ax_known) {
#pragma unroll
      for (int i = 0; i < M; ++i) ax[i] = ax_known[i];
    } else {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) s = fmaf(a[c][i], x[c], s);
        ax[i] = s;
      }
      gq_wave_sums_f<M>(ax);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float s = ridge * x[c];
#pragma unroll
      for (int i = 0; i < M; ++i) s = fmaf(a[c][i], ax[i], s);
      out[c] = s;
    }
  }
};

// qpth get_step ratio -v/dv (blocking only for dv < 0) with the hardware reciprocal; zeros and infinities behave like the
// IEEE division (dv = +-0 -> -+inf, 0/0 -> NaN)
__device__ __forceinline__ float gq_step_ratio_rcp(float v, float dv) {
  const float a = -v * __builtin_amdgcn_rcpf(dv);
  return (dv > 0.0f) ? GQ_INF : a;
}

// reduced-KKT solve for my columns (see gq_kkt_solve in qp_core.h)
template <int M, int NC, class SOLVER = GqLr<M, NC>>
__device__ __forceinline__ void gq_lr_kkt(const SOLVER& S, const float (&du)[NC], const float (&dl)[NC],
                                          const float (&idu)[NC], const float (&idl)[NC], const float (&rx)[NC], const float (&rsu)[NC], const float (&rsl)[NC],
                                          const float (&rzu)[NC], const float (&rzl)[NC], float (&dx)[NC],
                                          float (&dsu)[NC], float (&dsl)[NC], float (&dzu)[NC], float (&dzl)[NC],
                                          float* y_out = nullptr) {
  float rhs[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float tu = du[c] * rzu[c] - rsu[c], tl = dl[c] * rzl[c] - rsl[c];
    rhs[c] = -rx[c] - (tu - tl);
  }
  S.solve(rhs, dx, y_out);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    dzu[c] = du[c] * (dx[c] + rzu[c]) - rsu[c];
    dzl[c] = dl[c] * (-dx[c] + rzl[c]) - rsl[c];
    dsu[c] = (-rsu[c] - dzu[c]) * idu[c];  // idu = 1 / du
    dsl[c] = (-rsl[c] - dzl[c]) * idl[c];
  }
}

// All PDIPM iterations of one problem; S.a (my columns of A) and S.ridge are set by the caller.  SOLVER: anything with
// factor(lam, live) / solve(rhs, dx[, y_out]) / matvec(x, out) for (Q + diag(lam - g.ridge)) -- the low-rank form GqLr (A'A +
// ridge I by the Woodbury identity) or the dense LDS form of qp_dense_lds.h (any SPD Q, g.ridge = 0)
template <int M, int NC, class SOLVER = GqLr<M, NC>>
__device__ __forceinline__ void gq_qp_lr_iterate(const GqQpArgs& g, int row, int lane, SOLVER& S,
                                                 const bool (&live)[NC], const float (&p)[NC], const float (&hu)[NC],
                                                 const float (&hl)[NC], float* hist_resid = nullptr,
                                                 float* hist_mu = nullptr) {
  // hist_*: lane it keeps the residual / mu of iteration it (max_iter <= 64) for the caller's stop-rule epilogue
  float h_r = 0.0f, h_m = 0.0f;
  const int nz = g.nz;
  const float m2 = 2.0f * (float)nz;
  float x[NC], su[NC], sl[NC], zu[NC], zl[NC];
  float lam[NC], ones[NC], zero[NC], nhu[NC], nhl[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    lam[c] = g.ridge + 2.0f;
    ones[c] = 1.0f;
    zero[c] = 0.0f;
    nhu[c] = -hu[c];
    nhl[c] = -hl[c];
  }
  // ---- initial point: solve_kkt(d = 1, rx = p, rs = 0, rz = -h) ----------------------------------------------
  S.factor(lam, live);
  gq_lr_kkt<M, NC, SOLVER>(S, ones, ones, ones, ones, p, zero, zero, nhu, nhl, x, su, sl, zu, zl);
  {
    float ms = GQ_INF, mz = GQ_INF;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (live[c]) {
        ms = gq_nanmin(ms, gq_nanmin(su[c], sl[c]));
        mz = gq_nanmin(mz, gq_nanmin(zu[c], zl[c]));
      }
    }
    ms = gq_dpp_nanmin(ms);
    mz = gq_dpp_nanmin(mz);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (ms < 0.0f) {
        su[c] = su[c] - ms + 1.0f;
        sl[c] = sl[c] - ms + 1.0f;
      }
      if (mz < 0.0f) {
        zu[c] = zu[c] - mz + 1.0f;
        zl[c] = zl[c] - mz + 1.0f;
      }
      if (!live[c]) {
        x[c] = 0.0f;
        su[c] = sl[c] = zu[c] = zl[c] = 1.0f;
      }
    }
  }

  float best = 0.0f;
#ifdef GQ_QP_CARRY_AX
  float ax_c[M];
#pragma unroll
  for (int i = 0; i < M; ++i) ax_c[i] = 0.0f;
#endif
  for (int it = 0; it < g.max_iter; ++it) {
    float Qx[NC], rx[NC], rzu[NC], rzl[NC];
#ifdef GQ_QP_CARRY_AX  // A/B: A x carried in scalar registers from the second iteration on (changes the last bits of rx)
    S.matvec(x, Qx, it > 0 ? ax_c : nullptr);
#else
    S.matvec(x, Qx);
#endif
    float a_sz = 0.0f, a_rz = 0.0f, a_rx = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      rx[c] = (zu[c] - zl[c]) + Qx[c] + p[c];
      rzu[c] = x[c] + su[c] - hu[c];
      rzl[c] = -x[c] + sl[c] - hl[c];
      if (live[c]) {
        a_sz += su[c] * zu[c] + sl[c] * zl[c];
        a_rz += rzu[c] * rzu[c] + rzl[c] * rzl[c];
        a_rx += rx[c] * rx[c];
      }
    }
    float red[3] = {a_sz, a_rz, a_rx};
    gq_wave_sums_f<3>(red);
    const float sz = red[0];
    const float mu = fabsf(sz / m2);
    const float resid = sqrtf(red[1]) + sqrtf(red[2]) + m2 * mu;
    const bool record = (it == 0) || (resid < best);  // false for NaN: a NaN iterate never becomes best
    if (record) {
      best = resid;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (live[c]) {
          float* s = g.snap + (((size_t)row * g.max_iter + it) * 5) * nz + lane + GQ_WAVE * c;
          s[0] = x[c];
          s[nz] = zu[c];
          s[2 * nz] = zl[c];
          s[3 * nz] = su[c];
          s[4 * nz] = sl[c];
        }
      }
    }
    if (lane == 0) {
      g.resid[(size_t)row * g.max_iter + it] = resid;
      g.mu[(size_t)row * g.max_iter + it] = mu;
    }
    if (lane == it) {
      h_r = resid;
      h_m = mu;
    }
    if (it == g.max_iter - 1) break;  // qpth returns `best` after the loop; the last update is never used

    // reciprocals of s and z once per iteration (v_rcp_f32, 1 ulp): d = z/s, 1/d = s/z, and the corrector's 1/s
    float du[NC], dl[NC], idu[NC], idl[NC], isu[NC], isl[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      isu[c] = __builtin_amdgcn_rcpf(su[c]);
      isl[c] = __builtin_amdgcn_rcpf(sl[c]);
      du[c] = zu[c] * isu[c];
      dl[c] = zl[c] * isl[c];
      idu[c] = su[c] * __builtin_amdgcn_rcpf(zu[c]);
      idl[c] = sl[c] * __builtin_amdgcn_rcpf(zl[c]);
      lam[c] = g.ridge + du[c] + dl[c];
    }
    S.factor(lam, live);
    float dxa[NC], dsua[NC], dsla[NC], dzua[NC], dzla[NC];
#ifdef GQ_QP_CARRY_AX
    float ya[M], yc[M];
    if (it == 0) {  // A x of the initial point: reduced once
#pragma unroll
      for (int i = 0; i < M; ++i) {
        float sacc = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) sacc = fmaf(S.a[c][i], x[c], sacc);
        ax_c[i] = sacc;
      }
      gq_wave_sums_f<M>(ax_c);
    }
    gq_lr_kkt<M, NC, SOLVER>(S, du, dl, idu, idl, rx, zu, zl, rzu, rzl, dxa, dsua, dsla, dzua, dzla, ya);
#else
    gq_lr_kkt<M, NC, SOLVER>(S, du, dl, idu, idl, rx, zu, zl, rzu, rzl, dxa, dsua, dsla, dzua, dzla);
#endif
    float st = GQ_INF;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (live[c])
        st = gq_nanmin(st, gq_nanmin(gq_nanmin(gq_step_ratio_rcp(zu[c], dzua[c]), gq_step_ratio_rcp(zl[c], dzla[c])),
                                     gq_nanmin(gq_step_ratio_rcp(su[c], dsua[c]), gq_step_ratio_rcp(sl[c], dsla[c]))));
    float alpha = gq_nanmin(gq_dpp_nanmin(st), 1.0f);
    float a_t3 = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (live[c])
        a_t3 += (su[c] + alpha * dsua[c]) * (zu[c] + alpha * dzua[c]) + (sl[c] + alpha * dsla[c]) * (zl[c] + alpha * dzla[c]);
    float sig = gq_dpp_sum(a_t3) / sz;
    sig = sig * sig * sig;
    float rs2u[NC], rs2l[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      rs2u[c] = (-mu * sig + dsua[c] * dzua[c]) * isu[c];
      rs2l[c] = (-mu * sig + dsla[c] * dzla[c]) * isl[c];
    }
    float dxc[NC], dsuc[NC], dslc[NC], dzuc[NC], dzlc[NC];
#ifdef GQ_QP_CARRY_AX
    gq_lr_kkt<M, NC, SOLVER>(S, du, dl, idu, idl, zero, rs2u, rs2l, zero, zero, dxc, dsuc, dslc, dzuc, dzlc, yc);
#else
    gq_lr_kkt<M, NC, SOLVER>(S, du, dl, idu, idl, zero, rs2u, rs2l, zero, zero, dxc, dsuc, dslc, dzuc, dzlc);
#endif
    st = GQ_INF;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      dxa[c] += dxc[c];
      dsua[c] += dsuc[c];
      dsla[c] += dslc[c];
      dzua[c] += dzuc[c];
      dzla[c] += dzlc[c];
      if (live[c])
        st = gq_nanmin(st, gq_nanmin(gq_nanmin(gq_step_ratio_rcp(zu[c], dzua[c]), gq_step_ratio_rcp(zl[c], dzla[c])),
                                     gq_nanmin(gq_step_ratio_rcp(su[c], dsua[c]), gq_step_ratio_rcp(sl[c], dsla[c]))));
    }
    alpha = gq_nanmin(0.999f * gq_dpp_nanmin(st), 1.0f);
#ifdef GQ_QP_CARRY_AX
#pragma unroll
    for (int i = 0; i < M; ++i)  // wave-uniform: kept in scalar registers
      ax_c[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fmaf(alpha, ya[i] + yc[i], ax_c[i]))));
#endif
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (live[c]) {
        x[c] += alpha * dxa[c];
        su[c] += alpha * dsua[c];
        sl[c] += alpha * dsla[c];
        zu[c] += alpha * dzua[c];
        zl[c] += alpha * dzla[c];
      }
    }
  }
  if (hist_resid) *hist_resid = h_r;
  if (hist_mu) *hist_mu = h_m;
}

template <int M, int NC>
__global__ __launch_bounds__(GQ_WAVE) void gq_qp_lr_iter_kernel(GqQpArgs g) {
  const int row = blockIdx.x;
  const int lane = gq_lane();
  const int nz = g.nz;
  GqLr<M, NC> S;
  S.ridge = g.ridge;
  bool live[NC];
  float p[NC], hu[NC], hl[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = lane + GQ_WAVE * c;
    live[c] = k < nz;
    p[c] = 0.0f;
#pragma unroll
    for (int r = 0; r < M; ++r) S.a[c][r] = (live[c] && r < g.m) ? g.A[((size_t)row * g.m + r) * nz + k] : 0.0f;
    if (g.b != nullptr) {
#pragma unroll
      for (int r = 0; r < M; ++r)
        if (r < g.m) p[c] = fmaf(-S.a[c][r], g.b[(size_t)row * g.m + r], p[c]);
    }
    const float up = live[c] ? (g.upper ? g.upper[(size_t)row * nz + k] : g.upper_s) : 1.0f;
    const float lo = live[c] ? (g.lower ? g.lower[(size_t)row * nz + k] : g.lower_s) : -1.0f;
    hu[c] = up;
    hl[c] = -lo;
  }
  gq_qp_lr_iterate<M, NC>(g, row, lane, S, live, p, hu, hl);
}

// backward: dx = -(Q + diag(d_u+d_l))^-1 grad_x, dlam = d * (G dx)
template <int M, int NC>
__global__ __launch_bounds__(GQ_WAVE) void gq_qp_lr_bwd_kernel(GqQpBwdArgs g) {
  const int row = blockIdx.x;
  const int lane = gq_lane();
  const int nz = g.nz;
  GqLr<M, NC> S;
  S.ridge = g.ridge;
  bool live[NC];
  float du[NC], dl[NC], lam[NC], rhs[NC], dx[NC];
  const float gscale = g.scale_ge ? g.scale_ge[row] * g.values_gain * expf(-g.svd_gain * g.scale_svd[row]) : 1.0f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = lane + GQ_WAVE * c;
    live[c] = k < nz;
#pragma unroll
    for (int r = 0; r < M; ++r) S.a[c][r] = (live[c] && r < g.m) ? g.A[((size_t)row * g.m + r) * nz + k] : 0.0f;
    du[c] = dl[c] = 1.0f;
    rhs[c] = 0.0f;
    if (live[c]) {
      const float* lm = g.lam + (size_t)row * 2 * nz;
      const float* sk = g.slack + (size_t)row * 2 * nz;
      du[c] = fmaxf(lm[k], 1e-8f) / fmaxf(sk[k], 1e-8f);
      dl[c] = fmaxf(lm[nz + k], 1e-8f) / fmaxf(sk[nz + k], 1e-8f);
      rhs[c] = -gscale * g.grad_x[(size_t)row * nz + k];
    }
    lam[c] = g.ridge + du[c] + dl[c];
  }
  S.factor(lam, live);
  S.solve(rhs, dx);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (live[c]) {
      const int k = lane + GQ_WAVE * c;
      g.dx[(size_t)row * nz + k] = dx[c];
      g.dlam[(size_t)row * 2 * nz + k] = du[c] * dx[c];
      g.dlam[(size_t)row * 2 * nz + nz + k] = -dl[c] * dx[c];
    }
  }
}

