// Box QP with a DENSE Hessian of 65..128 variables: qpth.qp.QPFunction(Q, p, G = [I; -I], h) as the reference's
// metrics/solver/qp_solver.py:101-125 calls it with 8-edge friction cones (nz = 96 at 12 contacts, 128 at 16).  The
// register-resident Cholesky of qp_core.h stops at 64 variables (one matrix row per lane in VGPRs); here the matrix lives
// in LDS.  One problem per wavefront, lane l owns variables l and l + 64; the PDIPM control flow is the shared
// gq_qp_lr_iterate (qp_lr.h; qpth 0.0.18 semantics), only the linear algebra differs:
//
//   one nz x LD square in LDS holds BOTH matrices: the strict UPPER triangle keeps Q (symmetric: Q_ik for i < k at [i][k]),
//   the strict LOWER triangle receives the Cholesky factor L of M = Q + diag(lam) (L_ik for i > k at [i][k]); the two
//   diagonals are vectors (Q_ii, 1 / L_ii).  Half the LDS of two squares: 4 problems per CU at nz = 96, 2 at 128.
//
//   factor   left-looking: column j of L from the dot products of row i with row j over k < j (16-byte LDS reads: the own
//            row per lane, row j broadcast)
//   solve    forward substitution column by column (y_j is read from its owner lane, every lane updates its rows with
//            L_ij from its own row), backward substitution with row j of L (consecutive lanes, consecutive words)
//   matvec   Q x with x broadcast from LDS, coefficient address min(i,k) * LD + max(i,k)
//
// All fp32, like the register kernels for nz <= 64 (the low-rank route behind SQPLsqSolver / energy_type graspqp, which
// knows Q = A'A + ridge I, stays the fast and the more accurate path: DESIGN.md section 4).
#include "qp_lr.h"

template <int NC>
struct GqDenseLds {
  float* U;     // nz x LD floats (LDS): upper = Q, lower = L
  float* qd;    // nz: diagonal of Q
  float* di;    // nz: 1 / L_ii
  float* vec;   // nz: broadcast scratch (matvec operand, backward-substitution results)
  int nz, LD, lane;
  float ridge;  // unused (interface of gq_qp_lr_iterate): the caller passes g.ridge = 0, lam = d_u + d_l

  // s[c] = sum_{k < len} L_{i_c,k} L_{j,k} for the lane's rows i_c = lane + 64 c (clamped to a valid row when the lane has
  // none: the result is discarded).  Both rows share the broadcast reads of row j; the loop is unrolled so that eight
  // 16-byte LDS reads are in flight (a single wavefront per SIMD hides no latency by itself)
  __device__ __forceinline__ void dot_rows(int j, int len, float (&s)[NC]) const {
    const float* rj = U + (size_t)j * LD;
    const float* ri[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = lane + GQ_WAVE * c;
      ri[c] = U + (size_t)(i < nz ? i : j) * LD;
      s[c] = 0.0f;
    }
    int k = 0;
#pragma unroll 2
    for (; k + 8 <= len; k += 8) {
      const float4 b0 = *reinterpret_cast<const float4*>(rj + k), b1 = *reinterpret_cast<const float4*>(rj + k + 4);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 a0 = *reinterpret_cast<const float4*>(ri[c] + k), a1 = *reinterpret_cast<const float4*>(ri[c] + k + 4);
        float t0 = fmaf(a0.x, b0.x, fmaf(a0.y, b0.y, fmaf(a0.z, b0.z, a0.w * b0.w)));
        float t1 = fmaf(a1.x, b1.x, fmaf(a1.y, b1.y, fmaf(a1.z, b1.z, a1.w * b1.w)));
        s[c] += t0 + t1;
      }
    }
    if (k < len) {  // tail of up to 7 elements: one more (in-bounds: rows are padded) pair of 16-byte reads, masked by select
      const float4 b0 = *reinterpret_cast<const float4*>(rj + k), b1 = *reinterpret_cast<const float4*>(rj + k + 4);
      const int r = len - k;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 a0 = *reinterpret_cast<const float4*>(ri[c] + k), a1 = *reinterpret_cast<const float4*>(ri[c] + k + 4);
        float t = 0.0f;
        t = (r > 0) ? fmaf(a0.x, b0.x, t) : t;
        t = (r > 1) ? fmaf(a0.y, b0.y, t) : t;
        t = (r > 2) ? fmaf(a0.z, b0.z, t) : t;
        t = (r > 3) ? fmaf(a0.w, b0.w, t) : t;
        t = (r > 4) ? fmaf(a1.x, b1.x, t) : t;
        t = (r > 5) ? fmaf(a1.y, b1.y, t) : t;
        t = (r > 6) ? fmaf(a1.z, b1.z, t) : t;
        s[c] += t;
      }
    }
  }

  // M = Q + diag(lam) -> L (lower triangle of U) and di
  __device__ __forceinline__ void factor(const float (&lam)[NC], const bool (&live)[NC]) {
    for (int j = 0; j < nz; ++j) {
      float s[NC], dt[NC];
      dot_rows(j, j, dt);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int i = lane + GQ_WAVE * c;
        s[c] = 0.0f;
        if (i >= j && i < nz) {
          const float m = (i == j) ? qd[j] + lam[c] : U[(size_t)j * LD + i];  // Q_ij for i > j sits at [j][i]
          s[c] = m - dt[c];
        }
      }
      // pivot: row j is owned by lane j % 64, slot j / 64 (wave-uniform)
      const float piv = (j < GQ_WAVE) ? gq_readlane(s[0], j) : gq_readlane(s[NC - 1], j - GQ_WAVE);
      const float r = 1.0f / sqrtf(piv);
      if (lane == 0) di[j] = r;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int i = lane + GQ_WAVE * c;
        if (i > j && i < nz) U[(size_t)i * LD + j] = s[c] * r;
      }
      gq_wave_sync();
    }
  }

  // dx = M^-1 rhs
  __device__ __forceinline__ void solve(const float (&rhs)[NC], float (&dx)[NC], float* y_out = nullptr) const {
    (void)y_out;
    float b[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) b[c] = rhs[c];
    // forward: L y = b, column by column; the lane's L_i,j..j+3 come with one 16-byte read of its own rows
    for (int j0 = 0; j0 < nz; j0 += 4) {
      float4 l4[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int i = lane + GQ_WAVE * c;
        l4[c] = *reinterpret_cast<const float4*>(U + (size_t)(i < nz ? i : 0) * LD + j0);
      }
      const float4 d4 = *reinterpret_cast<const float4*>(di + j0);  // nz is padded to a multiple of 4 in the vectors
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = j0 + t;
        if (j >= nz) break;
        const float bj = (j < GQ_WAVE) ? gq_readlane(b[0], j) : gq_readlane(b[NC - 1], j - GQ_WAVE);
        const float yj = bj * (t == 0 ? d4.x : t == 1 ? d4.y : t == 2 ? d4.z : d4.w);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int i = lane + GQ_WAVE * c;
          const float lij = t == 0 ? l4[c].x : t == 1 ? l4[c].y : t == 2 ? l4[c].z : l4[c].w;
          if (i == j) b[c] = yj;
          else if (i > j && i < nz) b[c] = fmaf(-lij, yj, b[c]);
        }
      }
    }
    // backward: L' x = y, x_j from the last row up; lane k < j subtracts L_jk x_j (row j of L: consecutive words); the reads
    // of four rows are issued together
    for (int j0 = nz - 1; j0 >= 0; j0 -= 4) {
      float l[4][NC], dj[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = j0 - t;
        dj[t] = di[j >= 0 ? j : 0];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int i = lane + GQ_WAVE * c;
          l[t][c] = (j >= 0 && i < j) ? U[(size_t)j * LD + i] : 0.0f;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = j0 - t;
        if (j < 0) break;
        const float yj = (j < GQ_WAVE) ? gq_readlane(b[0], j) : gq_readlane(b[NC - 1], j - GQ_WAVE);
        const float xj = yj * dj[t];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int i = lane + GQ_WAVE * c;
          if (i == j) b[c] = xj;
          else if (i < j) b[c] = fmaf(-l[t][c], xj, b[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) dx[c] = b[c];
  }

  // out = Q x
  __device__ __forceinline__ void matvec(const float (&x)[NC], float (&out)[NC], const float* ax_known = nullptr) const {
    (void)ax_known;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = lane + GQ_WAVE * c;
      if (i < nz) vec[i] = x[c];
    }
    gq_wave_sync();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = lane + GQ_WAVE * c;
      float acc = 0.0f;
      if (i < nz) {
        acc = qd[i] * x[c];
#pragma unroll 8
        for (int k = 0; k < nz; ++k) {
          const int lo = k < i ? k : i, hi = k < i ? i : k;
          const float q = (k == i) ? 0.0f : U[(size_t)lo * LD + hi];
          acc = fmaf(q, vec[k], acc);
        }
      }
      out[c] = acc;
    }
    gq_wave_sync();  // vec is reused
  }
};

template <int NC>
__device__ __forceinline__ GqDenseLds<NC> gq_dense_setup(const float* __restrict__ Q, int row, int nz, int lane, float* lds) {
  GqDenseLds<NC> S;
  S.nz = nz;
  S.LD = (nz + 3) / 4 * 4 + 8;  // 16-byte aligned rows with 8 words of slack (masked tail reads), not a multiple of 32 words
  S.lane = lane;
  S.ridge = 0.0f;
  S.U = lds;
  const int nzp = (nz + 3) / 4 * 4;  // 16-byte aligned vectors
  S.qd = lds + (size_t)nz * S.LD;
  S.di = S.qd + nzp;
  S.vec = S.di + nzp;
  const float* q = Q + (size_t)row * nz * nz;
  for (int e = lane; e < nz * nz; e += GQ_WAVE) {  // coalesced read of the row-major Q; keep the upper triangle + diagonal
    const int i = e / nz, k = e - i * nz;
    const float v = q[e];
    if (k > i) S.U[(size_t)i * S.LD + k] = v;
    else if (k == i) S.qd[i] = v;
  }
  gq_wave_sync();
  return S;
}
static inline size_t gq_dense_lds_bytes(int nz) {
  const int LD = (nz + 3) / 4 * 4 + 8;
  return ((size_t)nz * LD + 3 * (size_t)((nz + 3) / 4 * 4)) * sizeof(float);
}

__global__ __launch_bounds__(GQ_WAVE) void gq_qp_dense_iter_kernel(GqQpArgs g) {
  extern __shared__ float gq_dense_lds[];
  const int row = blockIdx.x, lane = gq_lane(), nz = g.nz;
  GqDenseLds<2> S = gq_dense_setup<2>(g.Q, row, nz, lane, gq_dense_lds);
  bool live[2];
  float p[2], hu[2], hl[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int k = lane + GQ_WAVE * c;
    live[c] = k < nz;
    p[c] = (g.p != nullptr && live[c]) ? g.p[(size_t)row * nz + k] : 0.0f;
    const float up = live[c] ? (g.upper ? g.upper[(size_t)row * nz + k] : g.upper_s) : 1.0f;
    const float lo = live[c] ? (g.lower ? g.lower[(size_t)row * nz + k] : g.lower_s) : -1.0f;
    hu[c] = up;
    hl[c] = -lo;
  }
  g.ridge = 0.0f;  // lam = d_u + d_l: the ridge, if any, is part of Q
  gq_qp_lr_iterate<1, 2, GqDenseLds<2>>(g, row, lane, S, live, p, hu, hl);
}

__global__ __launch_bounds__(GQ_WAVE) void gq_qp_dense_bwd_kernel(GqQpBwdArgs g) {
  extern __shared__ float gq_dense_lds[];
  const int row = blockIdx.x, lane = gq_lane(), nz = g.nz;
  GqDenseLds<2> S = gq_dense_setup<2>(g.Q, row, nz, lane, gq_dense_lds);
  bool live[2];
  float du[2], dl[2], lam[2], rhs[2], dx[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int k = lane + GQ_WAVE * c;
    live[c] = k < nz;
    du[c] = dl[c] = 1.0f;
    rhs[c] = 0.0f;
    if (live[c]) {
      const float* lm = g.lam + (size_t)row * 2 * nz;
      const float* sk = g.slack + (size_t)row * 2 * nz;
      du[c] = fmaxf(lm[k], 1e-8f) / fmaxf(sk[k], 1e-8f);
      dl[c] = fmaxf(lm[nz + k], 1e-8f) / fmaxf(sk[nz + k], 1e-8f);
      rhs[c] = -g.grad_x[(size_t)row * nz + k];  // solve_kkt(d, grad_x, 0, 0): rhs = -rx (qp_kernels.h::gq_qp_bwd_kernel)
    }
    lam[c] = du[c] + dl[c];
  }
  S.factor(lam, live);
  S.solve(rhs, dx);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (live[c]) {
      const int k = lane + GQ_WAVE * c;
      g.dx[(size_t)row * nz + k] = dx[c];
      g.dlam[(size_t)row * 2 * nz + k] = du[c] * dx[c];
      g.dlam[(size_t)row * 2 * nz + nz + k] = -dl[c] * dx[c];
    }
  }
}

int gq_qp_launch_iter_dense_lds(const GqQpArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(gq_qp_dense_iter_kernel, dim3(a.B), dim3(GQ_WAVE), gq_dense_lds_bytes(a.nz), st, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}
int gq_qp_launch_bwd_dense_lds(const GqQpBwdArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(gq_qp_dense_bwd_kernel, dim3(a.B), dim3(GQ_WAVE), gq_dense_lds_bytes(a.nz), st, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}
