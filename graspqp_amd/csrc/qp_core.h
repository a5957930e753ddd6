// One box-constrained QP per wavefront, entirely in registers.
//
//   min 1/2 x'Qx + p'x   s.t.  lower <= x <= upper          (G = [I; -I], h = [upper; -lower])
//
// Replaces qpth's batched PDIPM as driven by the reference (metrics/solver/qp_solver.py:8,101-126;
// algorithm restated in oracle/ref_cpu/qp.py::pdipm_forward_box).  Lane i owns variable i: row i of
// the (symmetric) KKT matrix lives in NZ VGPRs of lane i, vectors are one VGPR per lane.  The reduced
// KKT system (Q + diag(d_u + d_l)) dx = rhs is factored by a right-looking Cholesky whose pivot row
// is broadcast with v_readlane (no LDS): after the factorisation register k of lane i holds
// L[max(i,k)][min(i,k)], so both triangular solves are broadcast + masked FMA sweeps as well.
#pragma once
#include "common.h"

#define GQ_INF GQ_INF_F

template <int NZ>
struct GqChol {
  // in: a[k] = M[lane][k] (full symmetric row).  out: symmetric-L storage + dinv = 1/L[lane][lane].
  static __device__ __forceinline__ void factor(float (&a)[NZ], float& dinv, int lane) {
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
      const float piv = gq_readlane(a[j], j);
      const float r = 1.0f / sqrtf(piv);  // wave-uniform
      const float lij = a[j] * r;
      // lanes > j: eliminate with L_ij; lane j: scale its own row by r (a - (1-r) a = r a); lanes < j: untouched
      const float mult = (lane > j) ? lij * r : ((lane == j) ? (1.0f - r) : 0.0f);
      a[j] = (lane > j) ? lij : a[j];
      dinv = (lane == j) ? r : dinv;
#pragma unroll
      for (int k = j + 1; k < NZ; ++k) {
        const float ajk = gq_readlane(a[k], j);  // pivot row entry M'[j][k], uniform
        a[k] = fmaf(-mult, ajk, a[k]);
      }
    }
  }
  // solve (L L') x = b, b/x one value per lane
  static __device__ __forceinline__ float solve(const float (&a)[NZ], float dinv, int lane, float b) {
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
      const float yj = gq_readlane(b * dinv, j);
      const float m = (lane > j) ? a[j] : 0.0f;
      b = fmaf(-m, yj, b);
    }
    float y = b * dinv;
#pragma unroll
    for (int i = NZ - 1; i >= 0; --i) {
      const float xi = gq_readlane(y * dinv, i);
      const float m = (lane < i) ? a[i] : 0.0f;
      y = fmaf(-m, xi, y);
    }
    return y * dinv;
  }
  // M = Q + diag(dd) into a[]
  static __device__ __forceinline__ void form(float (&a)[NZ], const float (&q)[NZ], float dd, int lane) {
#pragma unroll
    for (int k = 0; k < NZ; ++k) a[k] = q[k] + ((lane == k) ? dd : 0.0f);
  }
  // y_lane = sum_k q[k] * x_k
  static __device__ __forceinline__ float matvec(const float (&q)[NZ], float x) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < NZ; ++k) acc = fmaf(q[k], gq_readlane(x, k), acc);
    return acc;
  }
};

// qpth solve_kkt on the reduced system (see oracle _solve_kkt_box)
template <int NZ>
__device__ __forceinline__ void gq_kkt_solve(const float (&a)[NZ], float dinv, int lane, float du, float dl, float rx,
                                             float rsu, float rsl, float rzu, float rzl, float& dx, float& dsu,
                                             float& dsl, float& dzu, float& dzl) {
  const float tu = du * rzu - rsu, tl = dl * rzl - rsl;
  const float rhs = -rx - (tu - tl);
  dx = GqChol<NZ>::solve(a, dinv, lane, rhs);
  dzu = du * (dx + rzu) - rsu;
  dzl = dl * (-dx + rzl) - rsl;
  dsu = (-rsu - dzu) / du;
  dsl = (-rsl - dzl) / dl;
}

// qpth get_step for one (v, dv) pair per lane and bound side; caller reduces with NaN-propagating min
__device__ __forceinline__ float gq_step_ratio(float v, float dv) {
  const float a = -v / dv;
  return (dv > 0.0f) ? GQ_INF : a;
}

// load row `lane` of Q = A'A + ridge*I from the lane's column of A (m <= 8 rows); idle lanes get identity rows
template <int NZ>
__device__ __forceinline__ void gq_build_q_from_cols(float (&q)[NZ], const float (&col)[8], int m, int nz, int lane,
                                                     float ridge) {
#pragma unroll
  for (int k = 0; k < NZ; ++k) {
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r < m) acc = fmaf(col[r], gq_readlane(col[r], k), acc);
    const bool live = (lane < nz) && (k < nz);
    q[k] = live ? acc + ((lane == k) ? ridge : 0.0f) : ((lane == k) ? 1.0f : 0.0f);
  }
}
