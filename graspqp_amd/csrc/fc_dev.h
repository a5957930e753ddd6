// Device pieces of the force-closure energy shared by fc.hip (one kernel per stage, C-ABI building blocks) and
// fcstep.hip (the fused per-row kernels of the MALA* stepper).
#pragma once
#include "common.h"
#include "wave.h"

struct GqCone {
  gq3 f;    // cone edge (already divided by k)
  gq3 tau;  // torque_weight * (r x f)
  gq3 r;
};

// column i = contact c = i / k, edge e = i % k
__device__ __forceinline__ GqCone gq_cone_column(const float* cp, const float* cn, const float* cog, int c, int e, int k,
                                                 float mu, float tw) {
  const gq3 n = gq_mk(cn[c * 3], cn[c * 3 + 1], cn[c * 3 + 2]);
  const gq3 p = gq_mk(cp[c * 3], cp[c * 3 + 1], cp[c * 3 + 2]);
  const float is3 = 0.57735026918962576f;
  gq3 b1 = gq_mk(is3, is3, is3);
  const float dot = (b1.x * n.x + b1.y * n.y + b1.z * n.z) * 1.0f / (sqrtf(gq_dot(n, n)) + 1e-6f);
  if (dot > 0.9f) b1.y -= 2.0f;
  const gq3 t1 = gq_cross(n, b1);
  const gq3 t2 = gq_cross(n, t1);
  const float cc = sqrtf(1.0f - mu * mu);
  gq3 dir;
  if (k == 4) {
    const float s = (e < 2) ? mu : -mu;
    dir = (e & 1) ? (s * t2) : (s * t1);
  } else {
    const float ang = 6.283185307179586f / (float)k * (float)e;
    dir = mu * (cosf(ang) * t1 + sinf(ang) * t2);
  }
  GqCone o;
  o.f = (1.0f / (float)k) * (dir + cc * n);
  o.r = p - gq_mk(cog[0], cog[1], cog[2]);
  o.tau = tw * gq_cross(o.r, o.f);
  return o;
}

// 6x6 SPD Cholesky in double (every lane redundantly); inv[i] = 1 / L_ii (Newton-refined rsqrt, no fp64 division or
// square root on the dependent chain); returns false if not positive definite
__device__ __forceinline__ bool gq_chol6(double (&G)[21], double (&Lm)[21], double (&inv)[6]) {
  // packed lower triangle: idx(i,j) = i(i+1)/2 + j
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = G[i * (i + 1) / 2 + j];
#pragma unroll
      for (int t = 0; t < j; ++t) s -= Lm[i * (i + 1) / 2 + t] * Lm[j * (j + 1) / 2 + t];
      if (i == j) {
        if (!(s > 0.0)) {
          ok = false;
          s = 1.0;
        }
        inv[i] = gq_rsq_d(s);
        Lm[i * (i + 1) / 2 + j] = s * inv[i];
      } else {
        Lm[i * (i + 1) / 2 + j] = s * inv[j];
      }
    }
  }
  return ok;
}

// per-contact pieces of E_dis (energy.py:25-28) and of the contact normal fed to E_fc (object_model.py:246)
struct GqContactTerm {
  gq3 vC;   // outward object normal = sign * (p - closest)/|.|
  gq3 g_p;  // w_dis * d E_dis / d p
  gq3 g_n;  // w_dis * d E_dis / d nH
};
__device__ __forceinline__ GqContactTerm gq_contact_term(float d2, float sg, gq3 on, gq3 nH, gq3 p, gq3 closest,
                                                         float w_dis) {
  GqContactTerm o;
  const float root = sqrtf(d2 + 1e-8f);
  o.vC = sg * on;
  const float e = expf(1.0f + gq_dot(o.vC, nH));  // (1 - sum((-vC) nH)).exp()
  // E_dis term = e * |distance| = e * root ; d root / d p = (p - closest)/root
  o.g_p = (w_dis * e / root) * (p - closest);
  o.g_n = (w_dis * e * root) * o.vC;
  return o;
}

// ---- host side: layout of the E_fc workspace (gq_fc_workspace_bytes) ----------------------------------------------
static inline size_t gq_al(size_t v) { return (v + 255) & ~(size_t)255; }
struct GqFcWs {
  float *F, *x, *lam, *slack, *Ftr, *dldx, *dx, *dlam, *val, *svd;
  void* qp;
  size_t qp_bytes;
};
static inline GqFcWs gq_fc_carve(void* base, size_t B, size_t nz, size_t total) {
  GqFcWs w;
  char* c = (char*)base;
  size_t o = 0;
  w.F = (float*)(c + o); o += gq_al(B * 6 * nz * 4);
  w.x = (float*)(c + o); o += gq_al(B * nz * 4);
  w.Ftr = (float*)(c + o); o += gq_al(B * nz * 4);
  w.dldx = (float*)(c + o); o += gq_al(B * nz * 4);
  w.dx = (float*)(c + o); o += gq_al(B * nz * 4);
  w.lam = (float*)(c + o); o += gq_al(B * 2 * nz * 4);
  w.slack = (float*)(c + o); o += gq_al(B * 2 * nz * 4);
  w.dlam = (float*)(c + o); o += gq_al(B * 2 * nz * 4);
  w.val = (float*)(c + o); o += gq_al(B * 4);
  w.svd = (float*)(c + o); o += gq_al(B * 4);
  w.qp = (void*)(c + o);
  w.qp_bytes = total > o ? total - o : 0;
  return w;
}

