// Device bodies of the reference's two other force-closure energies (metric.hip has the overview and the C ABI);
// shared with stage.hip, which runs them as a role of the fused launches of the MALA* stepper.
#pragma once
#include "common.h"
#include "wave.h"

// ---- dexgrasp ---------------------------------------------------------------------------------------------------
struct GqDexArgs {
  const float* cpts; const float* cnrm; const float* cog;
  int B, n;
  float tw;
  const float* grad_e;  // (B) upstream or null (-> w)
  float w;
  int accumulate;
  float* e;        // (B) or null
  float* g_cpts;   // (B,n,3) or null
};

// one wavefront per row; lane c handles contacts c, c + 64, ...  cp_row / cn_row: the row's contact points and object
// normals (n x 3 each) -- global memory in the stand-alone kernel, LDS when the contact terms were computed by the same
// wavefront just before (gq_stage_alt_kernel)
__device__ __forceinline__ void gq_dexgrasp_body(const GqDexArgs& g, int row, int lane, const float* cp_row, const float* cn_row) {
  const float* cg = g.cog + (size_t)row * 3;
  float s[6] = {0, 0, 0, 0, 0, 0};
  for (int c = lane; c < g.n; c += GQ_WAVE) {
    const float* p = cp_row + (size_t)c * 3;
    const float* nn = cn_row + (size_t)c * 3;
    const gq3 r = gq_mk(p[0] - cg[0], p[1] - cg[1], p[2] - cg[2]), n = gq_mk(nn[0], nn[1], nn[2]);
    const gq3 t = gq_cross(n, r);
    s[0] += n.x; s[1] += n.y; s[2] += n.z;
    s[3] += g.tw * t.x; s[4] += g.tw * t.y; s[5] += g.tw * t.z;
  }
  gq_wave_sums_f<6>(s);
  const float E = s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3] + s[4] * s[4] + s[5] * s[5];
  if (lane == 0 && g.e) g.e[row] = E;
  if (g.g_cpts) {
    const float up = g.grad_e ? g.grad_e[row] : g.w;
    // d |tau|^2 / d p_i = 2 tw (tau x n_i) with tau = tw sum_j n_j x r_j (already scaled once)
    const gq3 tau = gq_mk(s[3], s[4], s[5]);
    for (int c = lane; c < g.n; c += GQ_WAVE) {
      const float* nn = cn_row + (size_t)c * 3;
      const gq3 gr = (2.0f * g.tw * up) * gq_cross(tau, gq_mk(nn[0], nn[1], nn[2]));
      float* o = g.g_cpts + ((size_t)row * g.n + c) * 3;
      if (g.accumulate) { o[0] += gr.x; o[1] += gr.y; o[2] += gr.z; }
      else { o[0] = gr.x; o[1] = gr.y; o[2] = gr.z; }
    }
  }
}

// ---- TDG ------------------------------------------------------------------------------------------------------------
struct GqTdgArgs {
  const float* cpts; const float* cnrm; const float* cog;
  const float* dirs;  // (P,3) unit directions (the force part of the reference's target_direction_6D; torque part 0)
  int B, n, P;
  float miu, inv_obb, scale;
  int density;
  const float* grad_e; float w; int accumulate;
  float* e; float* g_cpts;
};

#define GQ_TDG_CH 16  // contacts whose gradients a thread accumulates in registers per pass

// friction-cone force (world frame) of one contact that reaches farthest along direction u: tdg.py:165-196
__device__ __forceinline__ gq3 gq_tdg_force(gq3 u, const float* fr, float miu, float bottom_angle, float cos_half) {
  const gq3 a0 = gq_mk(fr[0], fr[1], fr[2]), a1 = gq_mk(fr[3], fr[4], fr[5]), a2 = gq_mk(fr[6], fr[7], fr[8]);
  const float d0 = gq_dot(u, a0), d1 = gq_dot(u, a1), d2 = gq_dot(u, a2);
  const float inv = 1.0f / fmaxf(sqrtf(d0 * d0 + d1 * d1 + d2 * d2), 1e-12f);
  const float proj = d0 * inv, py = d1 * inv, pz = d2 * inv;
  const float ang = acosf(fminf(fmaxf(proj, -1.0f), 1.0f));
  const float r1 = ang <= bottom_angle ? 1.0f : 0.0f;
  const float r2 = (ang > bottom_angle && ang <= 1.5707963267948966f) ? 1.0f : 0.0f;
  const float r3 = ang > 1.5707963267948966f ? 1.0f : 0.0f;
  const float pn = sqrtf(py * py + pz * pz);
  const float help3 = pn / (pn - 2.0f * miu * fminf(proj, 0.0f));
  const float ipn = 1.0f / fmaxf(pn, 1e-12f);
  const float h2y = miu * py * ipn, h2z = miu * pz * ipn;
  const float den = fmaxf(proj, cos_half);
  // regions multiply (they do not select): a NaN in an inactive branch propagates exactly as in the reference
  const float ax = r1 * 1.0f + r2 * 1.0f + r3 * help3 * 1.0f;
  const float ay = r1 * (py / den) + r2 * h2y + r3 * help3 * h2y;
  const float az = r1 * (pz / den) + r2 * h2z + r3 * help3 * h2z;
  return gq_mk(ax * a0.x + ay * a1.x + az * a2.x, ax * a0.y + ay * a1.y + az * a2.y, ax * a0.z + ay * a1.z + az * a2.z);
}

// block of 256 threads per row; thread t takes directions t, t + 256, ...  gq_tdg_lds: gq_tdg_lds_floats(n) floats;
// cp_row / cn_row as in gq_dexgrasp_body
static inline size_t gq_tdg_lds_floats(int n) { return (size_t)n * 13 + 4 * (3 * GQ_TDG_CH + 1); }
__device__ __forceinline__ void gq_tdg_body(const GqTdgArgs& g, int row, float* gq_tdg_lds, const float* cp_row, const float* cn_row) {
  float* s_fr = gq_tdg_lds;          // n x 9: axis_0 (normal), axis_1, axis_2
  float* s_r = s_fr + g.n * 9;       // n x 3: (p - cog) / obb
  float* s_rho = s_r + g.n * 3;      // n: density
  float* s_red = s_rho + g.n;        // 4 x (3 GQ_TDG_CH + 1)
  const int tid = threadIdx.x, lane = gq_lane(), wv = tid / GQ_WAVE;
  const float* cg = g.cog + (size_t)row * 3;
  for (int c = tid; c < g.n; c += 256) {
    const float* p = cp_row + (size_t)c * 3;
    const float* nn = cn_row + (size_t)c * 3;
    const gq3 a0 = gq_mk(nn[0], nn[1], nn[2]);
    // utils_1axis_to_3axes (tdg.py:75-100): base (0,1,0) unless the normal is within ~8 deg of it, then (0,0,1)
    gq3 a1 = fabsf(a0.y) > 0.99f ? gq_mk(0.0f, 0.0f, 1.0f) : gq_mk(0.0f, 1.0f, 0.0f);
    for (int rep = 0; rep < 2; ++rep) {
      a1 = a1 - gq_dot(a1, a0) * a0;
      a1 = (1.0f / fmaxf(sqrtf(gq_dot(a1, a1)), 1e-12f)) * a1;
    }
    const gq3 a2 = gq_cross(a0, a1);
    float* f = s_fr + c * 9;
    f[0] = a0.x; f[1] = a0.y; f[2] = a0.z; f[3] = a1.x; f[4] = a1.y; f[5] = a1.z; f[6] = a2.x; f[7] = a2.y; f[8] = a2.z;
    s_r[c * 3] = (p[0] - cg[0]) * g.inv_obb; s_r[c * 3 + 1] = (p[1] - cg[1]) * g.inv_obb; s_r[c * 3 + 2] = (p[2] - cg[2]) * g.inv_obb;
    float rho = 1.0f;
    if (g.density) {  // estimate_density (tdg.py:124-127)
      float acc = 0.0f;
      for (int j = 0; j < g.n; ++j) {
        const float* nj = cn_row + (size_t)j * 3;
        acc += fmaxf(a0.x * nj[0] + a0.y * nj[1] + a0.z * nj[2], 0.0f);
      }
      rho = 1.0f / fmaxf(acc, 1e-4f);
    }
    s_rho[c] = rho;
  }
  __syncthreads();
  const float bottom_angle = atanf(g.miu), cos_half = 0.5f * cosf(bottom_angle);
  const float up = g.grad_e ? g.grad_e[row] : g.w;
  float e_acc = 0.0f;
  for (int c0 = 0; c0 < g.n; c0 += GQ_TDG_CH) {  // gradient chunks (one pass for n <= 16)
    float acc[3 * GQ_TDG_CH + 1];
#pragma unroll
    for (int k = 0; k < 3 * GQ_TDG_CH + 1; ++k) acc[k] = 0.0f;
    for (int p = tid; p < g.P; p += 256) {
      const gq3 u = gq_mk(g.dirs[p * 3], g.dirs[p * 3 + 1], g.dirs[p * 3 + 2]);
      gq3 Wf = gq_mk(0, 0, 0), Wt = gq_mk(0, 0, 0);
      for (int c = 0; c < g.n; ++c) {
        const gq3 f = s_rho[c] * gq_tdg_force(u, s_fr + c * 9, g.miu, bottom_angle, cos_half);
        Wf = Wf + f;
        Wt = Wt + gq_cross(gq_mk(s_r[c * 3], s_r[c * 3 + 1], s_r[c * 3 + 2]), f);
      }
      const float nrm = sqrtf(gq_dot(Wf, Wf) + gq_dot(Wt, Wt));
      const float wu = gq_dot(Wf, u);
      if (c0 == 0) acc[3 * GQ_TDG_CH] += 1.0f - wu / fmaxf(nrm, 1e-12f);
      // d cos / d W_tau = -(W_f.u) W_tau / |W|^3 ; W_tau = sum rho_i r_i x f_i  ->  d E / d r_i ~ (W_f.u)/|W|^3 f_i x W_tau
      const float coef = wu / (nrm * nrm * nrm);
#pragma unroll
      for (int k = 0; k < GQ_TDG_CH; ++k) {
        const int c = c0 + k;
        if (c < g.n) {
          const gq3 f = s_rho[c] * gq_tdg_force(u, s_fr + c * 9, g.miu, bottom_angle, cos_half);
          const gq3 gr = coef * gq_cross(f, Wt);
          acc[3 * k] += gr.x; acc[3 * k + 1] += gr.y; acc[3 * k + 2] += gr.z;
        }
      }
    }
    gq_wave_sums_f<3 * GQ_TDG_CH + 1>(acc);  // fixed summation tree: bitwise reproducible
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 3 * GQ_TDG_CH + 1; ++k) s_red[wv * (3 * GQ_TDG_CH + 1) + k] = acc[k];
    }
    __syncthreads();
    if (tid < 3 * GQ_TDG_CH + 1) {
      const int st = 3 * GQ_TDG_CH + 1;
      const float tot = ((s_red[tid] + s_red[st + tid]) + s_red[2 * st + tid]) + s_red[3 * st + tid];
      if (tid == 3 * GQ_TDG_CH) {
        if (c0 == 0) e_acc = tot;
      } else {
        const int c = c0 + tid / 3;
        if (c < g.n && g.g_cpts) {
          const float v = up * g.scale * g.inv_obb * tot / (float)g.P;
          float* o = g.g_cpts + ((size_t)row * g.n + c) * 3 + (tid % 3);
          *o = g.accumulate ? *o + v : v;
        }
      }
    }
    __syncthreads();
  }
  if (tid == 3 * GQ_TDG_CH && g.e) g.e[row] = g.scale * e_acc / (float)g.P;
}

