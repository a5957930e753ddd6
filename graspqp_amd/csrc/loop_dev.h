// Device bodies of MalaStar.try_step / accept_step (core/optimizer.py:199-273, 289-340), shared by the stand-alone
// kernels of loop.hip and the FK kernels of kin.hip, which run them as head / tail of the same wavefront.
#pragma once
#include "common.h"

struct GqProposeArgs {
  const float* hand_pose;
  const float* grad;
  const float* g2;        // (D)
  const int64_t* idx;     // (B,n)
  const float* u_switch;  // (B,n)
  const int64_t* new_idx; // (B,n)
  int B, D, n, clip;
  float step_size, decay, mu, switch_p;
  int stepsize_period;
  float* ema;       // (B,D) in/out
  int64_t* step;    // (B) in/out
  float* pose_out;  // (B,D)
  int64_t* idx_out; // (B,n)
  float* s_out;     // (B) or null
  const float* energy;  // (B) accepted energies or null
  int batch_each;
  float* z_out;     // (B) per-object z-score of `energy` (fit.py:403-406), written when energy != null
  // fused form (gq_fk_forward with a gqProposeDesc): u_switch / new_idx hold `slots` iterations of draws and
  // slot_ctr[0] selects the current one
  int* slot_ctr;
  int slots;
  int g2_inline;  // gq_fk_forward only: the block's query wavefronts reduce the column means themselves (no launch)
};

__device__ __forceinline__ void gq_zscore_row(const GqProposeArgs& g, int row, int lane);
// Everything the proposal reads except the column means g2: requested up front (gq_fk_forward issues it before the
// block barrier behind which g2 becomes available, so the proposal itself starts with all operands in registers).
struct GqProposePre {
  int64_t st;
  float gr[2], em[2], hp[2];
  float us;          // switch draw of contact `lane`
  int64_t nix, oix;  // drawn / current index of contact `lane`
  size_t draw0;
  int ctr;
};
__device__ __forceinline__ GqProposePre gq_propose_prefetch(const GqProposeArgs& g, int row, int lane, int slot_now = -1) {
  GqProposePre p;
  // slot_now >= 0: the caller has already read slot_ctr[0] (early, so that the dependent loads below need not wait for it)
  p.ctr = g.slot_ctr ? (slot_now >= 0 ? slot_now : g.slot_ctr[0]) : 0;
  p.draw0 = g.slot_ctr ? (size_t)(p.ctr % g.slots) * g.B * g.n : 0;
  p.st = g.step[row];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int d = lane + GQ_WAVE * c;
    const size_t o = (size_t)row * g.D + d;
    p.gr[c] = d < g.D ? g.grad[o] : 0.0f;
    p.em[c] = d < g.D ? g.ema[o] : 0.0f;
    p.hp[c] = d < g.D ? g.hand_pose[o] : 0.0f;
  }
  p.us = 2.0f;
  p.nix = p.oix = 0;
  if (lane < g.n) {
    const size_t o = (size_t)row * g.n + lane;
    p.us = g.u_switch[p.draw0 + o];
    p.nix = g.new_idx[p.draw0 + o];
    p.oix = g.idx[o];
  }
  return p;
}
// s_pose (LDS, D floats) / my_idx: optional copies of the proposal for the caller's own use (pose element d in s_pose[d],
// the index of contact `lane` in *my_idx), so that the FK code of the same wavefront need not wait for its own stores;
// with_z = false leaves the z-score to a separate gq_zscore_row call (e.g. by another, idle wavefront).
__device__ __forceinline__ void gq_propose_finish(const GqProposeArgs& g, const GqProposePre& p, int row, int lane,
                                                  const float (&g2)[2], float* s_pose = nullptr, int64_t* my_idx = nullptr,
                                                  bool with_z = true) {
  if (g.slot_ctr && row == 0 && lane == 0) g.slot_ctr[1] = p.ctr + 1;  // read by the accept of this iteration
  const int64_t st = p.st;
  const float s = g.step_size * powf(g.decay, (float)((int)st / g.stepsize_period));  // 32-bit division: st < 2^31
  float v[2] = {0.0f, 0.0f};
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int d = lane + GQ_WAVE * c;
    if (d < g.D) {
      const size_t o = (size_t)row * g.D + d;
      float gr = p.gr[c];
      if (g.clip) {
        // torch.clip keeps NaN (then zeroed, optimizer.py:211-213); fminf / fmaxf would turn it into a bound instead
        gr = (gr != gr) ? 0.0f : fminf(fmaxf(gr, -100.0f), 100.0f);
      }
      float em = g.mu * g2[c] + (1.0f - g.mu) * p.em[c];
      if (em != em) em = 0.0f;
      g.ema[o] = em;
      v[c] = p.hp[c] - s * gr / (sqrtf(em) + 1e-6f);
      bad |= (v[c] != v[c]);
    }
  }
  const bool zero_row = __ballot(bad) != 0ull;  // optimizer.py:242-244: a row with any NaN is zeroed
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int d = lane + GQ_WAVE * c;
    if (d < g.D) {
      const float pv = zero_row ? 0.0f : v[c];
      g.pose_out[(size_t)row * g.D + d] = pv;
      if (s_pose) s_pose[d] = pv;
    }
  }
  if (lane < g.n) {
    const int64_t ix = (p.us < g.switch_p) ? p.nix : p.oix;
    g.idx_out[(size_t)row * g.n + lane] = ix;
    if (my_idx) *my_idx = ix;
  }
  for (int c = lane + GQ_WAVE; c < g.n; c += GQ_WAVE) {
    const size_t o = (size_t)row * g.n + c;
    g.idx_out[o] = (g.u_switch[p.draw0 + o] < g.switch_p) ? g.new_idx[p.draw0 + o] : g.idx[o];
  }
  if (lane == 0) {
    g.step[row] = st + 1;
    if (g.s_out) g.s_out[row] = s;
  }
  if (with_z) gq_zscore_row(g, row, lane);
}
// one wavefront per row: lane d owns pose elements d and d + 64 (D <= 128), lane c owns contact c
__device__ __forceinline__ void gq_propose_body(const GqProposeArgs& g, int row, int lane, float* s_pose = nullptr,
                                                int64_t* my_idx = nullptr, bool with_z = true, int slot_now = -1) {
  float g2[2];
  g2[0] = lane < g.D ? g.g2[lane] : 0.0f;
  g2[1] = lane + GQ_WAVE < g.D ? g.g2[lane + GQ_WAVE] : 0.0f;
  const GqProposePre p = gq_propose_prefetch(g, row, lane, slot_now);
  gq_propose_finish(g, p, row, lane, g2, s_pose, my_idx, with_z);
}
// z = (E - mean_obj) / std_obj (unbiased) over the rows of this row's object (fit.py:403-406)
__device__ __forceinline__ void gq_zscore_row(const GqProposeArgs& g, int row, int lane) {
  if (g.energy) {
    const float* e = g.energy + (size_t)(row / g.batch_each) * g.batch_each;
    float acc = 0.0f;
    for (int i = lane; i < g.batch_each; i += GQ_WAVE) acc += e[i];
    const float mean = gq_dpp_sum(acc) / (float)g.batch_each;
    acc = 0.0f;
    for (int i = lane; i < g.batch_each; i += GQ_WAVE) {
      const float d = e[i] - mean;
      acc = fmaf(d, d, acc);
    }
    const float sd = sqrtf(gq_dpp_sum(acc) / (float)(g.batch_each - 1));
    if (lane == 0) g.z_out[row] = (g.energy[row] - mean) / sd;
  }
}

// Column means of clip(grad)^2 over all rows (optimizer.py:231) for batches <= 512 rows and D <= 64, in ONE canonical
// order so that the stand-alone launch (loop.hip) and the query wavefronts of gq_fk_forward (kin.hip) give the same bits:
// unit u < 16 sums the rows u, u + 16, ... of every column (lane = column) with fma(v, v, acc); the 16 partial sums are
// added in unit order and divided by the number of rows.
#define GQ_COLSQ_UNITS 16
__device__ __forceinline__ float gq_colsq_unit(const float* __restrict__ grad, int B, int D, int clip, int unit, int col) {
  float acc = 0.0f;
#pragma unroll 4
  for (int r = unit; r < B; r += GQ_COLSQ_UNITS) {
    float v = grad[(size_t)r * D + col];
    if (clip) v = (v != v) ? 0.0f : fminf(fmaxf(v, -100.0f), 100.0f);  // NaN -> 0 first: fmaxf(NaN, -100) would be -100
    acc = fmaf(v, v, acc);
  }
  return acc;
}
// sPart: [GQ_COLSQ_UNITS][D] partial sums
__device__ __forceinline__ float gq_colsq_finish(const float* sPart, int B, int D, int col) {
  float t = 0.0f;
  for (int u = 0; u < GQ_COLSQ_UNITS; ++u) t += sPart[u * D + col];
  return t / (float)B;
}

struct GqAcceptArgs {
  const float* new_energy;
  const float* u_accept;
  const float* z;            // (B) or null
  const uint8_t* reset_mask; // (B) or null
  const int64_t* step;       // post-propose counter
  const float* pose_new;
  const int64_t* idx_new;
  const float* grad_new;
  int B, D, n;
  float T0, decay;
  int annealing_period;
  float* energy;   // (B) in/out
  float* pose;     // (B,D) in/out (accepted state)
  int64_t* idx;    // (B,n) in/out
  float* grad;     // (B,D) in/out
  uint8_t* accept; // (B)
  float* temperature; // (B) or null
  int n_terms;
  const float* terms_new;  // (n_terms,B) or null
  float* terms;            // (n_terms,B) or null
  // fused form (gq_fk_backward with a gqAcceptDesc): u_accept holds `slots` iterations of draws; slot_ctr[1] - 1 is
  // the current one (written by this iteration's propose), and slot_ctr[0] is advanced for the next iteration
  int* slot_ctr;
  int slots;
};

__device__ __forceinline__ void gq_accept_body(const GqAcceptArgs& g, int row, int lane) {
  const size_t draw0 = g.slot_ctr ? (size_t)((g.slot_ctr[1] - 1) % g.slots) * g.B : 0;
  if (g.slot_ctr && row == 0 && lane == 0) g.slot_ctr[0] = g.slot_ctr[1];
  float T = g.T0 * powf(g.decay, (float)((int)g.step[row] / g.annealing_period));
  if (g.z) {
    const float proba = 0.5f * (1.0f + erff(g.z[row] * 0.70710678118654752f));
    T = T * (1.0f + proba);
  }
  const float e_old = g.energy[row], e_new = g.new_energy[row];
  bool acc = g.u_accept[draw0 + row] < expf((e_old - e_new) / T);
  if (g.reset_mask && g.reset_mask[row]) acc = true;
  if (lane == 0) {
    g.accept[row] = acc ? 1 : 0;
    if (g.temperature) g.temperature[row] = T;
    if (acc) g.energy[row] = e_new;
  }
  if (acc) {  // wave-uniform
    for (int d = lane; d < g.D; d += GQ_WAVE) {
      const size_t o = (size_t)row * g.D + d;
      g.pose[o] = g.pose_new[o];
      g.grad[o] = g.grad_new[o];
    }
    for (int c = lane; c < g.n; c += GQ_WAVE) g.idx[(size_t)row * g.n + c] = g.idx_new[(size_t)row * g.n + c];
    for (int t = lane; t < g.n_terms; t += GQ_WAVE) g.terms[(size_t)t * g.B + row] = g.terms_new[(size_t)t * g.B + row];
  }
}

// The same step split for the FK backward kernel: everything that does not depend on the new energy is loaded (and
// the temperature computed) at the top of the kernel; the finish takes the new energy and gradient from registers /
// LDS instead of re-reading what the wavefront has just stored.
struct GqAcceptPre {
  float u, e_old, T;
  bool reset;
  float pose_new[2];
  int64_t idx_new;
};
__device__ __forceinline__ GqAcceptPre gq_accept_prefetch(const GqAcceptArgs& g, int row, int lane) {
  GqAcceptPre p;
  const size_t draw0 = g.slot_ctr ? (size_t)((g.slot_ctr[1] - 1) % g.slots) * g.B : 0;
  p.u = g.u_accept[draw0 + row];
  p.e_old = g.energy[row];
  p.reset = g.reset_mask && g.reset_mask[row];
  p.pose_new[0] = lane < g.D ? g.pose_new[(size_t)row * g.D + lane] : 0.0f;
  p.pose_new[1] = lane + GQ_WAVE < g.D ? g.pose_new[(size_t)row * g.D + lane + GQ_WAVE] : 0.0f;
  p.idx_new = lane < g.n ? g.idx_new[(size_t)row * g.n + lane] : 0;
  float T = g.T0 * powf(g.decay, (float)((int)g.step[row] / g.annealing_period));
  if (g.z) {
    const float proba = 0.5f * (1.0f + erff(g.z[row] * 0.70710678118654752f));
    T = T * (1.0f + proba);
  }
  p.T = T;
  return p;
}
// e_new: the row's new total (wave-uniform); sG: the row's new gradient (D floats, LDS); term: lane t < n_terms holds
// the t-th energy term of the new state.
__device__ __forceinline__ void gq_accept_finish(const GqAcceptArgs& g, const GqAcceptPre& p, int row, int lane,
                                                 float e_new, const float* sG, float term) {
  if (g.slot_ctr && row == 0 && lane == 0) g.slot_ctr[0] = g.slot_ctr[1];
  bool acc = p.u < expf((p.e_old - e_new) / p.T);
  if (p.reset) acc = true;
  if (lane == 0) {
    g.accept[row] = acc ? 1 : 0;
    if (g.temperature) g.temperature[row] = p.T;
    if (acc) g.energy[row] = e_new;
  }
  if (acc) {  // wave-uniform
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int d = lane + GQ_WAVE * c;
      if (d < g.D) {
        const size_t o = (size_t)row * g.D + d;
        g.pose[o] = p.pose_new[c];
        g.grad[o] = sG[d];
      }
    }
    if (lane < g.n) g.idx[(size_t)row * g.n + lane] = p.idx_new;
    for (int c = lane + GQ_WAVE; c < g.n; c += GQ_WAVE) g.idx[(size_t)row * g.n + c] = g.idx_new[(size_t)row * g.n + c];
    if (lane < g.n_terms) g.terms[(size_t)lane * g.B + row] = term;
  }
}

