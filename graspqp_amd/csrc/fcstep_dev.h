// Device bodies of the fused force-closure step (see fcstep.hip), shared with the stage kernels of stage.hip.
#pragma once
#include "qp_lr.h"
#include "fc_dev.h"

struct GqFcStepArgs {
  const float* dist_sq;   // (B,n) object SDF of the contact points
  const int32_t* sign;    // (B,n)
  const float* onrm;      // (B,n,3)
  const float* closest;   // (B,n,3)
  const float* cpts;      // (B,n,3)
  const float* cnrm;      // (B,n,3) hand contact normals (world)
  const float* cog;       // (B,3)
  int B, n, k, nz, max_iter, not_improved_lim;
  float mu, tw, w_dis, w_fc, lower, upper, ridge, svd_gain, values_gain, eps_add, eps;
  float* obj_normal;  // (B,n,3)
  float* g_cpts;      // (B,n,3)  head: w_dis dE_dis/dp ; tail: += w_fc dE_fc/dp
  float* g_cnrm;      // (B,n,3)
  float* F;           // (B,6,nz)
  float* resid;       // (B,max_iter)
  float* mu_tab;      // (B,max_iter)
  float* snap;        // (B,max_iter,5,nz)
  int* kstar;         // [stop iteration, iterations]
  int32_t* n_iter;    // or null
  float* e_fc;        // (B)
  float* val;         // (B)
  float* svd;         // (B)
  float* x;           // (B,nz) best iterate (kept for gq_fc_peek)
  float* x_sum;       // (B,n) or null
  // batches too large for the tail to replay the stop rule per row (gq_qp_stop_rows): the head blocks leave per-block
  // aggregates and the block that finishes last applies qpth's rule to them (gq_fc_head_epilogue) -- no stop launch
  unsigned* agg;      // (head_blocks, GQ_AGG_WORDS) or null
  unsigned* head_ctr; // one word, zero between launches
  int head_blocks;
};
#ifndef GQ_HEAD_ROWS
#define GQ_HEAD_ROWS 4  // fc-head rows (wavefronts) per block
#endif
#define GQ_AGG_WORDS 36  // 16 x max of the running best residual | 16 x min of mu | improved-bit mask | 3 pad
#define GQ_HEAD_LDS_WORDS (4 * GQ_AGG_WORDS)  // epilogue: the aggregates of the block's rows

// one wavefront = one row; sh: n*6 floats of LDS (contact points, object normals)
template <int NC>
__device__ __forceinline__ void gq_fc_head_body(const GqFcStepArgs& g, int row, float* sh, float* hist_resid = nullptr,
                                                float* hist_mu = nullptr) {
  const int lane = gq_lane();
  float* s_cp = sh;
  float* s_on = sh + g.n * 3;
  for (int c = lane; c < g.n; c += GQ_WAVE) {
    const size_t t = (size_t)row * g.n + c;
    const gq3 on = gq_mk(g.onrm[t * 3], g.onrm[t * 3 + 1], g.onrm[t * 3 + 2]);
    const gq3 nH = gq_mk(g.cnrm[t * 3], g.cnrm[t * 3 + 1], g.cnrm[t * 3 + 2]);
    const gq3 p = gq_mk(g.cpts[t * 3], g.cpts[t * 3 + 1], g.cpts[t * 3 + 2]);
    const gq3 cl = gq_mk(g.closest[t * 3], g.closest[t * 3 + 1], g.closest[t * 3 + 2]);
    const GqContactTerm ct = gq_contact_term(g.dist_sq[t], (float)g.sign[t], on, nH, p, cl, g.w_dis);
    g.obj_normal[t * 3] = ct.vC.x;
    g.obj_normal[t * 3 + 1] = ct.vC.y;
    g.obj_normal[t * 3 + 2] = ct.vC.z;
    g.g_cpts[t * 3] = ct.g_p.x;
    g.g_cpts[t * 3 + 1] = ct.g_p.y;
    g.g_cpts[t * 3 + 2] = ct.g_p.z;
    g.g_cnrm[t * 3] = ct.g_n.x;
    g.g_cnrm[t * 3 + 1] = ct.g_n.y;
    g.g_cnrm[t * 3 + 2] = ct.g_n.z;
    s_cp[c * 3] = p.x;
    s_cp[c * 3 + 1] = p.y;
    s_cp[c * 3 + 2] = p.z;
    s_on[c * 3] = ct.vC.x;
    s_on[c * 3 + 1] = ct.vC.y;
    s_on[c * 3 + 2] = ct.vC.z;
  }
  gq_wave_sync();
  const int nz = g.nz;
  GqLr<6, NC> S;
  S.ridge = g.ridge;
  bool live[NC];
  float p[NC], hu[NC], hl[NC];
  const float* cog = g.cog + (size_t)row * 3;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane + GQ_WAVE * c;
    live[c] = i < nz;
    p[c] = 0.0f;  // b = 0 (span.py:333)
    hu[c] = live[c] ? g.upper : 1.0f;
    hl[c] = live[c] ? -g.lower : 1.0f;
#pragma unroll
    for (int r = 0; r < 6; ++r) S.a[c][r] = 0.0f;
    if (live[c]) {
      const GqCone cone = gq_cone_column(s_cp, s_on, cog, i / g.k, i % g.k, g.k, g.mu, g.tw);
      S.a[c][0] = cone.f.x;
      S.a[c][1] = cone.f.y;
      S.a[c][2] = cone.f.z;
      S.a[c][3] = cone.tau.x;
      S.a[c][4] = cone.tau.y;
      S.a[c][5] = cone.tau.z;
      float* o = g.F + (size_t)row * 6 * nz + i;
#pragma unroll
      for (int r = 0; r < 6; ++r) o[r * nz] = S.a[c][r];
    }
  }
  GqQpArgs q{};
  q.nz = nz;
  q.max_iter = g.max_iter;
  q.ridge = g.ridge;
  q.resid = g.resid;
  q.mu = g.mu_tab;
  q.snap = g.snap;
  gq_qp_lr_iterate<6, NC>(q, row, lane, S, live, p, hu, hl, hist_resid, hist_mu);
}

// qpth's batch-global stop rule without a launch of its own, for any batch size (max_iter <= 16).  Called by every
// wavefront of a head block after its row (lane it holds the row's residual / mu of iteration it):
//   1. each wavefront replays its row: running best residual per iteration, "improved in iteration it" bits;
//   2. wavefront 0 folds the block's rows (max of the running bests, min of mu -- NaN-propagating like torch -- and the
//      OR of the bits) into one 33-word record in global memory, publishes it and counts the block;
//   3. the block that finishes LAST folds all records and applies the sequential rule (notImproved == lim | max best <
//      eps | min mu > 1e32) -> kstar / n_iter.  The counter wraps to zero by itself (atomicInc), so the workspace only
//      has to be zero once, when it is allocated.
// sh: GQ_HEAD_LDS_WORDS words of LDS; nrow: rows (wavefronts) of this block that are alive.
__device__ __forceinline__ void gq_fc_head_epilogue(const GqFcStepArgs& g, int blk, int wv, int nrow, float hist_resid,
                                                    float hist_mu, unsigned* sh) {
  const int lane = gq_lane();
  float bst = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hist_resid), 0));
  float my_bst = bst;
  unsigned any = 0u;
#pragma unroll
  for (int it = 1; it < 16; ++it) {
    const float rs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hist_resid), it));
    if (it < g.max_iter && rs < bst) {  // wave-uniform; false for NaN on either side, like the reference's comparison
      bst = rs;
      any |= 1u << it;
    }
    if (lane == it) my_bst = bst;
  }
  if (lane < 16) {
    sh[wv * GQ_AGG_WORDS + lane] = __float_as_uint(my_bst);
    sh[wv * GQ_AGG_WORDS + 16 + lane] = __float_as_uint(hist_mu);
  }
  if (lane == 32) sh[wv * GQ_AGG_WORDS + 32] = any;
  __syncthreads();
  if (wv != 0) return;
  unsigned word = 0u;
  if (lane < 16) {
    float v = -GQ_INF;
    for (int r = 0; r < nrow; ++r) v = gq_nanmax(v, __uint_as_float(sh[r * GQ_AGG_WORDS + lane]));
    word = __float_as_uint(v);
  } else if (lane < 32) {
    float v = GQ_INF;
    for (int r = 0; r < nrow; ++r) v = gq_nanmin(v, __uint_as_float(sh[r * GQ_AGG_WORDS + lane]));
    word = __float_as_uint(v);
  } else if (lane == 32) {
    for (int r = 0; r < nrow; ++r) word |= sh[r * GQ_AGG_WORDS + 32];
  }
  // The record goes out with agent-scope (write-through) stores and the counter is bumped once they are acknowledged.
  // NOT a release fence: that would write back this XCD's whole L2, which is full of the iterate snapshots the rows
  // have just stored (24 MB per launch at 2048 rows).
  if (lane <= 32) __hip_atomic_store(&g.agg[(size_t)blk * GQ_AGG_WORDS + lane], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned old = 0u;
  if (lane == 0) old = atomicInc(g.head_ctr, (unsigned)g.head_blocks - 1u);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  if (old != (unsigned)g.head_blocks - 1u) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // invalidate only: the other blocks' records come from memory
  // fold all records: lane L takes the records L, L + 64, ... whole
  // Register diet (this code is compiled into every wavefront of the launch although only the last block runs it): two
  // passes over the records -- the 16 maxima, then the 16 minima + the bit mask -- each with two records (4 + 4 sixteen-
  // byte loads) in flight per lane: 48 live registers instead of 68 + 36, the same number of round trips.
  float amx[16], amn[16];
  unsigned abits = 0u;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    amx[i] = -GQ_INF;
    amn[i] = GQ_INF;
  }
  const uint4* rec4 = reinterpret_cast<const uint4*>(g.agg);
  for (int b0 = lane; b0 < g.head_blocks; b0 += 2 * GQ_WAVE) {
    uint4 q[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int b = b0 + j * GQ_WAVE < g.head_blocks ? b0 + j * GQ_WAVE : b0;  // a duplicate record changes nothing
#pragma unroll
      for (int k = 0; k < 4; ++k) q[j][k] = rec4[(size_t)b * (GQ_AGG_WORDS / 4) + k];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        amx[4 * k + 0] = gq_nanmax(amx[4 * k + 0], __uint_as_float(q[j][k].x));
        amx[4 * k + 1] = gq_nanmax(amx[4 * k + 1], __uint_as_float(q[j][k].y));
        amx[4 * k + 2] = gq_nanmax(amx[4 * k + 2], __uint_as_float(q[j][k].z));
        amx[4 * k + 3] = gq_nanmax(amx[4 * k + 3], __uint_as_float(q[j][k].w));
      }
  }
  // the maxima leave the vector registers before the second pass starts: lane i keeps maximum i
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float mx = -gq_dpp_nanmin(-amx[i]);  // NaN-propagating max, wave-uniform
    if (lane == i) acc = mx;
  }
  for (int b0 = lane; b0 < g.head_blocks; b0 += 2 * GQ_WAVE) {
    uint4 q[2][4];
    unsigned qb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int b = b0 + j * GQ_WAVE < g.head_blocks ? b0 + j * GQ_WAVE : b0;
#pragma unroll
      for (int k = 0; k < 4; ++k) q[j][k] = rec4[(size_t)b * (GQ_AGG_WORDS / 4) + 4 + k];
      qb[j] = g.agg[(size_t)b * GQ_AGG_WORDS + 32];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        amn[4 * k + 0] = gq_nanmin(amn[4 * k + 0], __uint_as_float(q[j][k].x));
        amn[4 * k + 1] = gq_nanmin(amn[4 * k + 1], __uint_as_float(q[j][k].y));
        amn[4 * k + 2] = gq_nanmin(amn[4 * k + 2], __uint_as_float(q[j][k].z));
        amn[4 * k + 3] = gq_nanmin(amn[4 * k + 3], __uint_as_float(q[j][k].w));
      }
      abits |= qb[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float mn = gq_dpp_nanmin(amn[i]);
    if (lane == 16 + i) acc = mn;
  }
  unsigned bits = abits;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bits |= (unsigned)__shfl_xor((int)bits, o, GQ_WAVE);
  const unsigned anyb = (unsigned)__builtin_amdgcn_readlane((int)bits, 32);
  int not_improved = 0, stop_at = g.max_iter - 1;
  bool done = false;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const float bmx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), it));
    const float bmn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 16 + it));
    if (it < g.max_iter && !done) {
      not_improved = (it == 0) ? 0 : (((anyb >> it) & 1u) ? 0 : not_improved + 1);
      if ((not_improved == g.not_improved_lim) || (bmx < g.eps) || (bmn > 1e32f)) {
        stop_at = it;
        done = true;
      }
    }
  }
  if (lane == 0) {
    g.kstar[0] = stop_at;
    g.kstar[1] = stop_at + 1;
    if (g.n_iter) *g.n_iter = stop_at + 1;
  }
}

// qpth's batch-global stop rule (qp.hip::gq_qp_stop_wave_kernel) evaluated by one wavefront from registers: lane l owns
// rows l, l+64, ... (RPL of them); returns the last iteration whose record counts.  B <= 64*RPL, max_iter <= 16.
template <int RPL>
__device__ __forceinline__ int gq_qp_stop_rows(const float* __restrict__ resid, const float* __restrict__ mu, int B,
                                               int max_iter, float eps, int lim, int lane) {
  float rs[RPL][16], ms[RPL][16];
  if ((max_iter & 3) == 0) {  // rows are 16-byte aligned: four iterations per load
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
      const int r = lane + GQ_WAVE * k;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = r < B && q * 4 < max_iter;
        const float4 a = ok ? *reinterpret_cast<const float4*>(resid + (size_t)r * max_iter + q * 4) : make_float4(0, 0, 0, 0);
        const float4 b = ok ? *reinterpret_cast<const float4*>(mu + (size_t)r * max_iter + q * 4) : make_float4(0, 0, 0, 0);
        rs[k][q * 4] = a.x; rs[k][q * 4 + 1] = a.y; rs[k][q * 4 + 2] = a.z; rs[k][q * 4 + 3] = a.w;
        ms[k][q * 4] = b.x; ms[k][q * 4 + 1] = b.y; ms[k][q * 4 + 2] = b.z; ms[k][q * 4 + 3] = b.w;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
      const int r = lane + GQ_WAVE * k;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const bool ok = r < B && it < max_iter;
        rs[k][it] = ok ? resid[(size_t)r * max_iter + it] : 0.0f;
        ms[k][it] = ok ? mu[(size_t)r * max_iter + it] : 0.0f;
      }
    }
  }
  float run[RPL];
  int not_improved = 0, stop_at = max_iter - 1;
  bool done = false;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    if (it < max_iter && !done) {  // wave-uniform
      float mx = -GQ_INF, mn = GQ_INF;
      bool any = false;
#pragma unroll
      for (int k = 0; k < RPL; ++k) {
        if (lane + GQ_WAVE * k < B) {
          float bst = rs[k][it];
          if (it > 0) {
            bst = run[k];
            if (rs[k][it] < bst) {
              bst = rs[k][it];
              any = true;
            }
          }
          run[k] = bst;
          mx = gq_nanmax(mx, bst);
          mn = gq_nanmin(mn, ms[k][it]);
        }
      }
      const bool any_w = __ballot(any) != 0ull;
      const float mxw = -gq_dpp_nanmin(-mx);  // NaN-propagating max
      const float mnw = gq_dpp_nanmin(mn);
      not_improved = (it == 0) ? 0 : (any_w ? 0 : not_improved + 1);
      if ((not_improved == lim) || (mxw < eps) || (mnw > 1e32f)) {
        stop_at = it;
        done = true;
      }
    }
  }
  return stop_at;
}

// RPL > 0: the stop rule is replayed here (B <= 64*RPL, max_iter <= 16); RPL == 0: k* was written by a stop launch.
// one wavefront = one row; sh: nz*3 floats of LDS (per-column gradient contributions | x for x_sum)
template <int NC, int RPL>
__device__ __forceinline__ void gq_fc_tail_body(const GqFcStepArgs& g, int row, float* sh) {
  const int lane = gq_lane();
  const int nz = g.nz;
  // loads that depend on nothing computed here go first: the row's columns of F and the E_dis part of the contact
  // gradient that the E_fc part is added to at the very end
  GqLr<6, NC> S;
  S.ridge = g.ridge;
  bool live[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane + GQ_WAVE * c;
    live[c] = i < nz;
#pragma unroll
    for (int q = 0; q < 6; ++q) S.a[c][q] = live[c] ? g.F[((size_t)row * 6 + q) * nz + i] : 0.0f;
  }
  gq3 gc0 = gq_mk(0, 0, 0);
  if (lane < g.n) {
    const float* o = g.g_cpts + ((size_t)row * g.n + lane) * 3;
    gc0 = gq_mk(o[0], o[1], o[2]);
  }
  int ks;
  if (RPL > 0) {
    ks = gq_qp_stop_rows<(RPL > 0 ? RPL : 1)>(g.resid, g.mu_tab, g.B, g.max_iter, g.eps, g.not_improved_lim, lane);
    if (row == 0 && lane == 0) {
      g.kstar[0] = ks;
      g.kstar[1] = ks + 1;
      if (g.n_iter) *g.n_iter = ks + 1;
    }
  } else {
    ks = g.kstar[0];
  }
  // best iterate of this row among iterations 0..k* (qpth returns the per-row best, not the last): the row's
  // residuals are fetched together (lane it holds iteration it), then scanned from registers
  int bi = 0;
  {
    const float mine = (lane < g.max_iter && lane < 64) ? g.resid[(size_t)row * g.max_iter + lane] : 0.0f;
    float bst = 0.0f;
    for (int it = 0; it <= ks; ++it) {
      const float rs = it < 64 ? gq_readlane(mine, it) : g.resid[(size_t)row * g.max_iter + it];
      if (it == 0 || rs < bst) {
        bst = rs;
        bi = it;
      }
    }
  }
  const float* sn = g.snap + (((size_t)row * g.max_iter + bi) * 5) * nz;
  float x[NC], du[NC], dl[NC], lam[NC];
  double part[21];
#pragma unroll
  for (int i = 0; i < 21; ++i) part[i] = 0.0;
  float r[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane + GQ_WAVE * c;
    x[c] = 0.0f;
    du[c] = dl[c] = 1.0f;
    if (live[c]) {
      x[c] = sn[i];
      const float zu = sn[nz + i], zl = sn[2 * nz + i], su = sn[3 * nz + i], sl = sn[4 * nz + i];
      du[c] = fmaxf(zu, 1e-8f) / fmaxf(su, 1e-8f);
      dl[c] = fmaxf(zl, 1e-8f) / fmaxf(sl, 1e-8f);
      g.x[(size_t)row * nz + i] = x[c];
    }
    lam[c] = g.ridge + du[c] + dl[c];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      r[a] = fmaf(S.a[c][a], x[c], r[a]);
#pragma unroll
      for (int b = 0; b <= a; ++b) part[a * (a + 1) / 2 + b] += (double)S.a[c][a] * (double)S.a[c][b];
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) r[a] = gq_dpp_sum(r[a]);  // F x
  gq_wave_sums_d<21>(part);                             // F F'
  double Lm[21], inv[6];
  const bool ok = gq_chol6(part, Lm, inv);
  double lp = 1.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) lp *= Lm[i * (i + 1) / 2 + i];
  const float svd = ok ? powf((float)lp, 1.0f / 6.0f) : 0.0f;  // (prod sigma)^(1/6) = det(F F')^(1/12)
  float val = 0.0f;
#pragma unroll
  for (int a = 0; a < 6; ++a) val = fmaf(r[a], r[a], val);
  val *= 0.5f;
  const float ex = expf(-g.svd_gain * svd);
  if (lane == 0) {
    g.val[row] = val;
    g.svd[row] = svd;
    g.e_fc[row] = g.values_gain * (val + g.eps_add) * ex;
  }
  if (g.x_sum) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (live[c]) sh[lane + GQ_WAVE * c] = x[c];
    gq_wave_sync();
    for (int c = lane; c < g.n; c += GQ_WAVE) {
      float s = 0.0f;
      for (int e = 0; e < g.k; ++e) s += sh[c * g.k + e];
      g.x_sum[(size_t)row * g.n + c] = s;
    }
    gq_wave_sync();
  }
  // ---- backward with upstream gradient w_fc on E_fc ---------------------------------------------------------------
  const float gval = g.w_fc * g.values_gain * ex;
  const float gsvd = ok ? g.w_fc * g.values_gain * (val + g.eps_add) * ex * (-g.svd_gain) : 0.0f;
  // QP backward (qpth QPFunction.backward): dx = -(Q + diag(d_u + d_l))^-1 dl/dx with dl/dx = gval * F'(F x)
  float rhs[NC], dx[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float ftr = 0.0f;
#pragma unroll
    for (int q = 0; q < 6; ++q) ftr = fmaf(S.a[c][q], r[q], ftr);
    rhs[c] = live[c] ? -gval * ftr : 0.0f;
  }
  S.factor(lam, live);
  S.solve(rhs, dx);
  float fd[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int a = 0; a < 6; ++a) fd[a] = fmaf(S.a[c][a], live[c] ? dx[c] : 0.0f, fd[a]);
#pragma unroll
  for (int a = 0; a < 6; ++a) fd[a] = gq_dpp_sum(fd[a]);  // F dx
  const float s6 = gsvd * svd / 6.0f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (live[c]) {
      const int i = lane + GQ_WAVE * c;
      // d(prod sigma^(1/6))/dF = svd/6 * (F F')^-1 F : solve L L' w = f_i
      double w[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) w[a] = (double)S.a[c][a];
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
      // gradient wrt the torque rows of column i (rows 3..5); tau = tw (r x f) -> d/dr = tw (f x g_tau)
      gq3 gt;
      gt.x = (gval * r[3] + fd[3]) * x[c] + r[3] * dx[c] + s6 * (float)w[3];
      gt.y = (gval * r[4] + fd[4]) * x[c] + r[4] * dx[c] + s6 * (float)w[4];
      gt.z = (gval * r[5] + fd[5]) * x[c] + r[5] * dx[c] + s6 * (float)w[5];
      const gq3 gp = g.tw * gq_cross(gq_mk(S.a[c][0], S.a[c][1], S.a[c][2]), gt);
      sh[i * 3] = gp.x;
      sh[i * 3 + 1] = gp.y;
      sh[i * 3 + 2] = gp.z;
    }
  }
  gq_wave_sync();
  for (int c = lane; c < g.n; c += GQ_WAVE) {
    float sx = 0, sy = 0, sz = 0;
    for (int e = 0; e < g.k; ++e) {
      sx += sh[(c * g.k + e) * 3];
      sy += sh[(c * g.k + e) * 3 + 1];
      sz += sh[(c * g.k + e) * 3 + 2];
    }
    float* o = g.g_cpts + ((size_t)row * g.n + c) * 3;
    if (c < GQ_WAVE) {  // c == lane: the E_dis part was fetched at the top
      o[0] = gc0.x + sx;
      o[1] = gc0.y + sy;
      o[2] = gc0.z + sz;
    } else {
      o[0] += sx;
      o[1] += sy;
      o[2] += sz;
    }
  }
}


// ---- host side: argument block of the fused step (parameters of gq_fc_step) ----------------------------------------
int gq_qp_tables_(void* workspace, size_t workspace_bytes, int B, int nz, int max_iter, float** resid, float** mu,
                  float** snap, float** runmin, int** kstar, unsigned** agg);

static inline int gq_fc_step_fill(const float* dist_sq, const int32_t* sign, const float* obj_dir, const float* closest,
                                  const float* contact_pts, const float* hand_normals, const float* cog, int64_t batch,
                                  int n_contact, int n_cone, float friction, float torque_weight, float max_limit,
                                  float svd_gain, float values_gain, float eps, int max_iter, float w_dis, float w_fc,
                                  float* obj_normal, float* g_contact_pts, float* g_hand_normals, float* e_fc,
                                  float* x_sum, int32_t* n_iter, void* workspace, size_t workspace_bytes,
                                  GqFcStepArgs* out, float** runmin) {
  GQ_REQUIRE(dist_sq && sign && obj_dir && closest && contact_pts && hand_normals && cog && obj_normal &&
                 g_contact_pts && g_hand_normals && e_fc && workspace,
             "fc_step: null pointer");
  GQ_REQUIRE(batch > 0 && n_contact > 0 && n_cone > 0, "fc_step: bad sizes");
  const int nz = n_contact * n_cone;
  GQ_REQUIRE(nz <= 128, "fc_step: n_contact * n_cone = %d exceeds 128", nz);
  GQ_REQUIRE(max_iter >= 1 && max_iter <= 64, "fc_step: max_iter=%d out of range", max_iter);
  size_t need = 0;
  int rc = gq_fc_workspace_bytes(batch, n_contact, n_cone, max_iter, &need);
  if (rc) return rc;
  GQ_REQUIRE(workspace_bytes >= need, "fc_step: workspace too small (%zu < %zu)", workspace_bytes, need);
  GqFcWs w = gq_fc_carve(workspace, (size_t)batch, (size_t)nz, workspace_bytes);
  GqFcStepArgs a{};
  rc = gq_qp_tables_(w.qp, w.qp_bytes, (int)batch, nz, max_iter, &a.resid, &a.mu_tab, &a.snap, runmin, &a.kstar, &a.agg);
  if (rc) return rc;
  a.head_ctr = reinterpret_cast<unsigned*>(a.kstar) + 8;
  a.head_blocks = ((int)batch + GQ_HEAD_ROWS - 1) / GQ_HEAD_ROWS;
  if (max_iter > 16 || batch <= 4 * GQ_WAVE) a.agg = nullptr;  // stop launch | replayed per row by the tail
  a.dist_sq = dist_sq;
  a.sign = sign;
  a.onrm = obj_dir;
  a.closest = closest;
  a.cpts = contact_pts;
  a.cnrm = hand_normals;
  a.cog = cog;
  a.B = (int)batch;
  a.n = n_contact;
  a.k = n_cone;
  a.nz = nz;
  a.max_iter = max_iter;
  a.not_improved_lim = 3;
  a.mu = friction;
  a.tw = torque_weight;
  a.w_dis = w_dis;
  a.w_fc = w_fc;
  a.lower = 1.0f;  // bounds 1 <= x <= max_limit + 1, b = 0, ridge 1e-4 (span.py:348-349, qp_solver.py:101-112)
  a.upper = max_limit + 1.0f;
  a.ridge = 1e-4f;
  a.svd_gain = svd_gain;
  a.values_gain = values_gain;
  a.eps_add = 1e-2f;
  a.eps = eps;
  a.obj_normal = obj_normal;
  a.g_cpts = g_contact_pts;
  a.g_cnrm = g_hand_normals;
  a.F = w.F;
  a.n_iter = n_iter;
  a.e_fc = e_fc;
  a.val = w.val;
  a.svd = w.svd;
  a.x = w.x;
  a.x_sum = x_sum;
  *out = a;
  return GQ_OK;
}
