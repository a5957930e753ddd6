// Batched box-constrained QP (force-closure QP of GraspQP) for gfx950: one problem per wavefront.
//
// Forward = qpth-style primal-dual interior point with batch-global stopping, split in three launches
// so that no wave ever waits for another:
//   gq_qp_iter_kernel    every row runs all max_iter iterations, recording resid/mu per iteration and a
//                        snapshot (x, lam, slack) at each iteration that improves the row's best residual
//   gq_qp_stop_kernel    one block replays qpth's batch-global rule (notImproved == 3 | max best < eps |
//                        min mu > 1e32) on the (B, max_iter) residual table -> stop iteration k*
//   gq_qp_select_kernel  each row returns its best snapshot among iterations 0..k*
// Backward = one more reduced-KKT solve at the returned point (qpth QPFunction.backward).
#include "qp_kernels.h"

// ---- batch-global stopping rule (single block) -----------------------------------------------------------
__global__ __launch_bounds__(256) void gq_qp_stop_kernel(const float* __restrict__ resid, const float* __restrict__ mu,
                                                         int B, int max_iter, float eps, int not_improved_lim,
                                                         float* __restrict__ runmin, int* __restrict__ kstar,
                                                         int* __restrict__ n_iter_out) {
  __shared__ int s_any;
  __shared__ float s_red[256];
  __shared__ float s_red2[256];
  __shared__ int s_stop;
  const int tid = threadIdx.x;
  int not_improved = 0;
  int stop_at = max_iter - 1;
  for (int it = 0; it < max_iter; ++it) {
    if (tid == 0) s_any = 0;
    __syncthreads();
    float mx = -GQ_INF, mn = GQ_INF;
    int any = 0;
    for (int r = tid; r < B; r += 256) {
      const float rs = resid[(size_t)r * max_iter + it];
      float bst;
      if (it == 0) {
        bst = rs;
      } else {
        bst = runmin[r];
        if (rs < bst) {
          bst = rs;
          any = 1;
        }
      }
      runmin[r] = bst;
      mx = gq_nanmax(mx, bst);
      mn = gq_nanmin(mn, mu[(size_t)r * max_iter + it]);
    }
    if (any) atomicOr(&s_any, 1);
    s_red[tid] = mx;
    s_red2[tid] = mn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        s_red[tid] = gq_nanmax(s_red[tid], s_red[tid + o]);
        s_red2[tid] = gq_nanmin(s_red2[tid], s_red2[tid + o]);
      }
      __syncthreads();
    }
    if (tid == 0) {
      if (it == 0) not_improved = 0;
      else not_improved = s_any ? 0 : not_improved + 1;
      s_stop = (not_improved == not_improved_lim) || (s_red[0] < eps) || (s_red2[0] > 1e32f);
    }
    __syncthreads();
    if (s_stop) {
      stop_at = it;
      break;
    }
  }
  if (tid == 0) {
    kstar[0] = stop_at;      // last iteration whose record counts
    kstar[1] = stop_at + 1;  // qpth-style iteration count
    if (n_iter_out) *n_iter_out = stop_at + 1;
  }
}

// same rule, one wavefront (B <= 1024): the (B, max_iter) residual / mu tables are staged in LDS with all loads in
// flight at once, then replayed iteration by iteration with DPP reductions -- no block barriers, no dependent global
// round trips.
#define GQ_STOP_MAXB 1024
#define GQ_STOP_MAXIT 16
__global__ __launch_bounds__(GQ_WAVE) void gq_qp_stop_wave_kernel(const float* __restrict__ resid,
                                                                  const float* __restrict__ mu, int B, int max_iter,
                                                                  float eps, int not_improved_lim,
                                                                  float* __restrict__ runmin, int* __restrict__ kstar,
                                                                  int* __restrict__ n_iter_out) {
  __shared__ float s_res[GQ_STOP_MAXB * GQ_STOP_MAXIT];
  __shared__ float s_mu[GQ_STOP_MAXB * GQ_STOP_MAXIT];
  const int lane = gq_lane();
  const int tot = B * max_iter;
  for (int i = lane; i < tot; i += GQ_WAVE) {
    s_res[i] = resid[i];
    s_mu[i] = mu[i];
  }
  __syncthreads();
  int not_improved = 0;
  int stop_at = max_iter - 1;
  float run[GQ_STOP_MAXB / GQ_WAVE];  // running best residual of my rows (lane, lane+64, ...)
  for (int it = 0; it < max_iter; ++it) {
    float mx = -GQ_INF, mn = GQ_INF;
    bool any = false;
#pragma unroll
    for (int k = 0; k < GQ_STOP_MAXB / GQ_WAVE; ++k) {
      const int r = lane + GQ_WAVE * k;
      if (r < B) {
        const float rs = s_res[r * max_iter + it];
        float bst = rs;
        if (it > 0) {
          bst = run[k];
          if (rs < bst) {
            bst = rs;
            any = true;
          }
        }
        run[k] = bst;
        mx = gq_nanmax(mx, bst);
        mn = gq_nanmin(mn, s_mu[r * max_iter + it]);
      }
    }
    const bool any_w = __ballot(any) != 0ull;
    const float mxw = -gq_dpp_nanmin(-mx);  // NaN-propagating max
    const float mnw = gq_dpp_nanmin(mn);
    not_improved = (it == 0) ? 0 : (any_w ? 0 : not_improved + 1);
    if ((not_improved == not_improved_lim) || (mxw < eps) || (mnw > 1e32f)) {
      stop_at = it;
      break;
    }
  }
  if (lane == 0) {
    kstar[0] = stop_at;
    kstar[1] = stop_at + 1;
    if (n_iter_out) *n_iter_out = stop_at + 1;
  }
}

// same rule for any number of rows, one block of 1024 threads.  The per-row running best does not depend on where the
// batch stops, so every thread replays its rows (one per 1024-row tile) over ALL iterations and keeps, per iteration,
// "some row of mine improved" / max of the running bests / min of mu; ONE block reduction at the end yields the three
// per-iteration aggregates and thread 0 applies the sequential rule to them.  The (rows, max_iter) tables are read as
// contiguous streams and transposed through LDS -- a thread reading "its" row directly costs one cache line per lane
// and load instruction, which made the previous variant L1-bound (33 us at 2048 rows).
__global__ __launch_bounds__(1024) void gq_qp_stop_tiled_kernel(const float* __restrict__ resid,
                                                                const float* __restrict__ mu, int B, int max_iter,
                                                                float eps, int not_improved_lim, int* __restrict__ kstar,
                                                                int* __restrict__ n_iter_out) {
  __shared__ float s_tab[1024 * GQ_STOP_MAXIT];
  __shared__ float s_mx[16][GQ_STOP_MAXIT], s_mn[16][GQ_STOP_MAXIT];
  __shared__ unsigned s_any[16];
  const int tid = threadIdx.x, lane = gq_lane(), wv = tid / GQ_WAVE;
  float amx[GQ_STOP_MAXIT], amn[GQ_STOP_MAXIT];
  unsigned aany = 0u;
#pragma unroll
  for (int it = 0; it < GQ_STOP_MAXIT; ++it) {
    amx[it] = -GQ_INF;
    amn[it] = GQ_INF;
  }
  for (int r0 = 0; r0 < B; r0 += 1024) {
    const int nr = (B - r0 < 1024) ? B - r0 : 1024;
    const int tot = nr * max_iter;
    const bool mine = tid < nr;
    float rs[GQ_STOP_MAXIT];
    // both tiles are requested before anything is stored (a copy loop would wait for every load in turn)
    float va[GQ_STOP_MAXIT], vb[GQ_STOP_MAXIT];
    const float* src_a = resid + (size_t)r0 * max_iter;
    const float* src_b = mu + (size_t)r0 * max_iter;
#pragma unroll
    for (int j = 0; j < GQ_STOP_MAXIT; ++j) {
      const int i = tid + 1024 * j;
      va[j] = i < tot ? src_a[i] : 0.0f;
      vb[j] = i < tot ? src_b[i] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < GQ_STOP_MAXIT; ++j)
      if (tid + 1024 * j < tot) s_tab[tid + 1024 * j] = va[j];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < GQ_STOP_MAXIT; ++it) rs[it] = (mine && it < max_iter) ? s_tab[tid * max_iter + it] : 0.0f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GQ_STOP_MAXIT; ++j)
      if (tid + 1024 * j < tot) s_tab[tid + 1024 * j] = vb[j];
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int it = 0; it < GQ_STOP_MAXIT; ++it)
        if (it < max_iter) amn[it] = gq_nanmin(amn[it], s_tab[tid * max_iter + it]);
      float bst = rs[0];
      amx[0] = gq_nanmax(amx[0], bst);
#pragma unroll
      for (int it = 1; it < GQ_STOP_MAXIT; ++it) {
        if (it < max_iter) {
          if (rs[it] < bst) {
            bst = rs[it];
            aany |= 1u << it;
          }
          amx[it] = gq_nanmax(amx[it], bst);
        }
      }
    }
    __syncthreads();  // the next tile overwrites s_tab
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) aany |= (unsigned)__shfl_xor((int)aany, o, GQ_WAVE);
#pragma unroll
  for (int it = 0; it < GQ_STOP_MAXIT; ++it) {
    const float mxw = -gq_dpp_nanmin(-amx[it]), mnw = gq_dpp_nanmin(amn[it]);
    if (lane == 0) {
      s_mx[wv][it] = mxw;
      s_mn[wv][it] = mnw;
    }
  }
  if (lane == 0) s_any[wv] = aany;
  __syncthreads();
  if (tid == 0) {
    unsigned any = 0u;
    for (int w = 0; w < 16; ++w) any |= s_any[w];
    int not_improved = 0, stop_at = max_iter - 1;
    for (int it = 0; it < max_iter; ++it) {
      float bmx = -GQ_INF, bmn = GQ_INF;
      for (int w = 0; w < 16; ++w) {
        bmx = gq_nanmax(bmx, s_mx[w][it]);
        bmn = gq_nanmin(bmn, s_mn[w][it]);
      }
      not_improved = (it == 0) ? 0 : (((any >> it) & 1u) ? 0 : not_improved + 1);
      if ((not_improved == not_improved_lim) || (bmx < eps) || (bmn > 1e32f)) {
        stop_at = it;
        break;
      }
    }
    kstar[0] = stop_at;
    kstar[1] = stop_at + 1;
    if (n_iter_out) *n_iter_out = stop_at + 1;
  }
}

__global__ __launch_bounds__(GQ_WAVE) void gq_qp_select_kernel(const float* __restrict__ resid,
                                                               const float* __restrict__ snap,
                                                               const int* __restrict__ kstar, int B, int nz,
                                                               int max_iter, float* __restrict__ x,
                                                               float* __restrict__ lam, float* __restrict__ slack,
                                                               int* __restrict__ best_iter) {
  const int row = blockIdx.x;
  const int lane = gq_lane();
  const int ks = kstar[0];
  float bst = 0.0f;
  int bi = 0;
  for (int it = 0; it <= ks; ++it) {
    const float rs = resid[(size_t)row * max_iter + it];
    if (it == 0 || rs < bst) {
      bst = rs;
      bi = it;
    }
  }
  for (int k = lane; k < nz; k += GQ_WAVE) {
    const float* s = snap + (((size_t)row * max_iter + bi) * 5) * nz + k;
    x[(size_t)row * nz + k] = s[0];
    lam[(size_t)row * 2 * nz + k] = s[nz];
    lam[(size_t)row * 2 * nz + nz + k] = s[2 * nz];
    slack[(size_t)row * 2 * nz + k] = s[3 * nz];
    slack[(size_t)row * 2 * nz + nz + k] = s[4 * nz];
  }
  if (lane == 0 && best_iter) best_iter[row] = bi;
}

// ---- host side ----------------------------------------------------------------------------------------------------
static size_t gq_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct GqQpWs {
  float *resid, *mu, *snap, *runmin;
  int* kstar;
  unsigned* agg;  // per-block stop-rule aggregates of the fused force-closure step (fcstep_dev.h): 36 words per 4 rows
  size_t total;
};
static GqQpWs gq_qp_carve(void* base, int B, int nz, int max_iter) {
  GqQpWs w;
  size_t off = 0;
  char* c = (char*)base;
  w.resid = (float*)(c + off);
  off += gq_align((size_t)B * max_iter * 4);
  w.mu = (float*)(c + off);
  off += gq_align((size_t)B * max_iter * 4);
  w.snap = (float*)(c + off);
  off += gq_align((size_t)B * max_iter * 5 * nz * 4);
  w.runmin = (float*)(c + off);
  off += gq_align((size_t)B * 4);
  w.kstar = (int*)(c + off);
  off += 256;
  w.agg = (unsigned*)(c + off);
  off += gq_align((size_t)((B + 3) / 4) * 36 * 4);
  w.total = off;
  return w;
}

static void gq_qp_stop_dispatch(const float* resid, const float* mu, int B, int max_iter, float eps, int lim,
                                float* runmin, int* kstar, int32_t* n_iter, hipStream_t st) {
  if (B <= GQ_STOP_MAXB && max_iter <= GQ_STOP_MAXIT)
    hipLaunchKernelGGL(gq_qp_stop_wave_kernel, dim3(1), dim3(GQ_WAVE), 0, st, resid, mu, B, max_iter, eps, lim, runmin,
                       kstar, n_iter);
  else if (max_iter <= GQ_STOP_MAXIT)
    hipLaunchKernelGGL(gq_qp_stop_tiled_kernel, dim3(1), dim3(1024), 0, st, resid, mu, B, max_iter, eps, lim, kstar,
                       n_iter);
  else
    hipLaunchKernelGGL(gq_qp_stop_kernel, dim3(1), dim3(256), 0, st, resid, mu, B, max_iter, eps, lim, runmin, kstar,
                       n_iter);
}

int gq_qp_launch_iter_dense_lds(const GqQpArgs& a, hipStream_t st);    // qp_dense.hip: dense Q, 65..128 variables
int gq_qp_launch_bwd_dense_lds(const GqQpBwdArgs& a, hipStream_t st);

static int gq_launch_iter(const GqQpArgs& a, int mode, hipStream_t st) {
  if (mode == 0) return gq_qp_lr_launch_iter(a, st);
  if (a.nz > 64) return gq_qp_launch_iter_dense_lds(a, st);
  if (a.nz <= 16) return gq_qp_launch_iter_16(a, mode, st);
  if (a.nz <= 32) return gq_qp_launch_iter_32(a, mode, st);
  if (a.nz <= 48) return gq_qp_launch_iter_48(a, mode, st);
  return gq_qp_launch_iter_64(a, mode, st);
}
static int gq_launch_bwd(const GqQpBwdArgs& a, int mode, hipStream_t st) {
  if (mode == 0) return gq_qp_lr_launch_bwd(a, st);
  if (a.nz > 64) return gq_qp_launch_bwd_dense_lds(a, st);
  if (a.nz <= 16) return gq_qp_launch_bwd_16(a, mode, st);
  if (a.nz <= 32) return gq_qp_launch_bwd_32(a, mode, st);
  if (a.nz <= 48) return gq_qp_launch_bwd_48(a, mode, st);
  return gq_qp_launch_bwd_64(a, mode, st);
}

static int gq_qp_forward_common(GqQpArgs a, float eps, int not_improved_lim, float* x, float* lam, float* slack,
                                int* best_iter, int* n_iter, void* ws, size_t ws_bytes, hipStream_t st, int mode) {
  GQ_REQUIRE(a.B > 0 && a.nz > 0, "boxqp: empty batch (B=%d nz=%d)", a.B, a.nz);
  GQ_REQUIRE(a.nz <= 128, "boxqp: nz=%d exceeds the supported size (128)", a.nz);
  GQ_REQUIRE(a.max_iter >= 1 && a.max_iter <= 64, "boxqp: max_iter=%d out of range", a.max_iter);
  GQ_REQUIRE(mode == 1 || (a.m >= 1 && a.m <= 8), "boxqp: m=%d must be in [1,8]", a.m);
  GQ_REQUIRE(ws, "boxqp: null workspace pointer");
  GqQpWs w = gq_qp_carve(ws, a.B, a.nz, a.max_iter);
  GQ_REQUIRE(ws_bytes >= w.total, "boxqp: workspace too small (%zu < %zu)", ws_bytes, w.total);
  a.resid = w.resid;
  a.mu = w.mu;
  a.snap = w.snap;
  int rc = gq_launch_iter(a, mode, st);
  if (rc) return rc;
  gq_qp_stop_dispatch(w.resid, w.mu, a.B, a.max_iter, eps, not_improved_lim, w.runmin, w.kstar, n_iter, st);
  GQ_LAUNCH_CHECK();
  if (x == nullptr) return GQ_OK;  // internal callers (fc.hip) select the best iterate inside their own kernel
  hipLaunchKernelGGL(gq_qp_select_kernel, dim3(a.B), dim3(GQ_WAVE), 0, st, w.resid, w.snap, w.kstar, a.B, a.nz,
                     a.max_iter, x, lam, slack, best_iter);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// internal (fcstep.hip): table pointers inside a box-QP workspace, and the stand-alone stop-rule launch
int gq_qp_tables_(void* workspace, size_t workspace_bytes, int B, int nz, int max_iter, float** resid, float** mu,
                  float** snap, float** runmin, int** kstar, unsigned** agg) {
  GQ_REQUIRE(workspace && B > 0 && nz > 0 && max_iter >= 1 && max_iter <= 64, "qp_tables: bad arguments");
  GqQpWs w = gq_qp_carve(workspace, B, nz, max_iter);
  GQ_REQUIRE(workspace_bytes >= w.total, "boxqp: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  *resid = w.resid;
  *mu = w.mu;
  *snap = w.snap;
  *runmin = w.runmin;
  *kstar = w.kstar;
  *agg = w.agg;
  return GQ_OK;
}
int gq_qp_stop_launch_(const float* resid, const float* mu, int B, int max_iter, float eps, int not_improved_lim,
                       float* runmin, int* kstar, int32_t* n_iter, void* stream) {
  gq_qp_stop_dispatch(resid, mu, B, max_iter, eps, not_improved_lim, runmin, kstar, n_iter, (hipStream_t)stream);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// internal (fc.hip): run the iterations and the stop rule only; the caller picks each row's best iterate itself
int gq_lsq_boxqp_iterate_(const float* A, float lower_s, float upper_s, int64_t batch, int m, int nz, float ridge,
                          float eps, int max_iter, int32_t* n_iter, void* workspace, size_t workspace_bytes,
                          void* stream, const float** resid, const float** snap, const int** kstar) {
  GqQpArgs a{};
  a.A = A;
  a.lower_s = lower_s;
  a.upper_s = upper_s;
  a.ridge = ridge;
  a.B = (int)batch;
  a.m = m;
  a.nz = nz;
  a.max_iter = max_iter;
  int rc = gq_qp_forward_common(a, eps, 3, nullptr, nullptr, nullptr, nullptr, n_iter, workspace, workspace_bytes,
                                (hipStream_t)stream, 0);
  if (rc) return rc;
  GqQpWs w = gq_qp_carve(workspace, a.B, a.nz, a.max_iter);
  *resid = w.resid;
  *snap = w.snap;
  *kstar = w.kstar;
  return GQ_OK;
}

int gq_lsq_boxqp_backward_scaled_(const float* A, const float* lam, const float* slack, const float* grad_x,
                                  int64_t batch, int m, int nz, float ridge, float* dx, float* dlam,
                                  const float* scale_ge, const float* scale_svd, float svd_gain, float values_gain,
                                  void* stream) {
  GqQpBwdArgs a{};
  a.A = A;
  a.lam = lam;
  a.slack = slack;
  a.grad_x = grad_x;
  a.ridge = ridge;
  a.B = (int)batch;
  a.m = m;
  a.nz = nz;
  a.dx = dx;
  a.dlam = dlam;
  a.scale_ge = scale_ge;
  a.scale_svd = scale_svd;
  a.svd_gain = svd_gain;
  a.values_gain = values_gain;
  return gq_launch_bwd(a, 0, (hipStream_t)stream);
}

extern "C" {

int gq_boxqp_workspace_bytes(int64_t batch, int nz, int max_iter, size_t* bytes) {
  GQ_REQUIRE(bytes != nullptr && batch >= 0 && nz > 0 && max_iter > 0, "boxqp_workspace_bytes: bad arguments");
  *bytes = gq_qp_carve(nullptr, (int)batch, nz, max_iter).total;
  return GQ_OK;
}

int gq_lsq_boxqp_forward(const float* A, const float* b, const float* lower, const float* upper, float lower_s,
                         float upper_s, int64_t batch, int m, int nz, float ridge, float eps, int max_iter,
                         int not_improved_lim, float* x, float* lam, float* slack, int32_t* best_iter,
                         int32_t* n_iter, void* workspace, size_t workspace_bytes, void* stream) {
  GQ_REQUIRE(A != nullptr, "lsq_boxqp_forward: A is null");
  GqQpArgs a{};
  a.A = A;
  a.b = b;
  a.lower = lower;
  a.upper = upper;
  a.lower_s = lower_s;
  a.upper_s = upper_s;
  a.ridge = ridge;
  a.B = (int)batch;
  a.m = m;
  a.nz = nz;
  a.max_iter = max_iter;
  return gq_qp_forward_common(a, eps, not_improved_lim, x, lam, slack, best_iter, n_iter, workspace, workspace_bytes,
                              (hipStream_t)stream, 0);
}

int gq_boxqp_forward(const float* Q, const float* p, const float* lower, const float* upper, float lower_s,
                     float upper_s, int64_t batch, int nz, float eps, int max_iter, int not_improved_lim, float* x,
                     float* lam, float* slack, int32_t* best_iter, int32_t* n_iter, void* workspace,
                     size_t workspace_bytes, void* stream) {
  GQ_REQUIRE(Q != nullptr, "boxqp_forward: Q is null");
  GqQpArgs a{};
  a.Q = Q;
  a.p = p;
  a.lower = lower;
  a.upper = upper;
  a.lower_s = lower_s;
  a.upper_s = upper_s;
  a.B = (int)batch;
  a.m = 0;
  a.nz = nz;
  a.max_iter = max_iter;
  return gq_qp_forward_common(a, eps, not_improved_lim, x, lam, slack, best_iter, n_iter, workspace, workspace_bytes,
                              (hipStream_t)stream, 1);
}

// qpth's batch-global stop rule on a (B, max_iter) residual / mu table (what the forward entry points run after
// their iterations); exposed so that the rule itself can be tested on arbitrary tables.
int gq_boxqp_stop_rule(const float* resid, const float* mu, int64_t batch, int max_iter, float eps,
                       int not_improved_lim, float* runmin_scratch, int32_t* kstar, int32_t* n_iter, void* stream) {
  GQ_REQUIRE(resid && mu && runmin_scratch && kstar, "boxqp_stop_rule: null pointer");
  GQ_REQUIRE(batch > 0 && batch < (1ll << 31) && max_iter >= 1 && max_iter <= 64, "boxqp_stop_rule: bad sizes");
  return gq_qp_stop_launch_(resid, mu, (int)batch, max_iter, eps, not_improved_lim, runmin_scratch, kstar, n_iter, stream);
}

int gq_lsq_boxqp_backward(const float* A, const float* lam, const float* slack, const float* grad_x, int64_t batch,
                          int m, int nz, float ridge, float* dx, float* dlam, void* stream) {
  GQ_REQUIRE(A && lam && slack && grad_x && dx && dlam, "lsq_boxqp_backward: null pointer");
  GQ_REQUIRE(batch > 0 && nz > 0 && nz <= 128 && m >= 1 && m <= 8, "lsq_boxqp_backward: bad sizes B=%lld m=%d nz=%d",
             (long long)batch, m, nz);
  GqQpBwdArgs a{};
  a.A = A;
  a.lam = lam;
  a.slack = slack;
  a.grad_x = grad_x;
  a.ridge = ridge;
  a.B = (int)batch;
  a.m = m;
  a.nz = nz;
  a.dx = dx;
  a.dlam = dlam;
  return gq_launch_bwd(a, 0, (hipStream_t)stream);
}

int gq_boxqp_backward(const float* Q, const float* lam, const float* slack, const float* grad_x, int64_t batch, int nz,
                      float* dx, float* dlam, void* stream) {
  GQ_REQUIRE(Q && lam && slack && grad_x && dx && dlam, "boxqp_backward: null pointer");
  GQ_REQUIRE(batch > 0 && nz > 0 && nz <= 128, "boxqp_backward: bad sizes B=%lld nz=%d", (long long)batch, nz);
  GqQpBwdArgs a{};
  a.Q = Q;
  a.lam = lam;
  a.slack = slack;
  a.grad_x = grad_x;
  a.B = (int)batch;
  a.m = 0;
  a.nz = nz;
  a.dx = dx;
  a.dlam = dlam;
  return gq_launch_bwd(a, 1, (hipStream_t)stream);
}

}  // extern "C"
