// Mesh signed-distance queries with the TorchSDF output contract (reference call sites object_model.py:220,
// hand_model.py:953) and the fused hand-penetration query (hand_model.py:875-987).
//
// Two kernel shapes cover the two regimes of the grasp loop:
//   * "wave per query"  (contacts vs object mesh: few queries, 1e3-2e4 faces): the 64 lanes of a wavefront stride
//     over the face records (one 64-byte record per lane -> four coalesced 16-byte loads), keep a running
//     (dist^2, face) minimum and reduce it with a 64-bit key so the lowest face index wins ties.
//   * "point per lane"  (object surface points vs hand-link meshes: ~1e6 queries, 2e2-1e3 faces per link): every
//     lane owns one point; the face records are wave-uniform so they travel through the scalar cache / SGPRs
//     and the per-lane work is pure FP32 VALU.
#include "tri.h"
#include <hip/hip_ext.h>

// rec[i] = record of face perm[i] (perm == nullptr: identity)
__global__ void gq_face_prep_kernel(const float* __restrict__ fv, const int32_t* __restrict__ perm,
                                    GqFace* __restrict__ rec, int64_t F) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  const int64_t src = perm ? perm[i] : i;
  const float* v = fv + src * 9;
  rec[i] = gq_make_face(gq_mk(v[0], v[1], v[2]), gq_mk(v[3], v[4], v[5]), gq_mk(v[6], v[7], v[8]), (int)src);
}

#include "sdf_dev.h"

template <int TOPK>  // 4: <= 128 VGPRs, 4 wavefronts per SIMD; 2 (large launches): see gq_sdf_wave_query
__global__ __launch_bounds__(256, TOPK == 4 ? 4 : 5) void gq_sdf_wave_kernel(GqWaveArgs g) {
  int64_t q = (int64_t)blockIdx.x * (blockDim.x / GQ_WAVE) + (threadIdx.x / GQ_WAVE);
  if (g.xcd_meshes) {  // block -> (XCD slot, mesh of that slot, block inside the mesh); see GqWaveArgs
    const int b = (int)blockIdx.x, x = b & 7, i = b >> 3;
    const int mesh = x + 8 * (i / g.blocks_per_mesh);
    const int64_t in_mesh = (int64_t)(i % g.blocks_per_mesh) * (blockDim.x / GQ_WAVE) + (threadIdx.x / GQ_WAVE);
    if (mesh >= g.n_mesh || in_mesh >= g.queries_per_mesh) return;
    q = (int64_t)mesh * g.queries_per_mesh + in_mesh;
  }
  if (q >= g.N) return;
  const int lane = gq_lane();
  const GqSdfPre pre = gq_sdf_wave_prefetch(g, q, lane);
  gq_sdf_wave_query<TOPK>(g, q, gq_mk(g.points[q * 3 + 0], g.points[q * 3 + 1], g.points[q * 3 + 2]), lane, pre);
}
static int gq_sdf_topk_ = 0;  // gq_debug_set_sdf_topk: 0 = default (4), 2 / 4 forced (A/B runs)
static void gq_sdf_wave_launch(const GqWaveArgs& w, unsigned blocks, int64_t n_queries, hipStream_t st) {
  // two clusters per round: 86 instead of 128 VGPRs and 6 % fewer cluster visits, but every query lives longer --
  // -1.4 % at 2048 / 4096 rows, +1 % at 512 (tools: bench.py --sdf_topk): four stays the default at every size
  const int topk = gq_sdf_topk_ ? gq_sdf_topk_ : 4;
  (void)n_queries;
  if (topk == 2) hipLaunchKernelGGL(gq_sdf_wave_kernel<2>, dim3(blocks), dim3(256), 0, st, w);
  else hipLaunchKernelGGL(gq_sdf_wave_kernel<4>, dim3(blocks), dim3(256), 0, st, w);
}

// ---- point per lane -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gq_sdf_points_kernel(const float* __restrict__ points, int64_t N,
                                                            const GqFace* __restrict__ rec, int F,
                                                            float* __restrict__ dist_sq, int32_t* __restrict__ sign,
                                                            float* __restrict__ normal, float* __restrict__ closest) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = q < N;
  const int64_t qq = ok ? q : 0;
  const gq3 p = gq_mk(points[qq * 3 + 0], points[qq * 3 + 1], points[qq * 3 + 2]);
  float best = GQ_INF_F;
  int bi = 0;
  for (int f = 0; f < F; ++f) {  // f is wave-uniform: the record is fetched once per wave (scalar loads)
    const GqFace fc = rec[f];
    const gq3 d = p - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z);
    const float d2 = gq_tri_rank(fc, d);
    if (d2 < best) {
      best = d2;
      bi = f;
    }
  }
  if (!ok) return;
  const GqSdfOut o = gq_tri_finish(rec[bi], p);
  dist_sq[q] = o.dist2;
  sign[q] = o.sign;
  if (normal) {
    normal[q * 3 + 0] = o.normal.x;
    normal[q * 3 + 1] = o.normal.y;
    normal[q * 3 + 2] = o.normal.z;
  }
  closest[q * 3 + 0] = o.closest.x;
  closest[q * 3 + 1] = o.closest.y;
  closest[q * 3 + 2] = o.closest.z;
}

// d(dist_sq)/d(points) = 2 (p - closest) * g
__global__ void gq_sdf_bwd_kernel(const float* __restrict__ g, const float* __restrict__ points,
                                  const float* __restrict__ closest, int64_t N, float* __restrict__ grad_points) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 3) return;
  grad_points[i] = 2.0f * (points[i] - closest[i]) * g[i / 3];
}

#include "pen_dev.h"

template <bool EVAL, int ECAP, int ICAP, int PPT = 1>
__global__ __launch_bounds__(256) void gq_pen_grid_kernel(GqPenArgs g) {
  extern __shared__ char gq_lds[];
  // slice-major grid (x = row, y = slice block): the blocks dispatched last are those of the last, partly filled slice of
  // every row rather than all slices of the last rows -- the late starters decide when the launch ends
  gq_pen_grid_body<EVAL, ECAP, ICAP, PPT>(g, (int)blockIdx.y, (int)blockIdx.x, gq_lds);
}
__global__ __launch_bounds__(256) void gq_pen_cells_kernel(GqPenArgs g) {
  extern __shared__ char gq_lds[];
  gq_pen_cells_body(g, (int)blockIdx.x, gq_lds);
}
template <int K>
__global__ __launch_bounds__(256) void gq_hand_pen_bwd_kernel(GqPenBwdArgs g) {
  extern __shared__ char gq_lds[];
  gq_pen_bwd_body<K>(g, (int)blockIdx.x, gq_lds);
}

// MODE 0 ("exact"): dis is the exact max over links for every point.  A link is skipped for a whole wavefront
//   only when, for every lane, its AABB lower bound already proves dis_l <= best (points outside a link's AABB
//   are outside the link, so dis_l = -sqrt(d_l^2+1e-8) <= -sqrt(lb^2+1e-8)).
// MODE 1 ("penetration only"): what E_pen needs (energy.py:59-61 zeroes dis <= 0): a link is evaluated only if
//   some lane's point lies in a voxel of the link's 32^3 occupancy grid that touches the surface or the interior;
//   dis is exact wherever it is > 0 and merely <= 0 elsewhere.  MODE 2: the same with the AABB test only.
template <int MODE>
__global__ __launch_bounds__(256) void gq_hand_pen_kernel(GqPenArgs g) {
  const int row = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (g.span && threadIdx.x == 0) gq_span_open(g.span, blockIdx.x + blockIdx.y * gridDim.x);
  const bool ok = pt < g.P;
  const int obj = row / g.batch_each;
  const float* sp = g.surf + ((size_t)obj * g.P + (ok ? pt : 0)) * 3;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  const gq3 xw = gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]);
  const gq3 xh = gq_mtv(R, xw);  // R^T (x - t)
  float best_dis = -GQ_INF_F;
  int best_link = 0;
  gq3 best_g = gq_mk(0, 0, 0);
  for (int l = 0; l < g.L; ++l) {
    const int f0 = g.off[l], f1 = g.off[l + 1];
    if (f1 <= f0) continue;
    const float* T = g.link_T + ((size_t)row * g.L + l) * 12;  // wave-uniform
    const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const gq3 tl = gq_mk(T[3], T[7], T[11]);
    const gq3 xl = gq_mtv(Rl, xh - tl);
    const float lb2 = gq_aabb_dist2(g.aabb + l * 8, xl);  // squared distance to the link's AABB
    bool need;
    if (MODE == 1 || MODE == 2) {
      need = ok && (lb2 <= 0.0f);
      if (MODE == 1 && need && g.occ) {
        // 32^3 occupancy grid over the link's AABB: a point in a voxel that neither touches the surface nor lies
        // inside the mesh is outside -> this link cannot be penetrated by it
        const float* bb = g.aabb + l * 8;
        const float ux = (xl.x - bb[0]) * bb[3], uy = (xl.y - bb[1]) * bb[7];  // bb[3], bb[7]: 32/extent x, y
        const float uz = (xl.z - bb[2]) * g.occ_invz[l];
        const int ix = min(max((int)ux, 0), 31), iy = min(max((int)uy, 0), 31), iz = min(max((int)uz, 0), 31);
        need = (g.occ[(size_t)l * 1024 + iz * 32 + iy] >> ix) & 1u;
      }
    } else {
      // can link l still beat best_dis?  only if it may be penetrated (inside AABB) or closer than the best so far
      need = ok && ((lb2 <= 0.0f) || (best_dis < 0.0f && fmaf(lb2, 0.9999f, 1e-8f) < best_dis * best_dis));
    }
    if (__ballot(need) == 0ull) continue;  // wave-uniform skip
    if (g.dbg) {
      const unsigned long long nm = __ballot(need);
      if (gq_lane() == 0) {
        atomicAdd(&g.dbg[0], (unsigned long long)__popcll(nm));
        atomicAdd(&g.dbg[1], 1ull);
      }
    }
    // exact nearest face of link l for the lanes that need it: visit 16-face sub-clusters, skipping (for the whole
    // wave) those whose box is farther than every needing lane's running minimum
    float bd = GQ_INF_F;
    unsigned bo = 0xffffffffu;
    int bi = f0;
    const int s0 = g.sub_off[l], s1 = g.sub_off[l + 1];
    for (int sc = s0; sc < s1; ++sc) {
      const float lbs = gq_aabb_dist2(g.sub_aabb + (size_t)sc * 8, xl);
      if (__ballot(need && lbs * 0.9999f <= bd) == 0ull) continue;
      if (g.dbg && gq_lane() == 0) atomicAdd(&g.dbg[2], 1ull);
      const int fa = f0 + (sc - s0) * 16;
      const int fb = (fa + 16 < f1) ? fa + 16 : f1;
      GqFace cur = g.rec[fa];
      for (int f = fa; f < fb; ++f) {
        const GqFace nxt = g.rec[(f + 1 < fb) ? f + 1 : f];  // scalar prefetch of the next record
        const gq3 d = xl - gq_mk(cur.r0.x, cur.r0.y, cur.r0.z);
        const float d2 = gq_tri_rank(cur, d);
        const unsigned orig = (unsigned)__float_as_int(cur.r5.z);
        if (d2 < bd || (d2 == bd && orig < bo)) {
          bd = d2;
          bo = orig;
          bi = f;
        }
        cur = nxt;
      }
    }
    const GqSdfOut o = gq_tri_finish(g.rec[bi], xl);
    const float root = sqrtf(o.dist2 + 1e-8f);
    const float dis = root * (float)(-o.sign);
    if (need && dis > best_dis) {  // lanes that did not need this link may have skipped its nearest face
      best_dis = dis;
      best_link = l;
      // d dis / d x_l = -sign (x_l - c) / sqrt(d^2 + 1e-8); rotate into the hand frame
      const gq3 gl = ((float)(-o.sign) / root) * (xl - o.closest);
      best_g = gq_mv(Rl, gl);
    }
  }
  if (ok) {
    const size_t o = (size_t)row * g.P + pt;
    g.dis[o] = (best_dis == -GQ_INF_F) ? -1e30f : best_dis;
    g.link[o] = best_link;
    g.gvec[o * 3 + 0] = best_g.x;
    g.gvec[o * 3 + 1] = best_g.y;
    g.gvec[o * 3 + 2] = best_g.z;
  }
  if (g.span) {  // last store of the block is done: close the launch's time span
    __syncthreads();
    if (threadIdx.x == 0) gq_span_close(g.span, blockIdx.x + blockIdx.y * gridDim.x);
  }
}

// ---- load-balanced penetration-only query (penetration_only = 1 with a workspace) -------------------------------------
// Only ~3 % of the (wavefront, link) combinations contain a point that can penetrate the link, but they cluster in a
// few wavefronts (the points in the middle of the hand touch many links), so a single kernel ends on a handful of
// waves that each walk 10+ link meshes.  The query is therefore split in three launches:
//   scan      every wavefront tests its 64 points against every link (AABB + occupancy grid) and appends one work
//             item (row, first point, link, 64-bit lane mask) per hit to a queue;
//   eval      the queue is processed by 8192 wavefronts in parallel: one link mesh for <= 64 points per item; a
//             penetrating point publishes max(dis) with a 64-bit atomicMax key (dis bits | 255-link | face), which is
//             independent of the order in which items are processed;
//   finalize  points with a key recompute closest point / gradient of the winning (link, face).
struct GqPenItem {  // one surface point that may penetrate the queue's link
  int row, pt;
};
struct GqPenQ {
  GqPenItem* items;  // (L, cap_link): one queue per link so that a block can stage the link's mesh in LDS once
  int* count;        // (L)
  unsigned long long* keys;  // (B, P)
  long long cap_link;        // = B * P: a queue can never overflow
};

__global__ __launch_bounds__(256) void gq_pen_scan_kernel(GqPenArgs g, GqPenQ q) {
  extern __shared__ float s_link[];  // L x 24: link transform (12) + padded AABB (8) + occupancy z scale (1) + pad
  const int row = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (g.span && threadIdx.x == 0) gq_span_open(g.span, blockIdx.x + blockIdx.y * gridDim.x);
  for (int i = threadIdx.x; i < g.L * 24; i += blockDim.x) {
    const int l = i / 24, k = i % 24;
    float v = 0.0f;
    if (k < 12) v = g.link_T[((size_t)row * g.L + l) * 12 + k];
    else if (k < 20) v = g.aabb[l * 8 + (k - 12)];
    else if (k == 20) v = g.occ ? g.occ_invz[l] : 0.0f;
    s_link[i] = v;
  }
  const bool ok = pt < g.P;
  const int obj = row / g.batch_each;
  const float* sp = g.surf + ((size_t)obj * g.P + (ok ? pt : 0)) * 3;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  const gq3 xh = gq_mtv(R, gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]));
  if (ok) q.keys[(size_t)row * g.P + pt] = 0ull;
  __syncthreads();
  const int lane = gq_lane();
  for (int l = 0; l < g.L; ++l) {
    const float* T = s_link + l * 24;
    const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const gq3 xl = gq_mtv(Rl, xh - gq_mk(T[3], T[7], T[11]));
    const float* bb = T + 12;
    bool need = ok && (g.off[l + 1] > g.off[l]) && (gq_aabb_dist2(bb, xl) <= 0.0f);
    if (need && g.occ) {
      const float ux = (xl.x - bb[0]) * bb[3], uy = (xl.y - bb[1]) * bb[7], uz = (xl.z - bb[2]) * T[20];
      const int ix = min(max((int)ux, 0), 31), iy = min(max((int)uy, 0), 31), iz = min(max((int)uz, 0), 31);
      need = (g.occ[(size_t)l * 1024 + iz * 32 + iy] >> ix) & 1u;
    }
    const unsigned long long m = __ballot(need);
    if (m != 0ull) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&q.count[l], __popcll(m));
      base = gq_readlane_i(base, 0);
      if (need) {
        GqPenItem it;
        it.row = row;
        it.pt = pt;
        q.items[(size_t)l * q.cap_link + base + __popcll(m & ((1ull << lane) - 1ull))] = it;
      }
    }
  }
}

// A block = 8 wavefronts takes one chunk of 64 queue entries of one link (chunks of all links are numbered
// consecutively, so blocks spread over the queues in proportion to their length).  Lane j of EVERY wavefront owns
// entry j; wavefront w ranks the w-th eighth of the link's faces (records are wave-uniform -> scalar loads), the eight
// partial minima meet in LDS and wavefront 0 finishes the winner.  All 64 lanes work, a block lasts a few us, and
// there are enough blocks (entries / 64) to cover the chip.
#define GQ_PEN_EVAL_BLOCKS 4096
__global__ __launch_bounds__(512) void gq_pen_eval_kernel(GqPenArgs g, GqPenQ q) {
  __shared__ float s_d[8][GQ_WAVE];
  __shared__ unsigned s_o[8][GQ_WAVE];
  __shared__ int s_i[8][GQ_WAVE];
  __shared__ int s_sel[2];
  const int lane = gq_lane(), wv = threadIdx.x / GQ_WAVE;
  for (int chunk = blockIdx.x;; chunk += gridDim.x) {
    __syncthreads();  // LDS of the previous chunk fully consumed
    if (threadIdx.x == 0) {
      int c = chunk, l = 0;
      for (; l < g.L; ++l) {
        const int nc = (q.count[l] + GQ_WAVE - 1) / GQ_WAVE;
        if (c < nc) break;
        c -= nc;
      }
      s_sel[0] = (l < g.L) ? l : -1;
      s_sel[1] = c;
    }
    __syncthreads();
    const int l = s_sel[0];
    if (l < 0) return;  // block-uniform: past the last chunk
    const int count = q.count[l];
    const int i = s_sel[1] * GQ_WAVE + lane;
    const bool need = i < count;
    const int f0 = g.off[l], f1 = g.off[l + 1];
    GqPenItem it;
    it.row = 0;
    it.pt = 0;
    if (need) it = q.items[(size_t)l * q.cap_link + i];
    const int row = it.row, pt = it.pt;
    const int obj = row / g.batch_each;
    const float* sp = g.surf + ((size_t)obj * g.P + pt) * 3;
    const float* hp = g.hand_pose + (size_t)row * g.D;
    const float* R = g.Rg + (size_t)row * 9;
    const gq3 xh = gq_mtv(R, gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]));
    const float* T = g.link_T + ((size_t)row * g.L + l) * 12;
    const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const gq3 xl = gq_mtv(Rl, xh - gq_mk(T[3], T[7], T[11]));
    const int per = (f1 - f0 + 7) / 8;
    const int fa = f0 + wv * per, fb = min(fa + per, f1);
    float bd = GQ_INF_F;
    unsigned bo = 0xffffffffu;
    int bi = f0;
    if (fa < fb) {
      GqFace cur = g.rec[fa];
      for (int f = fa; f < fb; ++f) {  // f is wave-uniform: scalar loads, next record prefetched
        const GqFace nxt = g.rec[(f + 1 < fb) ? f + 1 : f];
        const gq3 d = xl - gq_mk(cur.r0.x, cur.r0.y, cur.r0.z);
        const float d2 = gq_tri_rank(cur, d);
        const unsigned orig = (unsigned)__float_as_int(cur.r5.z);
        if (d2 < bd || (d2 == bd && orig < bo)) {
          bd = d2;
          bo = orig;
          bi = f;
        }
        cur = nxt;
      }
    }
    s_d[wv][lane] = bd;
    s_o[wv][lane] = bo;
    s_i[wv][lane] = bi;
    __syncthreads();
    if (wv == 0 && need) {
#pragma unroll
      for (int w = 1; w < 8; ++w) {
        const float d2 = s_d[w][lane];
        const unsigned orig = s_o[w][lane];
        if (d2 < bd || (d2 == bd && orig < bo)) {
          bd = d2;
          bo = orig;
          bi = s_i[w][lane];
        }
      }
      const GqSdfOut o = gq_tri_finish(g.rec[bi], xl);
      if (o.sign < 0) {  // inside link l: dis = +sqrt(d^2 + 1e-8) > 0
        const float dis = sqrtf(o.dist2 + 1e-8f);
        const unsigned long long key = ((unsigned long long)__float_as_uint(dis) << 32) |
                                       ((unsigned long long)(255 - l) << 24) | (unsigned long long)(bi - f0);
        atomicMax(&q.keys[(size_t)row * g.P + pt], key);
      }
    }
  }
}

__global__ __launch_bounds__(256) void gq_pen_finalize_kernel(GqPenArgs g, GqPenQ q) {
  const int row = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < g.L) q.count[threadIdx.x] = 0;  // ready for the next query
  if (pt < g.P) {
    const size_t o = (size_t)row * g.P + pt;
    const unsigned long long key = q.keys[o];
    float dis = -1e30f;
    int link = 0;
    gq3 gh = gq_mk(0, 0, 0);
    if (key != 0ull) {
      link = 255 - (int)((key >> 24) & 0xffull);
      const int face = g.off[link] + (int)(key & 0xffffffull);
      const int obj = row / g.batch_each;
      const float* sp = g.surf + ((size_t)obj * g.P + pt) * 3;
      const float* hp = g.hand_pose + (size_t)row * g.D;
      const float* R = g.Rg + (size_t)row * 9;
      const gq3 xh = gq_mtv(R, gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]));
      const float* T = g.link_T + ((size_t)row * g.L + link) * 12;
      const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
      const gq3 xl = gq_mtv(Rl, xh - gq_mk(T[3], T[7], T[11]));
      const GqSdfOut r = gq_tri_finish(g.rec[face], xl);
      const float root = sqrtf(r.dist2 + 1e-8f);
      dis = root * (float)(-r.sign);
      gh = gq_mv(Rl, ((float)(-r.sign) / root) * (xl - r.closest));
    }
    g.dis[o] = dis;
    g.link[o] = link;
    g.gvec[o * 3 + 0] = gh.x;
    g.gvec[o * 3 + 1] = gh.y;
    g.gvec[o * 3 + 2] = gh.z;
  }
  if (g.span) {
    __syncthreads();
    if (threadIdx.x == 0) gq_span_close(g.span, blockIdx.x + blockIdx.y * gridDim.x);
  }
}

// ---- candidate lists (setup) ----------------------------------------------------------------------------------------
// Voxel with centre c and half diagonal r: for a point p of the voxel let f* be its closest face and f_c the closest
// face of c.  |p f*| <= |p f_c| <= |c f_c| + r, hence |c f*| <= |p f*| + r <= d_min(c) + 2r: every face that can win
// somewhere in the voxel lies within d_min(c) + 2r of the centre (a 1e-5 m margin absorbs rounding and the ranking
// noise of gq_tri_rank).  FILL = false counts, FILL = true writes the (ascending) face indices.
template <bool FILL>
__global__ __launch_bounds__(256) void gq_cand_kernel(const GqFace* __restrict__ rec, const int32_t* __restrict__ off,
                                                      const float* __restrict__ aabb, const float* __restrict__ invz,
                                                      const uint32_t* __restrict__ occ, uint32_t* __restrict__ count,
                                                      const uint32_t* __restrict__ cand_off,
                                                      uint16_t* __restrict__ cand_idx) {
  const int m = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;  // 0..32767 = iz*1024 + iy*32 + ix
  const int ix = v & 31, iy = (v >> 5) & 31, iz = v >> 10;
  const bool marked = (occ[(size_t)m * 1024 + iz * 32 + iy] >> ix) & 1u;
  if (!marked) {
    if (!FILL) count[(size_t)m * 32768 + v] = 0u;
    return;
  }
  const float* bb = aabb + m * 8;
  const float sx = 1.0f / bb[3], sy = 1.0f / bb[7], sz = 1.0f / invz[m];
  const gq3 p = gq_mk(bb[0] + ((float)ix + 0.5f) * sx, bb[1] + ((float)iy + 0.5f) * sy, bb[2] + ((float)iz + 0.5f) * sz);
  const float r = 0.5f * sqrtf(sx * sx + sy * sy + sz * sz);
  const int f0 = off[m], f1 = off[m + 1];
  float dmin = GQ_INF_F;
  for (int f = f0; f < f1; ++f) {
    const GqFace fc = rec[f];
    dmin = fminf(dmin, gq_tri_dist2(fc, p - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z)));
  }
  const float lim = sqrtf(fmaxf(dmin, 0.0f)) + 2.0f * r + 1e-5f;
  const float lim2 = lim * lim * 1.0001f;
  uint32_t n = 0;
  const uint32_t base = FILL ? cand_off[(size_t)m * 32768 + v] : 0u;
  for (int f = f0; f < f1; ++f) {
    const GqFace fc = rec[f];
    if (gq_tri_dist2(fc, p - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z)) <= lim2) {
      if (FILL) cand_idx[base + n] = (uint16_t)(f - f0);
      ++n;
    }
  }
  if (!FILL) count[(size_t)m * 32768 + v] = n;
}

// ---- occupancy grid construction (setup) ---------------------------------------------------------------------------
__global__ void gq_occ_faces_kernel(const GqFace* __restrict__ rec, const int32_t* __restrict__ off, int n_mesh,
                                    const float* __restrict__ aabb, const float* __restrict__ invz,
                                    uint32_t* __restrict__ occ) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= off[n_mesh]) return;
  int m = 0;
  while (m + 1 < n_mesh && f >= off[m + 1]) ++m;
  const GqFace fc = rec[f];
  gq3 a, b, c;
  gq_face_corners(fc, a, b, c);
  const float* bb = aabb + m * 8;
  const float sc[3] = {bb[3], bb[7], invz[m]};
  const float lo[3] = {fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z)};
  const float hi[3] = {fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z)};
  int i0[3], i1[3];
  for (int k = 0; k < 3; ++k) {  // one extra voxel of margin on both sides
    i0[k] = min(max((int)floorf((lo[k] - bb[k]) * sc[k]) - 1, 0), 31);
    i1[k] = min(max((int)floorf((hi[k] - bb[k]) * sc[k]) + 1, 0), 31);
  }
  const uint32_t xm = (i1[0] >= 31 ? 0xffffffffu : ((1u << (i1[0] + 1)) - 1u)) & ~((1u << i0[0]) - 1u);
  for (int iz = i0[2]; iz <= i1[2]; ++iz)
    for (int iy = i0[1]; iy <= i1[1]; ++iy) atomicOr(&occ[(size_t)m * 1024 + iz * 32 + iy], xm);
}

// one thread per voxel of mesh blockIdx.y: inside?  (32768 voxels per mesh).  "Inside" is TorchSDF's rule -- the sign of
// the closest face -- which is not constant over a voxel that touches no face: next to edges whose first-minimum face
// looks away, and inside links whose mesh is several shells (shadow hand), it flips over a fraction of a millimetre.  The
// voxel is therefore probed at its centre AND its eight corners (brute force over the faces, setup time only) and marked
// when any probe is inside: a marked voxel only means "evaluate the point itself", so probing more can never make a
// result wrong, it only leaves fewer points to the "unmarked = outside" shortcut (deviation iv).
__global__ __launch_bounds__(256) void gq_occ_centres_kernel(const GqFace* __restrict__ rec,
                                                             const int32_t* __restrict__ off,
                                                             const float* __restrict__ aabb,
                                                             const float* __restrict__ invz, uint32_t* __restrict__ occ) {
  const int m = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;  // 0..32767
  const int ix = v & 31, iy = (v >> 5) & 31, iz = v >> 10;
  const float* bb = aabb + m * 8;
  const int f0 = off[m], f1 = off[m + 1];
  bool inside = false;
  for (int s = 0; s < 9 && !inside; ++s) {  // s = 0: centre; 1..8: corners
    const float ox = s == 0 ? 0.5f : (float)((s - 1) & 1), oy = s == 0 ? 0.5f : (float)(((s - 1) >> 1) & 1),
                oz = s == 0 ? 0.5f : (float)(((s - 1) >> 2) & 1);
    const gq3 p = gq_mk(bb[0] + ((float)ix + ox) / bb[3], bb[1] + ((float)iy + oy) / bb[7], bb[2] + ((float)iz + oz) / invz[m]);
    float best = GQ_INF_F;
    int bi = -1;
    for (int f = f0; f < f1; ++f) {
      const GqFace fc = rec[f];
      const float d2 = gq_tri_rank(fc, p - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z));
      if (d2 < best) {
        best = d2;
        bi = f;
      }
    }
    if (bi >= 0) inside = gq_tri_finish(rec[bi], p).sign < 0;
  }
  if (inside) atomicOr(&occ[(size_t)m * 1024 + iz * 32 + iy], 1u << ix);
}

#include <algorithm>
#include <cmath>
#include <vector>

static inline uint32_t gq_spread10(uint32_t v) {
  v &= 0x3ff;
  v = (v | (v << 16)) & 0x030000ff;
  v = (v | (v << 8)) & 0x0300f00f;
  v = (v | (v << 4)) & 0x030c30c3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}

static void gq_box_of(const float* fv, const int32_t* perm, int64_t a, int64_t b, float* out8) {
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  for (int64_t i = a; i < b; ++i) {
    const float* v = fv + (int64_t)perm[i] * 9;
    for (int k = 0; k < 9; ++k) {
      const int c = k % 3;
      lo[c] = v[k] < lo[c] ? v[k] : lo[c];
      hi[c] = v[k] > hi[c] ? v[k] : hi[c];
    }
  }
  out8[0] = lo[0]; out8[1] = lo[1]; out8[2] = lo[2]; out8[3] = 0.0f;
  out8[4] = hi[0]; out8[5] = hi[1]; out8[6] = hi[2]; out8[7] = 0.0f;
}

static unsigned long long* gq_pen_dbg_ = nullptr;
// Surface points per thread of the penetration query (gq_debug_set_pen_ppt: 0 = defaults, 1 / 2 forced).  Two points per
// thread halve the wavefronts and share prologue, loop overhead and list phases: +1.5..2 % at 256 rows, where the query is
// a role of stage A.  As a launch of its own beside the force-closure branch the same variant is much faster alone (167 ->
// 124 us at 2048 rows) but holds 84 instead of 62 VGPRs per wavefront, and the iteration gets 18 % SLOWER -- the
// stand-alone kernel keeps one point per thread.
static int gq_pen_ppt_ = 0;
int gq_pen_points_per_thread_() { return gq_pen_ppt_ == 1 ? 1 : 2; }          // fused (stage A role)
static int gq_pen_points_per_thread_standalone_() { return gq_pen_ppt_ == 2 ? 2 : 1; }
static int gq_pen_caps_ = 0;  // gq_debug_set_pen_caps: 0 = by launch size, 1 / 2 / 3 = 512 / 256 / 128 entries (A/B runs)
static int gq_sdf_plain_mapping_ = 0;  // gq_debug_set_sdf_mapping(1): A/B switch for the XCD-aware query placement

// Bound of a 64-face cluster: an oriented box, 16 floats = [centre.xyz, h_u][u.xyz, h_v][v.xyz, h_n][n.xyz, 0].
// n = area-weighted mean normal of the patch, u = principal direction of its vertices in the plane orthogonal to n,
// v = n x u.  A Morton patch of a surface mesh is nearly planar, so the box is ~1 mm thick along n and hugs the patch
// laterally -- for a query point several centimetres away the neighbouring patches are only millimetres farther than
// the nearest one, and an axis-aligned box around a tilted patch is too loose to tell them apart.
static void gq_cluster_bound(const float* fv, const int32_t* perm, int64_t a, int64_t b, float* out16) {
  double n[3] = {0, 0, 0}, c0[3] = {0, 0, 0};
  for (int64_t i = a; i < b; ++i) {
    const float* v = fv + (int64_t)perm[i] * 9;
    const double e1[3] = {(double)v[3] - v[0], (double)v[4] - v[1], (double)v[5] - v[2]};
    const double e2[3] = {(double)v[6] - v[0], (double)v[7] - v[1], (double)v[8] - v[2]};
    n[0] += e1[1] * e2[2] - e1[2] * e2[1];
    n[1] += e1[2] * e2[0] - e1[0] * e2[2];
    n[2] += e1[0] * e2[1] - e1[1] * e2[0];
    for (int k = 0; k < 9; ++k) c0[k % 3] += v[k];
  }
  const double cnt = 3.0 * (double)(b - a);
  for (int k = 0; k < 3; ++k) c0[k] /= cnt;
  double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  if (len > 1e-30) {
    for (int k = 0; k < 3; ++k) n[k] /= len;
  } else {
    n[0] = 0; n[1] = 0; n[2] = 1;
  }
  // t1 orthogonal to n (drop the smallest component), t2 = n x t1
  double t1[3], t2[3];
  {
    const int m = (std::fabs(n[0]) <= std::fabs(n[1]) && std::fabs(n[0]) <= std::fabs(n[2])) ? 0
                  : (std::fabs(n[1]) <= std::fabs(n[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d = n[m];
    for (int k = 0; k < 3; ++k) t1[k] = e[k] - d * n[k];
    const double l = std::sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
    for (int k = 0; k < 3; ++k) t1[k] /= l;
    t2[0] = n[1] * t1[2] - n[2] * t1[1];
    t2[1] = n[2] * t1[0] - n[0] * t1[2];
    t2[2] = n[0] * t1[1] - n[1] * t1[0];
  }
  double cxx = 0, cxy = 0, cyy = 0;
  for (int64_t i = a; i < b; ++i) {
    const float* v = fv + (int64_t)perm[i] * 9;
    for (int c = 0; c < 3; ++c) {
      const double q[3] = {v[c * 3] - c0[0], v[c * 3 + 1] - c0[1], v[c * 3 + 2] - c0[2]};
      const double x = q[0] * t1[0] + q[1] * t1[1] + q[2] * t1[2], y = q[0] * t2[0] + q[1] * t2[1] + q[2] * t2[2];
      cxx += x * x;
      cxy += x * y;
      cyy += y * y;
    }
  }
  const double th = 0.5 * std::atan2(2.0 * cxy, cxx - cyy);
  float ax[3][3];  // u, v, n rounded to fp32 (the extents below are taken along the ROUNDED axes)
  for (int k = 0; k < 3; ++k) {
    ax[0][k] = (float)(std::cos(th) * t1[k] + std::sin(th) * t2[k]);
    ax[1][k] = (float)(-std::sin(th) * t1[k] + std::cos(th) * t2[k]);
    ax[2][k] = (float)n[k];
  }
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, mag = 0.0;
  for (int64_t i = a; i < b; ++i) {
    const float* v = fv + (int64_t)perm[i] * 9;
    for (int c = 0; c < 3; ++c) {
      const double q[3] = {v[c * 3] - c0[0], v[c * 3 + 1] - c0[1], v[c * 3 + 2] - c0[2]};
      for (int k = 0; k < 3; ++k) {
        const double s = q[0] * ax[k][0] + q[1] * ax[k][1] + q[2] * ax[k][2];
        lo[k] = s < lo[k] ? s : lo[k];
        hi[k] = s > hi[k] ? s : hi[k];
      }
      const double m = std::fabs((double)v[c * 3]) + std::fabs((double)v[c * 3 + 1]) + std::fabs((double)v[c * 3 + 2]);
      mag = m > mag ? m : mag;
    }
  }
  double ctr[3] = {c0[0], c0[1], c0[2]};
  for (int k = 0; k < 3; ++k)
    for (int j = 0; j < 3; ++j) ctr[j] += 0.5 * (lo[k] + hi[k]) * ax[k][j];
  const double pad = 1e-6 * mag + 1e-12;  // fp32 evaluation on the device + rounding of the centre
  for (int k = 0; k < 3; ++k) {
    out16[k] = (float)ctr[k];
    out16[4 + k] = ax[0][k];
    out16[8 + k] = ax[1][k];
    out16[12 + k] = ax[2][k];
  }
  out16[3] = (float)(0.5 * (hi[0] - lo[0]) + pad);
  out16[7] = (float)(0.5 * (hi[1] - lo[1]) + pad);
  out16[11] = (float)(0.5 * (hi[2] - lo[2]) + pad);
  out16[15] = 0.0f;
}

// internal (kin.hip): argument block of the wave-per-query kernel for a mesh set; `points` is left to the caller
int gq_sdf_wave_args_(const gqMeshSet* ms, int64_t n_points, int64_t queries_per_mesh, float* dist_sq, int32_t* sign,
                      float* normal, float* closest, GqWaveArgs* out) {
  GQ_REQUIRE(ms && dist_sq && sign && closest && out, "sdf_forward_meshset: null pointer");
  GQ_REQUIRE(queries_per_mesh > 0 && n_points == queries_per_mesh * ms->n_mesh,
             "sdf_forward_meshset: n_points=%lld != queries_per_mesh=%lld * n_mesh=%d", (long long)n_points,
             (long long)queries_per_mesh, ms->n_mesh);
  GqWaveArgs w{};
  w.N = n_points;
  w.rec = ms->rec;
  w.off = ms->off_dev;
  w.cl_aabb = ms->cl_aabb_dev;
  w.cl_off = ms->cl_off_dev;
  w.queries_per_mesh = queries_per_mesh;
  w.dbg = gq_pen_dbg_;
  w.dist_sq = dist_sq;
  w.sign = sign;
  w.normal = normal;
  w.closest = closest;
  *out = w;
  return GQ_OK;
}

extern "C" {

// diagnostics: device pointer to 4 uint64 counters filled by gq_hand_pen_forward (NULL = off, the default)
int gq_debug_set_sdf_topk(int topk) {
  gq_sdf_topk_ = (topk == 2 || topk == 4) ? topk : 0;
  return GQ_OK;
}
int gq_debug_set_pen_ppt(int ppt) {
  gq_pen_ppt_ = (ppt == 1 || ppt == 2) ? ppt : 0;
  return GQ_OK;
}
int gq_debug_set_pen_caps(int mode) {
  gq_pen_caps_ = mode;
  return GQ_OK;
}
int gq_debug_set_sdf_mapping(int plain) {
  gq_sdf_plain_mapping_ = plain;
  return GQ_OK;
}

int gq_debug_set_pen_counters(uint64_t* counters) {
  gq_pen_dbg_ = (unsigned long long*)counters;
  return GQ_OK;
}

int gq_meshset_create(const float* face_verts_host, const int32_t* face_offset_host, int n_mesh, gqMeshSet** out) {
  GQ_REQUIRE(face_verts_host && face_offset_host && out && n_mesh > 0, "meshset_create: bad arguments");
  const int64_t F = face_offset_host[n_mesh];
  GQ_REQUIRE(F > 0 && face_offset_host[0] == 0 && F < (1ll << 31), "meshset_create: empty or oversized face list");
  for (int i = 0; i < n_mesh; ++i)
    GQ_REQUIRE(face_offset_host[i + 1] >= face_offset_host[i], "meshset_create: offsets must be non-decreasing");
  gqMeshSet* ms = new gqMeshSet();
  ms->n_mesh = n_mesh;
  ms->n_faces = F;
  ms->off_host = new int32_t[n_mesh + 1];
  memcpy(ms->off_host, face_offset_host, sizeof(int32_t) * (n_mesh + 1));
  // Morton order of the face centroids inside each mesh, then cluster boxes
  std::vector<int32_t> perm(F);
  std::vector<float> mesh_bb((size_t)n_mesh * 8), sub_bb, cl_bb;
  std::vector<int32_t> sub_off(n_mesh + 1, 0), cl_off(n_mesh + 1, 0);
  for (int m = 0; m < n_mesh; ++m) {
    const int64_t a = face_offset_host[m], b = face_offset_host[m + 1];
    for (int64_t i = a; i < b; ++i) perm[i] = (int32_t)i;
    gq_box_of(face_verts_host, perm.data(), a, b, &mesh_bb[(size_t)m * 8]);
    const float* bb = &mesh_bb[(size_t)m * 8];
    std::vector<std::pair<uint32_t, int32_t>> keys;
    keys.reserve(b - a);
    for (int64_t i = a; i < b; ++i) {
      const float* v = face_verts_host + i * 9;
      uint32_t code = 0;
      for (int c = 0; c < 3; ++c) {
        const float ctr = (v[c] + v[3 + c] + v[6 + c]) * (1.0f / 3.0f);
        const float ext = bb[4 + c] - bb[c];
        float t = ext > 0.0f ? (ctr - bb[c]) / ext : 0.0f;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        code |= gq_spread10((uint32_t)(t * 1023.0f)) << c;
      }
      keys.emplace_back(code, (int32_t)i);
    }
    std::stable_sort(keys.begin(), keys.end());
    for (int64_t i = a; i < b; ++i) perm[i] = keys[i - a].second;
    for (int64_t i = a; i < b; i += 16) {
      sub_bb.resize(sub_bb.size() + 8);
      gq_box_of(face_verts_host, perm.data(), i, std::min<int64_t>(i + 16, b), &sub_bb[sub_bb.size() - 8]);
    }
    for (int64_t i = a; i < b; i += 64) {
      cl_bb.resize(cl_bb.size() + 16);
      gq_cluster_bound(face_verts_host, perm.data(), i, std::min<int64_t>(i + 64, b), &cl_bb[cl_bb.size() - 16]);
    }
    sub_off[m + 1] = (int32_t)(sub_bb.size() / 8);
    cl_off[m + 1] = (int32_t)(cl_bb.size() / 16);
  }
  float* tmp = nullptr;
  int32_t* perm_dev = nullptr;
  GQ_CHECK_HIP(hipMalloc(&tmp, (size_t)F * 9 * 4));
  GQ_CHECK_HIP(hipMalloc(&perm_dev, (size_t)F * 4));
  GQ_CHECK_HIP(hipMalloc(&ms->rec, (size_t)F * sizeof(GqFace)));
  GQ_CHECK_HIP(hipMalloc(&ms->off_dev, sizeof(int32_t) * (n_mesh + 1)));
  GQ_CHECK_HIP(hipMalloc(&ms->aabb_dev, sizeof(float) * n_mesh * 8));
  GQ_CHECK_HIP(hipMalloc(&ms->sub_aabb_dev, sizeof(float) * sub_bb.size()));
  GQ_CHECK_HIP(hipMalloc(&ms->cl_aabb_dev, sizeof(float) * cl_bb.size()));
  GQ_CHECK_HIP(hipMalloc(&ms->sub_off_dev, sizeof(int32_t) * (n_mesh + 1)));
  GQ_CHECK_HIP(hipMalloc(&ms->cl_off_dev, sizeof(int32_t) * (n_mesh + 1)));
  GQ_CHECK_HIP(hipMemcpy(tmp, face_verts_host, (size_t)F * 9 * 4, hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(perm_dev, perm.data(), (size_t)F * 4, hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(ms->off_dev, face_offset_host, sizeof(int32_t) * (n_mesh + 1), hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(ms->aabb_dev, mesh_bb.data(), sizeof(float) * n_mesh * 8, hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(ms->sub_aabb_dev, sub_bb.data(), sizeof(float) * sub_bb.size(), hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(ms->cl_aabb_dev, cl_bb.data(), sizeof(float) * cl_bb.size(), hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(ms->sub_off_dev, sub_off.data(), sizeof(int32_t) * (n_mesh + 1), hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(ms->cl_off_dev, cl_off.data(), sizeof(int32_t) * (n_mesh + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gq_face_prep_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, 0, tmp, perm_dev, ms->rec, F);
  GQ_LAUNCH_CHECK();
  GQ_CHECK_HIP(hipDeviceSynchronize());
  GQ_CHECK_HIP(hipFree(tmp));
  GQ_CHECK_HIP(hipFree(perm_dev));
  *out = ms;
  return GQ_OK;
}

// Occupancy grid of every mesh (setup-time).  Voxel (ix,iy,iz) of the 32^3 grid over the mesh AABB is marked when
// (a) the bounding box of some face overlaps it (+- one voxel), or (b) its centre or one of its corners is inside the mesh
// (sign of the closest face).  An unmarked voxel does not touch the surface and nine probes of it are outside.
int gq_meshset_build_occupancy(gqMeshSet* ms) {
  GQ_REQUIRE(ms, "meshset_build_occupancy: null");
  if (ms->occ_dev) return GQ_OK;
  const int n = ms->n_mesh;
  std::vector<float> bb((size_t)n * 8), invz(n);
  GQ_CHECK_HIP(hipMemcpy(bb.data(), ms->aabb_dev, sizeof(float) * n * 8, hipMemcpyDeviceToHost));
  for (int m = 0; m < n; ++m) {
    // grow the box a little so that surface points are strictly inside the grid, then store 32/extent
    for (int c = 0; c < 3; ++c) {
      const float ext = bb[m * 8 + 4 + c] - bb[m * 8 + c];
      const float pad = 1e-4f * ext + 1e-7f;
      bb[m * 8 + c] -= pad;
      bb[m * 8 + 4 + c] += pad;
    }
    bb[m * 8 + 3] = 32.0f / (bb[m * 8 + 4] - bb[m * 8 + 0]);
    bb[m * 8 + 7] = 32.0f / (bb[m * 8 + 5] - bb[m * 8 + 1]);
    invz[m] = 32.0f / (bb[m * 8 + 6] - bb[m * 8 + 2]);
  }
  GQ_CHECK_HIP(hipMemcpy(ms->aabb_dev, bb.data(), sizeof(float) * n * 8, hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMalloc(&ms->occ_invz_dev, sizeof(float) * n));
  GQ_CHECK_HIP(hipMemcpy(ms->occ_invz_dev, invz.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMalloc(&ms->occ_dev, sizeof(uint32_t) * n * 1024));
  GQ_CHECK_HIP(hipMemset(ms->occ_dev, 0, sizeof(uint32_t) * n * 1024));
  hipLaunchKernelGGL(gq_occ_faces_kernel, dim3((unsigned)((ms->n_faces + 255) / 256)), dim3(256), 0, 0, ms->rec,
                     ms->off_dev, n, ms->aabb_dev, ms->occ_invz_dev, ms->occ_dev);
  GQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(gq_occ_centres_kernel, dim3(128, (unsigned)n), dim3(256), 0, 0, ms->rec, ms->off_dev, ms->aabb_dev,
                     ms->occ_invz_dev, ms->occ_dev);
  GQ_LAUNCH_CHECK();
  GQ_CHECK_HIP(hipDeviceSynchronize());
  // per-voxel candidate faces (only when 16-bit local face indices suffice)
  bool small = true;
  for (int m = 0; m < n; ++m) small = small && (ms->off_host[m + 1] - ms->off_host[m] < 65536);
  if (small) {
    const size_t nv = (size_t)n * 32768;
    uint32_t* cnt_dev = nullptr;
    GQ_CHECK_HIP(hipMalloc(&cnt_dev, sizeof(uint32_t) * nv));
    hipLaunchKernelGGL(gq_cand_kernel<false>, dim3(128, (unsigned)n), dim3(256), 0, 0, ms->rec, ms->off_dev, ms->aabb_dev,
                       ms->occ_invz_dev, ms->occ_dev, cnt_dev, nullptr, nullptr);
    GQ_LAUNCH_CHECK();
    std::vector<uint32_t> cnt(nv), offs(nv + 1);
    GQ_CHECK_HIP(hipMemcpy(cnt.data(), cnt_dev, sizeof(uint32_t) * nv, hipMemcpyDeviceToHost));
    GQ_CHECK_HIP(hipFree(cnt_dev));
    uint64_t run = 0;
    for (size_t i = 0; i < nv; ++i) {
      offs[i] = (uint32_t)run;
      run += cnt[i];
    }
    GQ_REQUIRE(run < (1ull << 32), "meshset_build_occupancy: candidate lists too long");
    offs[nv] = (uint32_t)run;
    ms->n_cand = (int64_t)run;
    GQ_CHECK_HIP(hipMalloc(&ms->cand_off_dev, sizeof(uint32_t) * (nv + 1)));
    GQ_CHECK_HIP(hipMalloc(&ms->cand_idx_dev, sizeof(uint16_t) * (run + 1)));
    GQ_CHECK_HIP(hipMemcpy(ms->cand_off_dev, offs.data(), sizeof(uint32_t) * (nv + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gq_cand_kernel<true>, dim3(128, (unsigned)n), dim3(256), 0, 0, ms->rec, ms->off_dev, ms->aabb_dev,
                       ms->occ_invz_dev, ms->occ_dev, nullptr, ms->cand_off_dev, ms->cand_idx_dev);
    GQ_LAUNCH_CHECK();
    GQ_CHECK_HIP(hipDeviceSynchronize());
  }
  return GQ_OK;
}

int gq_meshset_destroy(gqMeshSet* ms) {
  if (!ms) return GQ_OK;
  (void)hipFree(ms->rec);
  (void)hipFree(ms->off_dev);
  (void)hipFree(ms->aabb_dev);
  (void)hipFree(ms->sub_aabb_dev);
  (void)hipFree(ms->cl_aabb_dev);
  (void)hipFree(ms->sub_off_dev);
  (void)hipFree(ms->cl_off_dev);
  if (ms->occ_dev) (void)hipFree(ms->occ_dev);
  if (ms->occ_invz_dev) (void)hipFree(ms->occ_invz_dev);
  if (ms->cand_off_dev) (void)hipFree(ms->cand_off_dev);
  if (ms->cand_idx_dev) (void)hipFree(ms->cand_idx_dev);
  delete[] ms->off_host;
  delete ms;
  return GQ_OK;
}

int gq_meshset_num_faces(const gqMeshSet* ms, int mesh, int64_t* n) {
  GQ_REQUIRE(ms && n && mesh >= -1 && mesh < ms->n_mesh, "meshset_num_faces: bad arguments");
  *n = mesh < 0 ? ms->n_faces : (ms->off_host[mesh + 1] - ms->off_host[mesh]);
  return GQ_OK;
}

int gq_sdf_workspace_bytes(int64_t n_faces, size_t* bytes) {
  GQ_REQUIRE(bytes && n_faces >= 0, "sdf_workspace_bytes: bad arguments");
  *bytes = (size_t)n_faces * sizeof(GqFace) + 256;
  return GQ_OK;
}

// TorchSDF-contract query against a raw (F,3,3) device triangle soup.
int gq_sdf_forward(const float* points, int64_t n_points, const float* face_verts, int64_t n_faces, float* dist_sq,
                   int32_t* sign, float* normal, float* closest, void* workspace, size_t workspace_bytes,
                   void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_points == 0) return GQ_OK;
  GQ_REQUIRE(points && face_verts && dist_sq && sign && closest && workspace, "sdf_forward: null pointer");
  GQ_REQUIRE(n_points > 0 && n_faces > 0 && n_faces < (1ll << 31), "sdf_forward: bad sizes N=%lld F=%lld",
             (long long)n_points, (long long)n_faces);
  GQ_REQUIRE(workspace_bytes >= (size_t)n_faces * sizeof(GqFace) + 256, "sdf_forward: workspace too small");
  GqFace* rec = (GqFace*)workspace;
  hipLaunchKernelGGL(gq_face_prep_kernel, dim3((unsigned)((n_faces + 255) / 256)), dim3(256), 0, st, face_verts,
                     (const int32_t*)nullptr, rec, n_faces);
  GQ_LAUNCH_CHECK();
  if (n_points >= 131072) {
    hipLaunchKernelGGL(gq_sdf_points_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, points,
                       n_points, rec, (int)n_faces, dist_sq, sign, normal, closest);
  } else {
    GqWaveArgs w{};
    w.points = points;
    w.N = n_points;
    w.rec = rec;
    w.single_F = (int)n_faces;
    w.queries_per_mesh = n_points;
    w.dist_sq = dist_sq;
    w.sign = sign;
    w.normal = normal;
    w.closest = closest;
    gq_sdf_wave_launch(w, (unsigned)((n_points + 3) / 4), n_points, st);
  }
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Query against a mesh set: points (n_mesh * queries_per_mesh, 3); query q uses mesh q / queries_per_mesh
// (object_model.py:217-220: one mesh per object, batch_size_each * n_contact queries each).
int gq_sdf_forward_meshset(const gqMeshSet* ms, const float* points, int64_t n_points, int64_t queries_per_mesh,
                           float* dist_sq, int32_t* sign, float* normal, float* closest, void* stream) {
  if (n_points == 0) return GQ_OK;
  GQ_REQUIRE(points, "sdf_forward_meshset: null pointer");
  GqWaveArgs w{};
  int rc = gq_sdf_wave_args_(ms, n_points, queries_per_mesh, dist_sq, sign, normal, closest, &w);
  if (rc) return rc;
  w.points = points;
  unsigned blocks = (unsigned)((n_points + 3) / 4);
  if (ms->n_mesh >= 8 && !gq_sdf_plain_mapping_) {  // several meshes: keep each mesh's queries on one XCD (its L2)
    w.n_mesh = ms->n_mesh;
    w.xcd_meshes = (ms->n_mesh + 7) / 8;
    w.blocks_per_mesh = (int)((queries_per_mesh + 3) / 4);
    blocks = (unsigned)(8 * w.xcd_meshes * w.blocks_per_mesh);
  }
  gq_sdf_wave_launch(w, blocks, n_points, (hipStream_t)stream);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_sdf_backward(const float* grad_dist_sq, const float* points, const float* closest, int64_t n_points,
                    float* grad_points, void* stream) {
  if (n_points == 0) return GQ_OK;
  GQ_REQUIRE(grad_dist_sq && points && closest && grad_points && n_points > 0, "sdf_backward: bad arguments");
  hipLaunchKernelGGL(gq_sdf_bwd_kernel, dim3((unsigned)((n_points * 3 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, grad_dist_sq, points, closest, n_points, grad_points);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Fused hand-penetration query (HandModel.cal_distance, hand_model.py:875-987).
int gq_hand_pen_forward(const gqMeshSet* links, const float* surface_points, int64_t n_obj, int64_t n_surface,
                        int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg, const float* link_T,
                        int penetration_only, float* dis, int32_t* link, float* gvec, void* workspace,
                        size_t workspace_bytes, void* timer, uint64_t* span, const float* patch_spheres, void* stream) {
  GqPenArgs a{};
  int rc0 = gq_pen_fill(links, surface_points, n_obj, n_surface, batch_each, hand_pose, pose_dim, Rg, link_T, dis, link,
                        gvec, span, &a);
  if (rc0) return rc0;
  a.dbg = gq_pen_dbg_;
  a.patch = patch_spheres;
  const dim3 grid((unsigned)((a.P + 255) / 256), (unsigned)a.B);
  const dim3 grid_sm((unsigned)a.B, (unsigned)((a.P + 255) / 256));  // gq_pen_grid_kernel: x = row, y = slice block
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timer) {  // gqTimer: the kernel's own start/stop timestamps (hipExtLaunchKernelGGL), not stream markers
    e0 = ((hipEvent_t*)timer)[0];
    e1 = ((hipEvent_t*)timer)[1];
  }
  if (penetration_only == 1 && a.occ && a.cand_off) {
    // one pass over the (point, link) pairs, candidate faces from the voxel grid (see gq_pen_grid_kernel)
    // LDS list capacities: smaller lists double the blocks per CU and make THIS kernel faster (168 -> 115 us at 2048
    // rows) but not the iteration -- the branch it runs beside loses the slots it gains (tools/ab_caps.sh) -- and the
    // smallest ones overflow into inline ranking; so the large lists stay the default
    const int caps = gq_pen_caps_ ? gq_pen_caps_ : 1;
    if (caps == 3)
      hipExtLaunchKernelGGL((gq_pen_grid_kernel<true, 128, 1024>), grid_sm, dim3(256), gq_pen_grid_lds_bytes(a.L, 128, 1024),
                            (hipStream_t)stream, e0, e1, 0, a);
    else if (caps == 2)
      hipExtLaunchKernelGGL((gq_pen_grid_kernel<true, 256, 2048>), grid_sm, dim3(256), gq_pen_grid_lds_bytes(a.L, 256, 2048),
                            (hipStream_t)stream, e0, e1, 0, a);
    else if (gq_pen_points_per_thread_standalone_() == 2)
      hipExtLaunchKernelGGL((gq_pen_grid_kernel<true, GQ_PG_ECAP, GQ_PG_ICAP, 2>), dim3((unsigned)a.B, (unsigned)((a.P + 511) / 512)),
                            dim3(256), gq_pen_grid_lds_bytes(a.L, GQ_PG_ECAP, GQ_PG_ICAP, 2), (hipStream_t)stream, e0, e1, 0, a);
    else
      hipExtLaunchKernelGGL((gq_pen_grid_kernel<true, GQ_PG_ECAP, GQ_PG_ICAP>), grid_sm, dim3(256), gq_pen_grid_lds_bytes(a.L),
                            (hipStream_t)stream, e0, e1, 0, a);
  } else if (penetration_only == 9 && a.occ && a.cand_off) {  // diagnostics: the scan without candidate evaluation
    hipExtLaunchKernelGGL((gq_pen_grid_kernel<false, GQ_PG_ECAP, GQ_PG_ICAP>), grid_sm, dim3(256), gq_pen_grid_lds_bytes(a.L),
                          (hipStream_t)stream, e0, e1, 0, a);
  } else if ((penetration_only == 1 || penetration_only == 3) && workspace != nullptr) {
    // queue-based, load-balanced path without candidate lists (see gq_pen_scan_kernel); 3 forces it for A/B tests
    const size_t cap_link = (size_t)a.B * a.P;
    const size_t cap = cap_link * a.L;
    const size_t need = 1024 + cap * sizeof(GqPenItem) + (size_t)a.B * a.P * 8;
    GQ_REQUIRE(workspace_bytes >= need, "hand_pen_forward: workspace too small (%zu < %zu)", workspace_bytes, need);
    GQ_REQUIRE(a.L <= 255 && cap_link < (1ull << 31), "hand_pen_forward: too many links / items for the queue path");
    GqPenQ q;
    q.count = (int*)workspace;
    q.items = (GqPenItem*)((char*)workspace + 1024);
    q.keys = (unsigned long long*)((char*)workspace + 1024 + cap * sizeof(GqPenItem));
    q.cap_link = (long long)cap_link;
    // q.count is zero on entry: the workspace starts zeroed and gq_pen_finalize_kernel re-zeroes it after use
    hipExtLaunchKernelGGL(gq_pen_scan_kernel, grid, dim3(256), (size_t)a.L * 24 * sizeof(float), (hipStream_t)stream, e0,
                          nullptr, 0, a, q);
    GQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(gq_pen_eval_kernel, dim3(GQ_PEN_EVAL_BLOCKS), dim3(512), 0, (hipStream_t)stream, a, q);
    GQ_LAUNCH_CHECK();
    hipExtLaunchKernelGGL(gq_pen_finalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, nullptr, e1, 0, a, q);
  } else if (penetration_only == 1 || penetration_only == 3)
    hipExtLaunchKernelGGL(gq_hand_pen_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  else if (penetration_only == 2)  // AABB test only, no occupancy grid (kept for A/B tests)
    hipExtLaunchKernelGGL(gq_hand_pen_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  else
    hipExtLaunchKernelGGL(gq_hand_pen_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Bounding sphere of every 256-point slice of every object's surface points (set-up time; one block per slice).
__global__ __launch_bounds__(256) void gq_patch_kernel(const float* __restrict__ surf, int P, float* __restrict__ out) {
  __shared__ float s_red[4][6];
  __shared__ float s_c[3];
  const int obj = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x, lane = gq_lane(), wv = tid / GQ_WAVE;
  const int pt = bx * 256 + tid;
  const bool ok = pt < P;
  const float* sp = surf + ((size_t)obj * P + (ok ? pt : bx * 256)) * 3;  // out-of-range lanes repeat the slice's first point
  const gq3 x = gq_mk(sp[0], sp[1], sp[2]);
  float v[6] = {x.x, x.y, x.z, -x.x, -x.y, -x.z};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float m = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, GQ_WAVE));
    if (lane == 0) s_red[wv][k] = m;
  }
  __syncthreads();
  if (tid < 3) {
    const float lo = fminf(fminf(s_red[0][tid], s_red[1][tid]), fminf(s_red[2][tid], s_red[3][tid]));
    const float nh = fminf(fminf(s_red[0][3 + tid], s_red[1][3 + tid]), fminf(s_red[2][3 + tid], s_red[3][3 + tid]));
    s_c[tid] = 0.5f * (lo - nh);
  }
  __syncthreads();
  const gq3 d = x - gq_mk(s_c[0], s_c[1], s_c[2]);
  float r = sqrtf(gq_dot(d, d));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) r = fmaxf(r, __shfl_xor(r, o, GQ_WAVE));
  __syncthreads();
  if (lane == 0) s_red[wv][0] = r;
  __syncthreads();
  if (tid == 0) {
    float* o = out + ((size_t)obj * gridDim.x + bx) * 4;
    o[0] = s_c[0]; o[1] = s_c[1]; o[2] = s_c[2];
    o[3] = fmaxf(fmaxf(s_red[0][0], s_red[1][0]), fmaxf(s_red[2][0], s_red[3][0])) * 1.0001f + 1e-7f;
  }
}

int gq_surface_patches(const float* surface_points, int64_t n_obj, int64_t n_surface, float* patch_spheres, void* stream) {
  GQ_REQUIRE(surface_points && patch_spheres && n_obj > 0 && n_surface > 0, "surface_patches: bad arguments");
  hipLaunchKernelGGL(gq_patch_kernel, dim3((unsigned)((n_surface + 255) / 256), (unsigned)n_obj), dim3(256), 0,
                     (hipStream_t)stream, surface_points, (int)n_surface, patch_spheres);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Uniform grid over the surface points of every object (host build, set-up time).
int gq_pointgrid_create(const float* surface_points_host, int64_t n_obj, int64_t n_surface, int cells_per_axis,
                        gqPointGrid** out) {
  GQ_REQUIRE(surface_points_host && out && n_obj > 0 && n_surface > 0 && n_surface <= 65535, "pointgrid_create: bad arguments");
  const int G = cells_per_axis > 0 ? cells_per_axis : 8;
  GQ_REQUIRE(G >= 1 && G <= 32, "pointgrid_create: cells_per_axis must be in 1..32");
  const int NC = G * G * G;
  const int P = (int)n_surface;
  std::vector<float> box((size_t)n_obj * 8, 0.0f);
  std::vector<int32_t> start((size_t)n_obj * (NC + 1), 0);
  std::vector<uint16_t> pts((size_t)n_obj * P, 0);
  std::vector<int> cell(P);
  for (int64_t o = 0; o < n_obj; ++o) {
    const float* sp = surface_points_host + (size_t)o * P * 3;
    float lo[3] = {sp[0], sp[1], sp[2]}, hi[3] = {sp[0], sp[1], sp[2]};
    for (int i = 0; i < P; ++i)
      for (int k = 0; k < 3; ++k) {
        lo[k] = sp[i * 3 + k] < lo[k] ? sp[i * 3 + k] : lo[k];
        hi[k] = sp[i * 3 + k] > hi[k] ? sp[i * 3 + k] : hi[k];
      }
    float* b = box.data() + (size_t)o * 8;
    for (int k = 0; k < 3; ++k) {
      b[k] = lo[k];
      const float ext = hi[k] - lo[k];
      b[4 + k] = (float)G / (ext > 1e-9f ? ext : 1e-9f);
    }
    int32_t* st = start.data() + (size_t)o * (NC + 1);
    for (int i = 0; i < P; ++i) {
      int c[3];
      for (int k = 0; k < 3; ++k) {
        int v = (int)std::floor((sp[i * 3 + k] - b[k]) * b[4 + k]);
        c[k] = v < 0 ? 0 : (v > G - 1 ? G - 1 : v);
      }
      cell[i] = (c[2] * G + c[1]) * G + c[0];
      st[cell[i] + 1]++;
    }
    for (int c = 0; c < NC; ++c) st[c + 1] += st[c];
    std::vector<int32_t> fill(st, st + NC);
    uint16_t* pp = pts.data() + (size_t)o * P;
    for (int i = 0; i < P; ++i) pp[fill[cell[i]]++] = (uint16_t)i;  // points of a cell in index order
  }
  gqPointGrid* g = new gqPointGrid();
  g->n_obj = (int)n_obj;
  g->P = P;
  g->G = G;
  g->box_dev = nullptr;
  g->start_dev = nullptr;
  g->pts_dev = nullptr;
  GQ_CHECK_HIP(hipMalloc(&g->box_dev, box.size() * sizeof(float)));
  GQ_CHECK_HIP(hipMalloc(&g->start_dev, start.size() * sizeof(int32_t)));
  GQ_CHECK_HIP(hipMalloc(&g->pts_dev, pts.size() * sizeof(uint16_t)));
  GQ_CHECK_HIP(hipMemcpy(g->box_dev, box.data(), box.size() * sizeof(float), hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(g->start_dev, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(g->pts_dev, pts.data(), pts.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  *out = g;
  return GQ_OK;
}

int gq_pointgrid_destroy(gqPointGrid* g) {
  if (!g) return GQ_OK;
  if (g->box_dev) (void)hipFree(g->box_dev);
  if (g->start_dev) (void)hipFree(g->start_dev);
  if (g->pts_dev) (void)hipFree(g->pts_dev);
  delete g;
  return GQ_OK;
}

// The penetration-only query (gq_hand_pen_forward with penetration_only = 1) driven by the links through the point grid.
int gq_hand_pen_forward_cells(const gqMeshSet* links, const gqPointGrid* grid, const float* surface_points, int64_t n_obj,
                              int64_t n_surface, int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                              const float* link_T, float* dis, int32_t* link, float* gvec, void* timer, uint64_t* span,
                              void* stream) {
  GQ_REQUIRE(grid, "hand_pen_forward_cells: null point grid");
  GqPenArgs a{};
  int rc0 = gq_pen_fill(links, surface_points, n_obj, n_surface, batch_each, hand_pose, pose_dim, Rg, link_T, dis, link,
                        gvec, span, &a, grid);
  if (rc0) return rc0;
  GQ_REQUIRE(a.occ && a.cand_off, "hand_pen_forward_cells: the link mesh set has no voxel candidate lists (gq_meshset_build_occupancy)");
  a.dbg = gq_pen_dbg_;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timer) {
    e0 = ((hipEvent_t*)timer)[0];
    e1 = ((hipEvent_t*)timer)[1];
  }
  hipExtLaunchKernelGGL(gq_pen_cells_kernel, dim3((unsigned)a.B), dim3(256), gq_pen_cells_lds_bytes(a.L, a.P),
                        (hipStream_t)stream, e0, e1, 0, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_hand_pen_workspace_bytes(int64_t batch, int64_t n_surface, int n_links, size_t* bytes) {
  GQ_REQUIRE(bytes && batch > 0 && n_surface > 0 && n_links > 0, "hand_pen_workspace_bytes: bad arguments");
  const size_t cap = (size_t)batch * n_surface * n_links;
  *bytes = 1024 + cap * sizeof(GqPenItem) + (size_t)batch * n_surface * 8;
  return GQ_OK;
}

int gq_hand_pen_backward(int n_links, const float* surface_points, int64_t n_obj, int64_t n_surface,
                         int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                         const float* grad_dis, const int32_t* link, const float* gvec, float* link_wrench, float* gRt,
                         const float* dis, float w_pen, float* e_pen, uint64_t* span, uint64_t* span_acc, void* stream) {
  GqPenBwdArgs a{};
  int rc0 = gq_pen_bwd_fill(n_links, surface_points, n_obj, n_surface, batch_each, hand_pose, pose_dim, Rg, grad_dis, link,
                            gvec, link_wrench, gRt, dis, w_pen, e_pen, span, span_acc, &a);
  if (rc0) return rc0;
  if (a.P <= 10 * 256)
    hipLaunchKernelGGL(gq_hand_pen_bwd_kernel<10>, dim3((unsigned)a.B), dim3(256), gq_pen_bwd_lds_bytes(), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(gq_hand_pen_bwd_kernel<GQ_PENB_K>, dim3((unsigned)a.B), dim3(256), gq_pen_bwd_lds_bytes(), (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
