// Device body of the wave-per-query mesh distance (object SDF of the contact points), shared by sdf.hip (stand-alone
// kernel) and kin.hip (the FK forward kernel answers the contact queries of its row in the same launch).
#pragma once
#include "tri.h"

// ---- wave per query -------------------------------------------------------------------------------------------
// queries are grouped: query q uses mesh (q / queries_per_mesh); mesh m's records are rec[off[m] .. off[m+1]).
// With a mesh set the faces of every mesh are Morton-sorted and grouped in clusters of 64 (one face per lane) with an
// AABB each: the wave first evaluates the cluster whose box is nearest to the query, then visits only clusters whose
// box is not farther than the best distance found so far -- an exact search (lower bound vs. running minimum).
struct GqWaveArgs {
  const float* points;
  int64_t N;
  const GqFace* rec;
  const int32_t* off;      // (n_mesh+1) or null (single soup of single_F faces, no clusters)
  const float* cl_aabb;    // (n_cl64, 16) oriented cluster boxes (gq_cluster_bound) or null
  const int32_t* cl_off;   // (n_mesh+1) first 64-cluster of each mesh
  int single_F;
  int64_t queries_per_mesh;
  // XCD-aware placement (mesh sets with several meshes): blocks b and b + 8 share an XCD (round-robin dispatch), so block
  // b serves mesh (b % 8) + 8 j only -- every XCD's L2 then holds its own meshes instead of all of them.
  int xcd_meshes;       // 0: plain mapping; else ceil(n_mesh / 8) = meshes per XCD
  int blocks_per_mesh;  // ceil(queries_per_mesh / wavefronts per block)
  int n_mesh;
  unsigned long long* dbg;  // diagnostics (gq_debug_set_pen_counters): [0] += cluster visits, [1] += queries
  float* dist_sq;
  int32_t* sign;
  float* normal;
  float* closest;
};

// lower bound of the squared distance from p to any face of a cluster (oriented box of gq_cluster_bound)
__device__ __forceinline__ float gq_cluster_lb(const float* __restrict__ r, gq3 p) {
  const float4 c = *reinterpret_cast<const float4*>(r), u = *reinterpret_cast<const float4*>(r + 4),
               v = *reinterpret_cast<const float4*>(r + 8), n = *reinterpret_cast<const float4*>(r + 12);
  const gq3 d = gq_mk(p.x - c.x, p.y - c.y, p.z - c.z);
  const float eu = fmaxf(fabsf(fmaf(d.x, u.x, fmaf(d.y, u.y, d.z * u.z))) - c.w, 0.0f);
  const float ev = fmaxf(fabsf(fmaf(d.x, v.x, fmaf(d.y, v.y, d.z * v.z))) - u.w, 0.0f);
  const float en = fmaxf(fabsf(fmaf(d.x, n.x, fmaf(d.y, n.y, d.z * n.z))) - v.w, 0.0f);
  return fmaf(eu, eu, fmaf(ev, ev, en * en));
}

__device__ __forceinline__ void gq_wave_eval_cluster(const GqFace* __restrict__ rec, int f, int f1, gq3 p, float& best,
                                                     unsigned& borig, int& bi) {
  if (f < f1) {
    const GqFace fc = rec[f];
    const gq3 d = p - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z);
    const float d2 = gq_tri_rank(fc, d);
    const unsigned orig = (unsigned)__float_as_int(fc.r5.z);
    if (d2 < best || (d2 == best && orig < borig)) {
      best = d2;
      borig = orig;
      bi = f;
    }
  }
}

// What a query can load before its point is known: mesh offsets and the oriented boxes of the first 256 clusters
// (lane l: clusters l, l + 64, l + 128, l + 192).  In the FK forward kernel the query wavefronts fetch this while
// wavefront 0 is still computing the contact points.
struct GqSdfPre {
  int f0, f1, c0, nC;
  float4 box[4][4];
};
__device__ __forceinline__ GqSdfPre gq_sdf_wave_prefetch(const GqWaveArgs& g, int64_t q, int lane) {
  GqSdfPre s;
  const int mesh = (int)(q / g.queries_per_mesh);
  s.f0 = g.off ? g.off[mesh] : 0;
  s.f1 = g.off ? g.off[mesh + 1] : g.single_F;
  s.c0 = 0;
  s.nC = 0;
  if (g.cl_aabb) {
    s.c0 = g.cl_off[mesh];
    s.nC = g.cl_off[mesh + 1] - s.c0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = k * GQ_WAVE + lane;
      const float4* r = reinterpret_cast<const float4*>(g.cl_aabb + (size_t)(s.c0 + (c < s.nC ? c : 0)) * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) s.box[k][j] = r[j];
    }
  }
  return s;
}
__device__ __forceinline__ float gq_cluster_lb4(const float4 (&r)[4], gq3 p) {
  const gq3 d = gq_mk(p.x - r[0].x, p.y - r[0].y, p.z - r[0].z);
  const float eu = fmaxf(fabsf(fmaf(d.x, r[1].x, fmaf(d.y, r[1].y, d.z * r[1].z))) - r[0].w, 0.0f);
  const float ev = fmaxf(fabsf(fmaf(d.x, r[2].x, fmaf(d.y, r[2].y, d.z * r[2].z))) - r[1].w, 0.0f);
  const float en = fmaxf(fabsf(fmaf(d.x, r[3].x, fmaf(d.y, r[3].y, d.z * r[3].z))) - r[2].w, 0.0f);
  return fmaf(eu, eu, fmaf(ev, ev, en * en));
}

// one wavefront answers query q at point p (all lanes pass the same q, p).  GQ_TOPK = clusters taken up per round: four
// record loads in flight make a round one L2 round trip for four clusters (latency: the FK forward block, small
// launches); two keep 40 registers less alive and tighten the bound after every second cluster (throughput: large
// launches, where registers x lifetime is what the kernels next to it compete for).  The answer does not depend on it:
// the search is exact and ties go to the smallest original face index.
template <int GQ_TOPK = 4>
__device__ __forceinline__ void gq_sdf_wave_query(const GqWaveArgs& g, int64_t q, gq3 p, int lane, const GqSdfPre& pre) {
  const int f0 = pre.f0, f1 = pre.f1;
  float best = GQ_INF_F;
  unsigned borig = 0xffffffffu;
  int bi = -1;
  if (g.cl_aabb == nullptr) {
    for (int f = f0 + lane; f < f1; f += GQ_WAVE) gq_wave_eval_cluster(g.rec, f, f1, p, best, borig, bi);
  } else {
    const int c0 = pre.c0, nC = pre.nC;
    // Best-first over the 64-face clusters: lane l keeps the lower bounds of clusters cb + l, cb + 64 + l, ... in
    // registers.  Each round takes the (up to) GQ_TOPK unvisited clusters with the smallest bounds that can still beat
    // the best distance found, loads their faces together (one face per lane and cluster -- the GQ_TOPK record loads
    // are in flight at once, which is what matters: a round is one L2 round trip) and evaluates them.  It stops as
    // soon as the smallest remaining bound exceeds the best distance: for a point at distance d only clusters whose
    // box intersects the ball of radius d are ever touched.
    constexpr int KC = 4;  // 256 clusters (16384 faces) per pass
    float ub = GQ_INF_F;
    int visits = 0;
    float lb[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k)  // first pass: the boxes fetched ahead of time (dead after this)
      lb[k] = (k * GQ_WAVE + lane < nC) ? gq_cluster_lb4(pre.box[k], p) * 0.9999f : GQ_INF_F;
    for (int cb = 0; cb < nC; cb += KC * GQ_WAVE) {
      if (cb > 0) {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          const int c = cb + k * GQ_WAVE + lane;
          lb[k] = (c < nC) ? gq_cluster_lb(g.cl_aabb + (size_t)(c0 + c) * 16, p) * 0.9999f : GQ_INF_F;
        }
      }
      for (;;) {
        int pick[GQ_TOPK];
        // which clusters can still beat the best distance?  (wave-uniform bit masks, one per register slot)
        unsigned long long cm[KC];
        int n_cand = 0;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          cm[k] = __ballot(lb[k] <= ub && lb[k] < GQ_INF_F);
          n_cand += __popcll(cm[k]);
        }
        if (n_cand == 0) break;
        if (n_cand <= GQ_TOPK) {
          // few left (the usual case after the first round): take them all, straight from the masks -- scalar bit
          // operations instead of GQ_TOPK wave-wide minimum reductions
#pragma unroll
          for (int j = 0; j < GQ_TOPK; ++j) {
            int pk = -1;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
              if (pk < 0 && cm[k] != 0ull) {
                const int src = __ffsll((long long)cm[k]) - 1;
                pk = cb + k * GQ_WAVE + src;
                cm[k] &= cm[k] - 1ull;
              }
            }
            pick[j] = pk;
          }
#pragma unroll
          for (int k = 0; k < KC; ++k)
            if (lb[k] <= ub) lb[k] = GQ_INF_F;  // visited
        } else {
#pragma unroll
          for (int j = 0; j < GQ_TOPK; ++j) {
            float m = lb[0];
            int mk = 0;
#pragma unroll
            for (int k = 1; k < KC; ++k) {
              if (lb[k] < m) {
                m = lb[k];
                mk = k;
              }
            }
            const float mw = gq_dpp_min(m);
            pick[j] = -1;
            if (mw <= ub && mw < GQ_INF_F) {  // wave-uniform; false when everything is visited (inf) or NaN
              const unsigned long long who = __ballot(m == mw);
              const int src = __ffsll((long long)who) - 1;
              pick[j] = cb + gq_readlane_i(mk, src) * GQ_WAVE + src;
              if (lane == src) {
#pragma unroll
                for (int k = 0; k < KC; ++k)
                  if (k == mk) lb[k] = GQ_INF_F;
              }
            }
          }
        }
        if (pick[0] < 0) break;
        GqFace fc[GQ_TOPK];
#pragma unroll
        for (int j = 0; j < GQ_TOPK; ++j) {
          const int f = f0 + pick[j] * 64 + lane;
          if (pick[j] >= 0 && f < f1) fc[j] = g.rec[f];
        }
#pragma unroll
        for (int j = 0; j < GQ_TOPK; ++j) {
          const int f = f0 + pick[j] * 64 + lane;
          if (pick[j] >= 0 && f < f1) {
            const gq3 d = p - gq_mk(fc[j].r0.x, fc[j].r0.y, fc[j].r0.z);
            const float d2 = gq_tri_rank(fc[j], d);
            const unsigned orig = (unsigned)__float_as_int(fc[j].r5.z);
            if (d2 < best || (d2 == best && orig < borig)) {
              best = d2;
              borig = orig;
              bi = f;
            }
          }
          visits += pick[j] >= 0;
        }
        ub = gq_dpp_min(best);
      }
    }
    if (g.dbg && lane == 0) {
      atomicAdd(&g.dbg[0], (unsigned long long)visits);
      atomicAdd(&g.dbg[1], 1ull);
      atomicMax(&g.dbg[2], (unsigned long long)visits);
      if (visits > 16) atomicAdd(&g.dbg[3], 1ull);
    }
  }
  // winner = smallest distance, then smallest original face index (two DPP min passes)
  const float dmin = gq_dpp_min(best);
  const bool tie = (best == dmin) && (bi >= 0);
  const float omin = gq_dpp_min(tie ? (float)borig : GQ_INF_F);  // face indices < 2^24 are exact in fp32
  const unsigned long long win = __ballot(tie && (float)borig == omin);
  const int face = win ? gq_readlane_i(bi, __ffsll((long long)win) - 1) : -1;
  if (lane == 0) {
    GqSdfOut o;
    if (face >= f0 && face < f1) {
      o = gq_tri_finish(g.rec[face], p);
    } else {  // empty mesh or all-NaN distances
      o.dist2 = GQ_INF_F;
      o.sign = 1;
      o.normal = gq_mk(0, 0, 0);
      o.closest = p;
    }
    g.dist_sq[q] = o.dist2;
    g.sign[q] = o.sign;
    if (g.normal) {
      g.normal[q * 3 + 0] = o.normal.x;
      g.normal[q * 3 + 1] = o.normal.y;
      g.normal[q * 3 + 2] = o.normal.z;
    }
    g.closest[q * 3 + 0] = o.closest.x;
    g.closest[q * 3 + 1] = o.closest.y;
    g.closest[q * 3 + 2] = o.closest.z;
  }
}

