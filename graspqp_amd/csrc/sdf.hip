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
  const float* cl_aabb;    // (n_cl64, 8) or null
  const int32_t* cl_off;   // (n_mesh+1) first 64-cluster of each mesh
  int single_F;
  int64_t queries_per_mesh;
  float* dist_sq;
  int32_t* sign;
  float* normal;
  float* closest;
};

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

__global__ __launch_bounds__(256) void gq_sdf_wave_kernel(GqWaveArgs g) {
  const int64_t q = (int64_t)blockIdx.x * (blockDim.x / GQ_WAVE) + (threadIdx.x / GQ_WAVE);
  if (q >= g.N) return;
  const int lane = gq_lane();
  const int mesh = (int)(q / g.queries_per_mesh);
  const int f0 = g.off ? g.off[mesh] : 0, f1 = g.off ? g.off[mesh + 1] : g.single_F;
  const gq3 p = gq_mk(g.points[q * 3 + 0], g.points[q * 3 + 1], g.points[q * 3 + 2]);
  float best = GQ_INF_F;
  unsigned borig = 0xffffffffu;
  int bi = -1;
  if (g.cl_aabb == nullptr) {
    for (int f = f0 + lane; f < f1; f += GQ_WAVE) gq_wave_eval_cluster(g.rec, f, f1, p, best, borig, bi);
  } else {
    const int c0 = g.cl_off[mesh], nC = g.cl_off[mesh + 1] - c0;
    // nearest cluster box first
    float lbmin = GQ_INF_F;
    int cmin = 0;
    for (int c = lane; c < nC; c += GQ_WAVE) {
      const float lb = gq_aabb_dist2(g.cl_aabb + (size_t)(c0 + c) * 8, p);
      if (lb < lbmin) {
        lbmin = lb;
        cmin = c;
      }
    }
    unsigned long long k = ((unsigned long long)__float_as_uint(lbmin) << 32) | (unsigned)cmin;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(k, o, GQ_WAVE);
      k = other < k ? other : k;
    }
    cmin = (int)(k & 0xffffffffu);
    gq_wave_eval_cluster(g.rec, f0 + cmin * 64 + lane, f1, p, best, borig, bi);
    float ub = best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ub = fminf(ub, __shfl_xor(ub, o, GQ_WAVE));
    for (int cb = 0; cb < nC; cb += GQ_WAVE) {
      const int c = cb + lane;
      const float lb = (c < nC && c != cmin) ? gq_aabb_dist2(g.cl_aabb + (size_t)(c0 + c) * 8, p) : GQ_INF_F;
      unsigned long long mask = __ballot(lb * 0.9999f <= ub);
      while (mask) {
        const int s = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        if (gq_readlane(lb, s) * 0.9999f > ub) continue;  // ub may have shrunk meanwhile
        gq_wave_eval_cluster(g.rec, f0 + (cb + s) * 64 + lane, f1, p, best, borig, bi);
        float m = best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, GQ_WAVE));
        ub = m;
      }
    }
  }
  unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | borig;
  unsigned long long kmin = key;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(kmin, o, GQ_WAVE);
    kmin = other < kmin ? other : kmin;
  }
  const unsigned long long win = __ballot(key == kmin && bi >= 0);
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

// ---- hand penetration: max over links of the signed distance (inside positive) of object surface points ----------
// link_T: (B, L, 12) row-major [R | t] of each mesh link in the hand frame; Rg (B,9) global rotation; hand_pose (B,D)
// holds the global translation in its first three entries.  Outputs per (row, point): dis, argmax link, and
// gvec = d dis / d x_h (hand frame).
struct GqPenArgs {
  const float* surf;  // (n_obj, P, 3)
  const float* hand_pose;
  const float* Rg;
  const float* link_T;
  const GqFace* rec;
  const int32_t* off;  // (L+1)
  const float* aabb;   // (L,8) lo.xyz,-,hi.xyz,-
  const float* sub_aabb;   // (n_sub,8) boxes of the 16-face sub-clusters (faces Morton-sorted per link)
  const int32_t* sub_off;  // (L+1)
  int B, P, L, D, batch_each;
  float* dis;     // (B, P)
  int32_t* link;  // (B, P)
  float* gvec;    // (B, P, 3)
};

// MODE 0 ("exact"): dis is the exact max over links for every point.  A link is skipped for a whole wavefront
//   only when, for every lane, its AABB lower bound already proves dis_l <= best (points outside a link's AABB
//   are outside the link, so dis_l = -sqrt(d_l^2+1e-8) <= -sqrt(lb^2+1e-8)).
// MODE 1 ("penetration only"): what E_pen needs (energy.py:59-61 zeroes dis <= 0): a link is evaluated only if
//   some lane's point lies inside its AABB; dis is exact wherever it is > 0 and merely <= 0 elsewhere.
template <int MODE>
__global__ __launch_bounds__(256) void gq_hand_pen_kernel(GqPenArgs g) {
  const int row = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
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
    if (MODE == 1) {
      need = ok && (lb2 <= 0.0f);
    } else {
      // can link l still beat best_dis?  only if it may be penetrated (inside AABB) or closer than the best so far
      need = ok && ((lb2 <= 0.0f) || (best_dis < 0.0f && fmaf(lb2, 0.9999f, 1e-8f) < best_dis * best_dis));
    }
    if (__ballot(need) == 0ull) continue;  // wave-uniform skip
    // exact nearest face of link l for the lanes that need it: visit 16-face sub-clusters, skipping (for the whole
    // wave) those whose box is farther than every needing lane's running minimum
    float bd = GQ_INF_F;
    unsigned bo = 0xffffffffu;
    int bi = f0;
    const int s0 = g.sub_off[l], s1 = g.sub_off[l + 1];
    for (int sc = s0; sc < s1; ++sc) {
      const float lbs = gq_aabb_dist2(g.sub_aabb + (size_t)sc * 8, xl);
      if (__ballot(need && lbs * 0.9999f <= bd) == 0ull) continue;
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
  if (!ok) return;
  const size_t o = (size_t)row * g.P + pt;
  g.dis[o] = (best_dis == -GQ_INF_F) ? -1e30f : best_dis;
  g.link[o] = best_link;
  g.gvec[o * 3 + 0] = best_g.x;
  g.gvec[o * 3 + 1] = best_g.y;
  g.gvec[o * 3 + 2] = best_g.z;
}

// Backward of the hand-penetration query for an upstream gradient w (B,P) on `dis`:
//   link wrench (hand frame, about the hand origin): f_l -= w G, m_l -= w x_h x G     (G = gvec)
//   gRt[0..2]  = sum w G   (so that grad_t = -R gsum)
//   gRt[3..11] = sum w x_h (x) G  (row-major K, so that grad_R = R K)
// One block per row; contributions are folded in a fixed order (lane order within a wave, wave order within the
// block) so the result is bitwise reproducible.
struct GqPenBwdArgs {
  const float* surf;
  const float* hand_pose;
  const float* Rg;
  const float* w;
  const int32_t* link;
  const float* gvec;
  int B, P, L, D, batch_each;
  float* wrench;  // (B, L, 6)
  float* gRt;     // (B, 12)
};

__global__ __launch_bounds__(256) void gq_hand_pen_bwd_kernel(GqPenBwdArgs g) {
  extern __shared__ float sm[];  // [4 waves][L*6 + 12]
  const int row = blockIdx.x;
  const int tid = threadIdx.x, lane = gq_lane(), wv = tid / GQ_WAVE;
  const int stride = g.L * 6 + 12;
  for (int i = tid; i < 4 * stride; i += 256) sm[i] = 0.0f;
  __syncthreads();
  float* acc = sm + wv * stride;
  const int obj = row / g.batch_each;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  for (int base = 0; base < g.P; base += 256) {
    const int pt = base + tid;
    float w = 0.0f;
    int lk = 0;
    gq3 G = gq_mk(0, 0, 0), xh = gq_mk(0, 0, 0);
    if (pt < g.P) {
      const size_t o = (size_t)row * g.P + pt;
      w = g.w[o];
      if (w != 0.0f) {
        lk = g.link[o];
        G = w * gq_mk(g.gvec[o * 3], g.gvec[o * 3 + 1], g.gvec[o * 3 + 2]);
        const float* sp = g.surf + ((size_t)obj * g.P + pt) * 3;
        xh = gq_mtv(R, gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]));
      }
    }
    unsigned long long mask = __ballot(w != 0.0f);
    while (mask) {  // wave-uniform loop over contributing lanes, in lane order
      const int s = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int l = gq_readlane_i(lk, s);
      const gq3 Gs = gq_mk(gq_readlane(G.x, s), gq_readlane(G.y, s), gq_readlane(G.z, s));
      const gq3 xs = gq_mk(gq_readlane(xh.x, s), gq_readlane(xh.y, s), gq_readlane(xh.z, s));
      const gq3 ms = gq_cross(xs, Gs);
      if (lane == 0) {
        float* a = acc + l * 6;
        a[0] -= Gs.x;
        a[1] -= Gs.y;
        a[2] -= Gs.z;
        a[3] -= ms.x;
        a[4] -= ms.y;
        a[5] -= ms.z;
        float* k = acc + g.L * 6;
        k[0] += Gs.x;
        k[1] += Gs.y;
        k[2] += Gs.z;
        k[3] += xs.x * Gs.x;
        k[4] += xs.x * Gs.y;
        k[5] += xs.x * Gs.z;
        k[6] += xs.y * Gs.x;
        k[7] += xs.y * Gs.y;
        k[8] += xs.y * Gs.z;
        k[9] += xs.z * Gs.x;
        k[10] += xs.z * Gs.y;
        k[11] += xs.z * Gs.z;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < stride; i += 256) {
    const float v = ((sm[i] + sm[stride + i]) + sm[2 * stride + i]) + sm[3 * stride + i];
    if (i < g.L * 6) g.wrench[(size_t)row * g.L * 6 + i] = v;
    else g.gRt[(size_t)row * 12 + (i - g.L * 6)] = v;
  }
}

// ---- mesh-set handle: concatenated face records of n_mesh meshes on the device -----------------------------------
struct gqMeshSet {
  GqFace* rec;         // records, faces Morton-sorted inside each mesh
  int32_t* off_dev;    // (n_mesh+1) face offsets
  int32_t* off_host;
  float* aabb_dev;     // (n_mesh, 8) box of each mesh in its own frame
  float* sub_aabb_dev; // (n_sub, 8) boxes of 16-face sub-clusters
  int32_t* sub_off_dev;   // (n_mesh+1)
  float* cl_aabb_dev;  // (n_cl, 8) boxes of 64-face clusters
  int32_t* cl_off_dev;    // (n_mesh+1)
  int n_mesh;
  int64_t n_faces;
};

#include <algorithm>
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

extern "C" {

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
      cl_bb.resize(cl_bb.size() + 8);
      gq_box_of(face_verts_host, perm.data(), i, std::min<int64_t>(i + 64, b), &cl_bb[cl_bb.size() - 8]);
    }
    sub_off[m + 1] = (int32_t)(sub_bb.size() / 8);
    cl_off[m + 1] = (int32_t)(cl_bb.size() / 8);
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

int gq_meshset_destroy(gqMeshSet* ms) {
  if (!ms) return GQ_OK;
  (void)hipFree(ms->rec);
  (void)hipFree(ms->off_dev);
  (void)hipFree(ms->aabb_dev);
  (void)hipFree(ms->sub_aabb_dev);
  (void)hipFree(ms->cl_aabb_dev);
  (void)hipFree(ms->sub_off_dev);
  (void)hipFree(ms->cl_off_dev);
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
    hipLaunchKernelGGL(gq_sdf_wave_kernel, dim3((unsigned)((n_points + 3) / 4)), dim3(256), 0, st, w);
  }
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Query against a mesh set: points (n_mesh * queries_per_mesh, 3); query q uses mesh q / queries_per_mesh
// (object_model.py:217-220: one mesh per object, batch_size_each * n_contact queries each).
int gq_sdf_forward_meshset(const gqMeshSet* ms, const float* points, int64_t n_points, int64_t queries_per_mesh,
                           float* dist_sq, int32_t* sign, float* normal, float* closest, void* stream) {
  if (n_points == 0) return GQ_OK;
  GQ_REQUIRE(ms && points && dist_sq && sign && closest, "sdf_forward_meshset: null pointer");
  GQ_REQUIRE(queries_per_mesh > 0 && n_points == queries_per_mesh * ms->n_mesh,
             "sdf_forward_meshset: n_points=%lld != queries_per_mesh=%lld * n_mesh=%d", (long long)n_points,
             (long long)queries_per_mesh, ms->n_mesh);
  GqWaveArgs w{};
  w.points = points;
  w.N = n_points;
  w.rec = ms->rec;
  w.off = ms->off_dev;
  w.cl_aabb = ms->cl_aabb_dev;
  w.cl_off = ms->cl_off_dev;
  w.queries_per_mesh = queries_per_mesh;
  w.dist_sq = dist_sq;
  w.sign = sign;
  w.normal = normal;
  w.closest = closest;
  hipLaunchKernelGGL(gq_sdf_wave_kernel, dim3((unsigned)((n_points + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w);
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
                        int penetration_only, float* dis, int32_t* link, float* gvec, void* timer, void* stream) {
  GQ_REQUIRE(links && surface_points && hand_pose && Rg && link_T && dis && link && gvec, "hand_pen_forward: null");
  GQ_REQUIRE(n_obj > 0 && n_surface > 0 && batch_each > 0 && pose_dim >= 9, "hand_pen_forward: bad sizes");
  GqPenArgs a{};
  a.surf = surface_points;
  a.hand_pose = hand_pose;
  a.Rg = Rg;
  a.link_T = link_T;
  a.rec = links->rec;
  a.off = links->off_dev;
  a.aabb = links->aabb_dev;
  a.sub_aabb = links->sub_aabb_dev;
  a.sub_off = links->sub_off_dev;
  a.B = (int)(n_obj * batch_each);
  a.P = (int)n_surface;
  a.L = links->n_mesh;
  a.D = pose_dim;
  a.batch_each = (int)batch_each;
  a.dis = dis;
  a.link = link;
  a.gvec = gvec;
  GQ_REQUIRE(a.B <= 65535, "hand_pen_forward: B=%d exceeds grid.y limit", a.B);
  const dim3 grid((unsigned)((a.P + 255) / 256), (unsigned)a.B);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timer) {  // gqTimer: the kernel's own start/stop timestamps (hipExtLaunchKernelGGL), not stream markers
    e0 = ((hipEvent_t*)timer)[0];
    e1 = ((hipEvent_t*)timer)[1];
  }
  if (penetration_only)
    hipExtLaunchKernelGGL(gq_hand_pen_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  else
    hipExtLaunchKernelGGL(gq_hand_pen_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_hand_pen_backward(int n_links, const float* surface_points, int64_t n_obj, int64_t n_surface,
                         int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                         const float* grad_dis, const int32_t* link, const float* gvec, float* link_wrench, float* gRt,
                         void* stream) {
  GQ_REQUIRE(surface_points && hand_pose && Rg && grad_dis && link && gvec && link_wrench && gRt,
             "hand_pen_backward: null");
  GQ_REQUIRE(n_links > 0 && n_links <= 256 && n_obj > 0 && n_surface > 0 && batch_each > 0, "hand_pen_backward: sizes");
  GqPenBwdArgs a{};
  a.surf = surface_points;
  a.hand_pose = hand_pose;
  a.Rg = Rg;
  a.w = grad_dis;
  a.link = link;
  a.gvec = gvec;
  a.B = (int)(n_obj * batch_each);
  a.P = (int)n_surface;
  a.L = n_links;
  a.D = pose_dim;
  a.batch_each = (int)batch_each;
  a.wrench = link_wrench;
  a.gRt = gRt;
  const size_t shm = (size_t)4 * (n_links * 6 + 12) * sizeof(float);
  hipLaunchKernelGGL(gq_hand_pen_bwd_kernel, dim3((unsigned)a.B), dim3(256), shm, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
