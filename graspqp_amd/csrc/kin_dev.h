// Device pieces of the hand kinematics shared by kin.hip and the stage kernels (stage.hip runs the sphere / self-
// penetration part of a row as a third role next to the force-closure tail and the penetration backward).
#pragma once
#include "common.h"
#include "wave.h"

struct gqHand {
  int J, L, C, S, NG, max_depth;  // J = moving joints of the tree (nodes)
  int JA;                         // actuated joints = pose dimension - 9 (== J unless the hand is coupled)
  float *coup, *coup0;            // (J,JA), (J): theta_tree = coup theta_act + coup0; null = identity
  int32_t *node_parent, *node_type, *link_node, *cand_link, *sphere_link, *group_off;
  int32_t *node_depth, *child_off, *child_idx;  // tree levels and per-node child lists (wave-parallel FK)
  int32_t* sphere_grp;                          // (S) group (= link run) of every penetration sphere
  float *node_pre, *node_axis, *link_offset, *cand_pos, *cand_nrm, *sphere, *jlo, *jhi;
};

// 3x4 row-major [R|t] helpers ------------------------------------------------------------------------------------
struct GqT {
  float m[12];
};
__device__ __forceinline__ GqT gq_t_identity() {
  GqT t;
#pragma unroll
  for (int i = 0; i < 12; ++i) t.m[i] = (i == 0 || i == 5 || i == 10) ? 1.0f : 0.0f;
  return t;
}
__device__ __forceinline__ GqT gq_t_load(const float* p) {
  GqT t;
#pragma unroll
  for (int i = 0; i < 12; ++i) t.m[i] = p[i];
  return t;
}
__device__ __forceinline__ void gq_t_store(float* p, const GqT& t) {
#pragma unroll
  for (int i = 0; i < 12; ++i) p[i] = t.m[i];
}
__device__ __forceinline__ GqT gq_t_mul(const GqT& a, const GqT& b) {
  GqT c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = a.m[i * 4 + 0] * b.m[0 * 4 + j] + a.m[i * 4 + 1] * b.m[1 * 4 + j] + a.m[i * 4 + 2] * b.m[2 * 4 + j];
      if (j == 3) v += a.m[i * 4 + 3];
      c.m[i * 4 + j] = v;
    }
  }
  return c;
}
__device__ __forceinline__ gq3 gq_t_rot(const GqT& t, gq3 v) {
  return gq_mk(t.m[0] * v.x + t.m[1] * v.y + t.m[2] * v.z, t.m[4] * v.x + t.m[5] * v.y + t.m[6] * v.z,
               t.m[8] * v.x + t.m[9] * v.y + t.m[10] * v.z);
}
__device__ __forceinline__ gq3 gq_t_pos(const GqT& t) { return gq_mk(t.m[3], t.m[7], t.m[11]); }
__device__ __forceinline__ gq3 gq_t_apply(const GqT& t, gq3 v) { return gq_t_rot(t, v) + gq_t_pos(t); }

// joint motion: Rodrigues rotation about a unit axis, or translation along it
__device__ __forceinline__ GqT gq_joint_motion(int type, gq3 a, float q) {
  GqT t = gq_t_identity();
  if (type == 1) {
    float s, c;
    sincosf(q, &s, &c);
    const float v = 1.0f - c;
    t.m[0] = c + v * a.x * a.x;
    t.m[1] = v * a.x * a.y - s * a.z;
    t.m[2] = v * a.x * a.z + s * a.y;
    t.m[4] = v * a.y * a.x + s * a.z;
    t.m[5] = c + v * a.y * a.y;
    t.m[6] = v * a.y * a.z - s * a.x;
    t.m[8] = v * a.z * a.x - s * a.y;
    t.m[9] = v * a.z * a.y + s * a.x;
    t.m[10] = c + v * a.z * a.z;
  } else {
    t.m[3] = a.x * q;
    t.m[7] = a.y * q;
    t.m[11] = a.z * q;
  }
  return t;
}

// rot6d (first two columns of R) -> R row-major; roma.special_gramschmidt
__device__ __forceinline__ void gq_rot6d(const float* six, float* R) {
  gq3 a = gq_mk(six[0], six[1], six[2]), b = gq_mk(six[3], six[4], six[5]);
  const gq3 x = (1.0f / sqrtf(gq_dot(a, a))) * a;
  gq3 y = b - gq_dot(x, b) * x;
  y = (1.0f / sqrtf(gq_dot(y, y))) * y;
  const gq3 z = gq_cross(x, y);
  R[0] = x.x; R[1] = y.x; R[2] = z.x;
  R[3] = x.y; R[4] = y.y; R[5] = z.y;
  R[6] = x.z; R[7] = y.z; R[8] = z.z;
}

// order-preserving map of floats onto unsigned (and back)
__device__ __forceinline__ unsigned gq_f2o(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float gq_o2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// one wavefront per row: lane j owns joint node j, then link j, then contacts / spheres j, j+64, ...

struct GqSpenArgs {
  float* spheres;    // (B,S,3) world centres or null
  float* e_spen;     // (B) or null (centres only)
  float* g_spheres;  // (B,S,3): spen_scale * dE_spen / d centre
  float spen_scale;
};

// World centres of the penetration spheres of one row and, with e_spen, their self penetration
// (hand_model.py:989-1040).  One wavefront; LT = the row's link transforms (LDS or global), R / tg = global rotation and
// translation; sC (3 S floats), sKey (64 x u64), sRad (S floats) = LDS scratch of this wavefront.
__device__ __forceinline__ void gq_spheres_row(const gqHand& h, const GqSpenArgs& g, const float* LT, const float* R,
                                               gq3 tg, int row, int lane, float* sC, unsigned long long* sKey,
                                               float* sRad) {
  {
    for (int sidx = lane; sidx < h.S; sidx += GQ_WAVE) {
      const int sl = h.sphere_link[sidx];
      float sp4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) sp4[k] = h.sphere[sidx * 4 + k];
      const GqT T = gq_t_load(LT + sl * 12);
      // explicit fused multiply-adds: this function is inlined into two different kernels (FK forward and the stage-B
      // role) whose results must agree bit for bit, so nothing is left to the compiler's contraction choices
      const gq3 ph = gq_mk(fmaf(T.m[0], sp4[0], fmaf(T.m[1], sp4[1], fmaf(T.m[2], sp4[2], T.m[3]))),
                           fmaf(T.m[4], sp4[0], fmaf(T.m[5], sp4[1], fmaf(T.m[6], sp4[2], T.m[7]))),
                           fmaf(T.m[8], sp4[0], fmaf(T.m[9], sp4[1], fmaf(T.m[10], sp4[2], T.m[11]))));
      const gq3 pw = gq_mk(fmaf(R[0], ph.x, fmaf(R[1], ph.y, fmaf(R[2], ph.z, tg.x))),
                           fmaf(R[3], ph.x, fmaf(R[4], ph.y, fmaf(R[5], ph.z, tg.y))),
                           fmaf(R[6], ph.x, fmaf(R[7], ph.y, fmaf(R[8], ph.z, tg.z))));
      if (g.spheres) {
        float* o = g.spheres + ((size_t)row * h.S + sidx) * 3;
        o[0] = pw.x; o[1] = pw.y; o[2] = pw.z;
      }
      if (g.e_spen) {
        sC[sidx * 3] = pw.x; sC[sidx * 3 + 1] = pw.y; sC[sidx * 3 + 2] = pw.z;
        sRad[sidx] = sp4[3];
      }
    }
  }
  if (g.e_spen) {
    const int ng = min(h.NG - 1, 64);  // the last sphere group has nothing after it
    // Self penetration (hand_model.py:989-1040) on the centres just computed: for every sphere group (= link) the most
    // penetrating pair against all LATER groups.  Lanes take the partners b of one sphere a at a time; the per-group
    // minimum is a 64-bit LDS atomicMin on (pen, a, b) -- the first minimal pair in (a, b) order, like a serial scan.
    sKey[lane] = ~0ull;
    gq_wave_sync();
    // four lanes per group (16 groups per pass): lane (g, q) scans the pairs (a in g) x (b = first later sphere + q,
    // + 4, ...), everything from LDS; the four partial minima of a group meet through two quad DPP steps on the
    // (pen, a, b) key, whose order is the serial scan's order.
    for (int g0 = 0; g0 < ng; g0 += 16) {
      const int gi = g0 + (lane >> 2), q = lane & 3;
      unsigned long long key = ~0ull;
      if (gi < ng) {
        const int a0 = h.group_off[gi], a1 = h.group_off[gi + 1];
        for (int a = a0; a < a1; ++a) {
          const gq3 pa = gq_mk(sC[a * 3], sC[a * 3 + 1], sC[a * 3 + 2]);
          const float ra = sRad[a];
#pragma unroll 4
          for (int b = a1 + q; b < h.S; b += 4) {
            const gq3 d = gq_mk(pa.x - sC[b * 3] + 1e-13f, pa.y - sC[b * 3 + 1] + 1e-13f, pa.z - sC[b * 3 + 2] + 1e-13f);
            const float pen = sqrtf(gq_dot(d, d)) - (ra + sRad[b]);
            const unsigned long long k =
                ((unsigned long long)gq_f2o(pen) << 32) | ((unsigned long long)a << 16) | (unsigned long long)b;
            key = k < key ? k : key;
          }
        }
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int lo = (int)(key & 0xffffffffull), hi = (int)(key >> 32);
        const int olo = st == 0 ? __builtin_amdgcn_mov_dpp(lo, 0xb1, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(lo, 0x4e, 0xf, 0xf, true);
        const int ohi = st == 0 ? __builtin_amdgcn_mov_dpp(hi, 0xb1, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(hi, 0x4e, 0xf, 0xf, true);
        const unsigned long long other = ((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo;
        key = other < key ? other : key;
      }
      if (gi < ng && q == 0) sKey[gi] = key;
    }
    gq_wave_sync();
    // energy: fixed DPP tree over the groups; gradient: lane s collects, in group order, what lands on sphere s
    float e = 0.0f;
    if (lane < ng) {
      const unsigned long long k = sKey[lane];
      const float best = gq_o2f((unsigned)(k >> 32));
      if (k != ~0ull && best < 0.0f) e = -best;
    }
    e = gq_dpp_sum(e);
    if (lane == 0) g.e_spen[row] = e;
    for (int sidx = lane; sidx < h.S; sidx += GQ_WAVE) {
      gq3 acc = gq_mk(0, 0, 0);
      for (int gi = 0; gi < ng; ++gi) {
        const unsigned long long k = sKey[gi];
        const float best = gq_o2f((unsigned)(k >> 32));
        const int ba = (int)((k >> 16) & 0xffffull), bb = (int)(k & 0xffffull);
        if (k == ~0ull || !(best < 0.0f) || (ba != sidx && bb != sidx)) continue;
        const gq3 d = gq_mk(sC[ba * 3] - sC[bb * 3] + 1e-13f, sC[ba * 3 + 1] - sC[bb * 3 + 1] + 1e-13f,
                            sC[ba * 3 + 2] - sC[bb * 3 + 2] + 1e-13f);
        // E += -|a-b| + ... : dE/da = -(a-b)/|a-b|, dE/db = +(a-b)/|a-b|
        const float sc = g.spen_scale / sqrtf(gq_dot(d, d)) * (ba == sidx ? -1.0f : 1.0f);
        acc = gq_mk(fmaf(sc, d.x, acc.x), fmaf(sc, d.y, acc.y), fmaf(sc, d.z, acc.z));
      }
      float* o = g.g_spheres + ((size_t)row * h.S + sidx) * 3;
      o[0] = acc.x; o[1] = acc.y; o[2] = acc.z;
    }
  }
}
