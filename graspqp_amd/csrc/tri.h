// Point-triangle machinery shared by the SDF kernels.
//
// A mesh is kept on the device as an array of 96-byte face records precomputed once per mesh (in double).  A record
// holds the face's own orthonormal frame -- e1 along ab, e2 in the plane, n the unit normal -- and the triangle's 2-D
// coordinates in it: a = (0, 0), b = (L, 0), c = (cx, cy), cy > 0:
//   r0 = a.xyz, L          r1 = e1.xyz, cx          r2 = e2.xyz, cy          r3 = n.xyz, cx - L
//   r4 = cy * L, 1/|ac|^2, 1/|bc|^2, gate            r5 = 0, 0, original face index (int bits), 0
// A query d = p - a becomes (x, y, h) = (e1.d, e2.d, n.d); everything else is 2-D: the projection lies inside the
// triangle when the three edge functions  y,  x cy - y cx,  (cx - L) y - cy x + cy L  are all >= gate (gate = 0; +inf for
// a degenerate face, which then takes the edge path: its record describes the segment it collapses to), and the squared
// distance is h^2 inside, h^2 + the smallest squared distance to the three edge segments otherwise -- the exact closest
// point on a triangle, the same result as the region classification in the oracle (oracle/ref_cpu/sdf.py) up to fp32
// round-off.  All differences are taken between 2-D coordinates of the size of the face, never between |d|^2-sized
// quantities, and the inside test is three well-conditioned edge functions: a sliver face (the fillet strips of the
// Allegro palm have sin^2 of their smallest angle at 3e-6) is classified as reliably as any other.  (Until round 3 the
// record held ab, ac and the barycentric constants (|ac|^2, ab.ac, |ab|^2) / |ab x ac|^2, whose rounding error is
// amplified by 1 / sin^2: queries above a sliver could get the distance to its edge, up to half its width off.)
//
// gq_tri_rank (36 VALU operations) is what the face loops run; gq_tri_finish runs once per query on the winning face and
// reports distance / closest point / sign / normal from the same 2-D residuals.
#pragma once
#include "common.h"

struct GqFace {  // 24 floats
  float4 r0, r1, r2, r3, r4, r5;
};

__device__ __forceinline__ float gq_sat(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }

// the query in the face's frame
struct GqTri2 {
  float x, y, h;
};
__device__ __forceinline__ GqTri2 gq_tri_local(const GqFace& f, gq3 d) {
  GqTri2 q;
  q.x = gq_dot(gq_mk(f.r1.x, f.r1.y, f.r1.z), d);
  q.y = gq_dot(gq_mk(f.r2.x, f.r2.y, f.r2.z), d);
  q.h = gq_dot(gq_mk(f.r3.x, f.r3.y, f.r3.z), d);
  return q;
}

// squared distance from d = p - a to the triangle described by `f`
__device__ __forceinline__ float gq_tri_rank(const GqFace& f, gq3 d) {
  const GqTri2 q = gq_tri_local(f, d);
  const float L = f.r0.w, cx = f.r1.w, cy = f.r2.w, cxl = f.r3.w;
  const float ycy = q.y * cy;
  const float s2 = fmaf(q.x, cy, -(q.y * cx));
  const float s3 = fmaf(cxl, q.y, fmaf(-cy, q.x, f.r4.x));
  const float m = fminf(fminf(q.y, s2), s3);
  const float qa = q.x - __builtin_amdgcn_fmed3f(q.x, 0.0f, L);  // edge ab
  const float ea = fmaf(qa, qa, q.y * q.y);
  const float t2 = gq_sat(fmaf(q.x, cx, ycy) * f.r4.y);  // edge ac
  const float ux = fmaf(-t2, cx, q.x), uy = fmaf(-t2, cy, q.y);
  const float eb = fmaf(ux, ux, uy * uy);
  const float xl = q.x - L;  // edge bc
  const float t3 = gq_sat(fmaf(xl, cxl, ycy) * f.r4.z);
  const float vx = fmaf(-t3, cxl, xl), vy = fmaf(-t3, cy, q.y);
  const float ec = fmaf(vx, vx, vy * vy);
  const float e = fminf(fminf(ea, eb), ec);
  return fmaf(q.h, q.h, (m >= f.r4.w) ? 0.0f : e);
}
__device__ __forceinline__ float gq_tri_dist2(const GqFace& f, gq3 d) { return gq_tri_rank(f, d); }

// full TorchSDF-contract result for the winning face
struct GqSdfOut {
  float dist2;
  int sign;
  gq3 normal, closest;
};
__device__ __forceinline__ GqSdfOut gq_tri_finish(const GqFace& f, gq3 p) {
  const gq3 a = gq_mk(f.r0.x, f.r0.y, f.r0.z);
  const gq3 e1 = gq_mk(f.r1.x, f.r1.y, f.r1.z), e2 = gq_mk(f.r2.x, f.r2.y, f.r2.z), n = gq_mk(f.r3.x, f.r3.y, f.r3.z);
  const GqTri2 q = gq_tri_local(f, p - a);
  const float L = f.r0.w, cx = f.r1.w, cy = f.r2.w, cxl = f.r3.w;
  const float ycy = q.y * cy;
  const float s2 = fmaf(q.x, cy, -(q.y * cx));
  const float s3 = fmaf(cxl, q.y, fmaf(-cy, q.x, f.r4.x));
  const bool inside = fminf(fminf(q.y, s2), s3) >= f.r4.w;
  float rx = 0.0f, ry = 0.0f;  // in-plane residual (query - closest point)
  if (!inside) {
    rx = q.x - __builtin_amdgcn_fmed3f(q.x, 0.0f, L);
    ry = q.y;
    float best = fmaf(rx, rx, ry * ry);
    const float t2 = gq_sat(fmaf(q.x, cx, ycy) * f.r4.y);
    const float ux = fmaf(-t2, cx, q.x), uy = fmaf(-t2, cy, q.y);
    const float eb = fmaf(ux, ux, uy * uy);
    if (eb < best) {
      best = eb;
      rx = ux;
      ry = uy;
    }
    const float xl = q.x - L;
    const float t3 = gq_sat(fmaf(xl, cxl, ycy) * f.r4.z);
    const float vx = fmaf(-t3, cxl, xl), vy = fmaf(-t3, cy, q.y);
    const float ec = fmaf(vx, vx, vy * vy);
    if (ec < best) {
      rx = vx;
      ry = vy;
    }
  }
  const bool degenerate = f.r4.w > 0.0f;
  const gq3 diff = rx * e1 + ry * e2 + q.h * n;
  GqSdfOut o;
  o.dist2 = fmaf(q.h, q.h, fmaf(rx, rx, ry * ry));
  o.closest = p - diff;
  o.sign = (q.h >= 0.0f || degenerate) ? 1 : -1;  // a degenerate face has no normal: dot(p - closest, 0) >= 0
  if (o.dist2 > 0.0f) {
    o.normal = (1.0f / sqrtf(o.dist2)) * diff;
  } else {
    o.normal = degenerate ? gq_mk(0.0f, 0.0f, 0.0f) : n;
  }
  return o;
}

// corners b and c of a record (set-up kernels; equal to the mesh's own up to fp32 round-off of the frame)
__device__ __forceinline__ void gq_face_corners(const GqFace& f, gq3& a, gq3& b, gq3& c) {
  const gq3 e1 = gq_mk(f.r1.x, f.r1.y, f.r1.z), e2 = gq_mk(f.r2.x, f.r2.y, f.r2.z);
  a = gq_mk(f.r0.x, f.r0.y, f.r0.z);
  b = a + f.r0.w * e1;
  c = a + f.r1.w * e1 + f.r2.w * e2;
}

// squared distance from p to an axis-aligned box stored as 8 floats: lo.xyz, -, hi.xyz, -
__device__ __forceinline__ float gq_aabb_dist2(const float* bb, gq3 p) {
  const float ex = fmaxf(fmaxf(bb[0] - p.x, p.x - bb[4]), 0.0f);
  const float ey = fmaxf(fmaxf(bb[1] - p.y, p.y - bb[5]), 0.0f);
  const float ez = fmaxf(fmaxf(bb[2] - p.z, p.z - bb[6]), 0.0f);
  return fmaf(ex, ex, fmaf(ey, ey, ez * ez));
}

// build one face record from three corners; `orig` = index of the face in the caller's ordering
__device__ __forceinline__ GqFace gq_make_face(gq3 a, gq3 b, gq3 c, int orig = 0) {
  // set-up time only: the frame and the 2-D coordinates are formed in double
  const double ab[3] = {(double)b.x - a.x, (double)b.y - a.y, (double)b.z - a.z};
  const double ac[3] = {(double)c.x - a.x, (double)c.y - a.y, (double)c.z - a.z};
  const double AA = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2], CC = ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2];
  const double L = sqrt(AA), lc = sqrt(CC);
  double e1[3] = {1.0, 0.0, 0.0};
  if (L > 0.0) {
    for (int k = 0; k < 3; ++k) e1[k] = ab[k] / L;
  } else if (lc > 0.0) {
    for (int k = 0; k < 3; ++k) e1[k] = ac[k] / lc;
  }
  double cx = ac[0] * e1[0] + ac[1] * e1[1] + ac[2] * e1[2];
  double pr[3] = {ac[0] - cx * e1[0], ac[1] - cx * e1[1], ac[2] - cx * e1[2]};
  double cy = sqrt(pr[0] * pr[0] + pr[1] * pr[1] + pr[2] * pr[2]);
  const bool ok = L > 0.0 && cy > 1e-6 * lc && L * cy > 1e-15;  // else: the face collapses to a segment along e1
  double e2[3];
  if (ok) {
    for (int k = 0; k < 3; ++k) e2[k] = pr[k] / cy;
  } else {  // any unit vector orthogonal to e1
    cy = 0.0;
    const int j = fabs(e1[0]) <= fabs(e1[1]) ? (fabs(e1[0]) <= fabs(e1[2]) ? 0 : 2) : (fabs(e1[1]) <= fabs(e1[2]) ? 1 : 2);
    double ax[3] = {0.0, 0.0, 0.0};
    ax[j] = 1.0;
    const double t = e1[j];
    double nrm = 0.0;
    for (int k = 0; k < 3; ++k) {
      e2[k] = ax[k] - t * e1[k];
      nrm += e2[k] * e2[k];
    }
    nrm = sqrt(nrm);
    for (int k = 0; k < 3; ++k) e2[k] /= nrm;
  }
  const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
  const double BC = (cx - L) * (cx - L) + cy * cy, CC2 = cx * cx + cy * cy;
  GqFace f;
  f.r0 = make_float4(a.x, a.y, a.z, (float)L);
  f.r1 = make_float4((float)e1[0], (float)e1[1], (float)e1[2], (float)cx);
  f.r2 = make_float4((float)e2[0], (float)e2[1], (float)e2[2], (float)cy);
  f.r3 = make_float4((float)n[0], (float)n[1], (float)n[2], (float)(cx - L));
  f.r4 = make_float4((float)(cy * L), CC2 > 0.0 ? (float)(1.0 / CC2) : 0.0f, BC > 0.0 ? (float)(1.0 / BC) : 0.0f,
                     ok ? 0.0f : GQ_INF_F);
  f.r5 = make_float4(0.0f, 0.0f, __int_as_float(orig), 0.0f);
  return f;
}
