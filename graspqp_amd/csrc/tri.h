// Point-triangle machinery shared by the SDF kernels.
//
// A mesh is kept on the device as an array of 96-byte face records precomputed once per mesh, so that the inner
// loop needs no per-lane division and no per-lane edge vectors:
//   r0 = a.xyz, 1/|ab|^2      r1 = ab.xyz, 1/|ac|^2      r2 = ac.xyz, 1/|bc|^2
//   r3 = |ac|^2/nn, (ab.ac)/nn, |ab|^2/nn, |ab|^2 - ab.ac   (nn = |ab x ac|^2; NaN for degenerate faces -> edge path)
//   r4 = unit normal.xyz, |ab|^2                            r5 = |ac|^2, |bc|^2, original face index (int bits), 0
// Squared distance = plane-projection distance when the projection's barycentrics are all >= 0, else the minimum
// over the three clamped edge projections -- the exact closest point on a triangle (same result as the region
// classification in the oracle, oracle/ref_cpu/sdf.py, up to fp32 round-off).
//
// Two evaluators: gq_tri_rank (45 VALU ops, used inside the face loops: distances from |d|^2 and dot products,
// absolute error ~1e-10 m^2 -- good for RANKING faces) and gq_tri_finish (direct differences, run once per query
// on the winning face: this is the distance / closest point / sign that is reported).
//
// Known limit (both evaluators; measured in round 3, tests/test_gpu_parity.py::test_sdf_box_hierarchy_...): the inside test
// uses the precomputed barycentric constants r3 = (|ac|^2, ab.ac, |ab|^2) / nn, whose rounding error is amplified by
// 1 / sin^2 of the face's smallest angle.  For SLIVER faces (sin^2 < ~1e-4: the fillet strips of the Allegro palm reach
// 3e-6) a query that projects inside the sliver can be classified as outside and gets the distance to the sliver's edge
// instead -- an error of at most half the sliver's width (2e-4 m on that mesh), only for queries closer to the face than
// that.  Signs are unaffected (the offset keeps its normal component).  A robust classification (three edge-plane tests,
// +27 operations) would cost 60 % more per face; TorchSDF's own fp32 answer at such points is not pinned either.
#pragma once
#include "common.h"

struct GqFace {  // 24 floats
  float4 r0, r1, r2, r3, r4, r5;
};

__device__ __forceinline__ float gq_sat(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }

// ranking distance: |p - closest|^2 up to ~1e-10 absolute (see header comment)
__device__ __forceinline__ float gq_tri_rank(const GqFace& f, gq3 d) {
  const gq3 ab = gq_mk(f.r1.x, f.r1.y, f.r1.z), ac = gq_mk(f.r2.x, f.r2.y, f.r2.z);
  const float dd = gq_dot(d, d);
  const float d1 = gq_dot(ab, d), d2 = gq_dot(ac, d);
  const float v = fmaf(f.r3.x, d1, -f.r3.y * d2);
  const float w = fmaf(f.r3.z, d2, -f.r3.y * d1);
  const float u = (1.0f - v) - w;
  const float m = fminf(fminf(v, w), u);
  const float h = gq_dot(gq_mk(f.r4.x, f.r4.y, f.r4.z), d);
  const float dp = h * h;
  const float AA = f.r4.w, CC = f.r5.x, BC = f.r5.y;
  const float t1 = gq_sat(d1 * f.r0.w);
  const float e1 = fmaf(-t1, fmaf(-t1, AA, d1) + d1, dd);
  const float t2 = gq_sat(d2 * f.r1.w);
  const float e2 = fmaf(-t2, fmaf(-t2, CC, d2) + d2, dd);
  const float gbc = (d2 - d1) + f.r3.w;  // (p - b).bc
  const float t3 = gq_sat(gbc * f.r2.w);
  const float ddb = fmaf(-2.0f, d1, dd) + AA;  // |p - b|^2
  const float e3 = fmaf(-t3, fmaf(-t3, BC, gbc) + gbc, ddb);
  return (m >= 0.0f) ? dp : fminf(fminf(e1, e2), e3);  // m is NaN for degenerate faces -> edges
}

// squared distance from d = p - a to the triangle described by `f`
__device__ __forceinline__ float gq_tri_dist2(const GqFace& f, gq3 d) {
  const gq3 ab = gq_mk(f.r1.x, f.r1.y, f.r1.z), ac = gq_mk(f.r2.x, f.r2.y, f.r2.z);
  const float d1 = gq_dot(ab, d), d2 = gq_dot(ac, d);
  const float v = f.r3.x * d1 - f.r3.y * d2;
  const float w = f.r3.z * d2 - f.r3.y * d1;
  const float u = 1.0f - v - w;
  const bool inside = fminf(fminf(v, w), u) >= 0.0f;  // false for NaN (degenerate face)
  const gq3 rp = d - v * ab - w * ac;
  const float dp = gq_dot(rp, rp);
  const float t1 = gq_sat(d1 * f.r0.w);
  const gq3 r1 = d - t1 * ab;
  const float e1 = gq_dot(r1, r1);
  const float t2 = gq_sat(d2 * f.r1.w);
  const gq3 r2 = d - t2 * ac;
  const float e2 = gq_dot(r2, r2);
  const float t3 = gq_sat((d2 - d1 + f.r3.w) * f.r2.w);
  const gq3 r3 = d - (1.0f - t3) * ab - t3 * ac;
  const float e3 = gq_dot(r3, r3);
  return inside ? dp : fminf(fminf(e1, e2), e3);
}

// closest point on the triangle (offset from a) + which feature won; used once per query for the best face
__device__ __forceinline__ gq3 gq_tri_closest_off(const GqFace& f, gq3 d, float& dist2) {
  const gq3 ab = gq_mk(f.r1.x, f.r1.y, f.r1.z), ac = gq_mk(f.r2.x, f.r2.y, f.r2.z);
  const float d1 = gq_dot(ab, d), d2 = gq_dot(ac, d);
  const float v = f.r3.x * d1 - f.r3.y * d2;
  const float w = f.r3.z * d2 - f.r3.y * d1;
  const float u = 1.0f - v - w;
  const bool inside = fminf(fminf(v, w), u) >= 0.0f;
  gq3 c = v * ab + w * ac;
  gq3 r = d - c;
  float best = gq_dot(r, r);
  if (!inside) {
    const float t1 = gq_sat(d1 * f.r0.w);
    gq3 c1 = t1 * ab;
    gq3 r1 = d - c1;
    best = gq_dot(r1, r1);
    c = c1;
    const float t2 = gq_sat(d2 * f.r1.w);
    gq3 c2 = t2 * ac;
    gq3 r2 = d - c2;
    const float e2 = gq_dot(r2, r2);
    if (e2 < best) {
      best = e2;
      c = c2;
    }
    const float t3 = gq_sat((d2 - d1 + f.r3.w) * f.r2.w);
    gq3 c3 = (1.0f - t3) * ab + t3 * ac;
    gq3 r3 = d - c3;
    const float e3 = gq_dot(r3, r3);
    if (e3 < best) {
      best = e3;
      c = c3;
    }
  }
  dist2 = best;
  return c;
}

// full TorchSDF-contract result for the winning face
struct GqSdfOut {
  float dist2;
  int sign;
  gq3 normal, closest;
};
__device__ __forceinline__ GqSdfOut gq_tri_finish(const GqFace& f, gq3 p) {
  const gq3 a = gq_mk(f.r0.x, f.r0.y, f.r0.z);
  const gq3 ab = gq_mk(f.r1.x, f.r1.y, f.r1.z), ac = gq_mk(f.r2.x, f.r2.y, f.r2.z);
  const gq3 d = p - a;
  GqSdfOut o;
  const gq3 off = gq_tri_closest_off(f, d, o.dist2);
  o.closest = a + off;
  const gq3 diff = d - off;
  const gq3 fn = gq_cross(ab, ac);
  o.sign = (gq_dot(diff, fn) >= 0.0f) ? 1 : -1;
  if (o.dist2 > 0.0f) {
    const float inv = 1.0f / sqrtf(o.dist2);
    o.normal = inv * diff;
  } else {
    const float nn = gq_dot(fn, fn);
    o.normal = (1.0f / sqrtf(fmaxf(nn, 1e-30f))) * fn;
  }
  return o;
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
  const gq3 ab = b - a, ac = c - a;
  // setup-time only: the per-face constants are formed in double so that slivers do not lose nn to cancellation
  const double abx = ab.x, aby = ab.y, abz = ab.z, acx = ac.x, acy = ac.y, acz = ac.z;
  const double AA = abx * abx + aby * aby + abz * abz, CC = acx * acx + acy * acy + acz * acz;
  const double AB = abx * acx + aby * acy + abz * acz;
  const double BC = AA - 2.0 * AB + CC;
  const double nn = AA * CC - AB * AB;
  const float nanv = __builtin_nanf("");
  GqFace f;
  f.r0 = make_float4(a.x, a.y, a.z, AA > 0.0 ? (float)(1.0 / AA) : 0.0f);
  f.r1 = make_float4(ab.x, ab.y, ab.z, CC > 0.0 ? (float)(1.0 / CC) : 0.0f);
  f.r2 = make_float4(ac.x, ac.y, ac.z, BC > 0.0 ? (float)(1.0 / BC) : 0.0f);
  const bool ok = nn > 1e-30 && nn > 1e-12 * AA * CC;
  f.r3 = make_float4(ok ? (float)(CC / nn) : nanv, ok ? (float)(AB / nn) : nanv, ok ? (float)(AA / nn) : nanv,
                     (float)(AA - AB));
  const double nx = aby * acz - abz * acy, ny = abz * acx - abx * acz, nz = abx * acy - aby * acx;
  const double nl = sqrt(nx * nx + ny * ny + nz * nz);
  const double inl = nl > 0.0 ? 1.0 / nl : 0.0;
  f.r4 = make_float4((float)(nx * inl), (float)(ny * inl), (float)(nz * inl), (float)AA);
  f.r5 = make_float4((float)CC, (float)BC, __int_as_float(orig), 0.0f);
  return f;
}
