// TorchSDF-shaped mesh distance query for MANY points against one mesh: one query per lane, exact closest face through an
// implicit 4-ary bounding-volume hierarchy (reference call sites: core/hand_model.py:914-953 -- batch * 2500 object surface
// points against every hand-link mesh, N = 640 000 at BASELINE configs[1] -- and core/object_model.py:217-220 for large
// batches).  Replaces the per-lane loop over ALL faces (gq_sdf_points_kernel: F x 36 VALU operations per query).
//
// Layout (one contiguous blob of 16-byte words, built once per mesh on the host, staged into LDS by every block when it fits):
//   faces are Morton-sorted; level 0 groups 4 consecutive faces ("leaf"), level k groups 4 nodes of level k-1, until at most
//   4 nodes are left (DEPTH levels).  No child pointers: the children of node i of level k are nodes 4i .. 4i+3 of level k-1.
//   node box = 2 words (lo.xyz hi.x | hi.yz - -); every level is padded to a multiple of 4 with empty boxes (lo = +inf);
//   per face: its own box (2 words) and its record (GqFace, 6 words; tri.h).
// Traversal per lane: at every node the 4 child boxes give 4 lower bounds; children are visited nearest first while their
// bound does not exceed the best distance found so far; in a leaf the face boxes filter the 4 faces before the exact
// ranking distance (gq_tri_rank) is evaluated.  Ties go to the smallest ORIGINAL face index (the face loop's rule).  The
// winner is finished exactly (gq_tri_finish), like in every other SDF kernel here.
// Wavefront cost = the slowest lane's path; measured in bench.py's plugin_surface block.
// LDS residency is what makes the per-lane gathers cheap: blobs up to 156 KB (~1 120 faces) are staged (one 1024-thread block
// per CU above 80 KB), larger ones are walked from L2 at about twice the cost per query.  On the reference's per-link call
// shape (far-field queries) the hierarchy beats the face loop and the cluster search at every mesh size either way; for many
// CONTACT-like queries (within centimetres of a 9 k-face surface) the cluster search is the faster one
// (tools/ab_contact_routes.py, profiles/r03_ab_sdf_routes.txt) -- the reference never issues such a call with >= 32 768 queries.
#include "tri.h"

#include <algorithm>
#include <vector>

struct gqBvh {
  float4* blob;      // device
  size_t words;      // 16-byte words in the blob
  int F, Fp, depth;  // faces, padded faces (4 * n[0]), levels
  int lvl_off[8];    // word offset of level k's boxes
  int lvl_n[8];      // padded node count of level k
  int fbox_off, rec_off;
  float centre[3];   // centre of the mesh's bounding box (direction bins of the sorted kernel)
};

struct GqBvhArgs {
  const float4* blob;
  unsigned words;
  int lvl_off[8];
  int top_n;  // padded node count of the top level (4)
  int fbox_off, rec_off;
  const float* points;
  long long N;
  float* dist_sq;
  int32_t* sign;
  float* normal;
  float* closest;
};

__device__ __forceinline__ float gq_bvh_box_lb(const float4 b0, const float4 b1, gq3 p) {
  const float ex = fmaxf(fmaxf(b0.x - p.x, p.x - b0.w), 0.0f);
  const float ey = fmaxf(fmaxf(b0.y - p.y, p.y - b1.x), 0.0f);
  const float ez = fmaxf(fmaxf(b0.z - p.z, p.z - b1.y), 0.0f);
  return fmaf(ex, ex, fmaf(ey, ey, ez * ez));
}
// a bound may exceed the best RANKING distance by the ranking noise (~1e-10 m^2 absolute, tri.h) without being farther
__device__ __forceinline__ float gq_bvh_thr(float best) { return fmaf(best, 1.0f + 1e-6f, 1e-11f); }

struct GqBvhBest {
  float d2;
  int idx, orig;
};

template <int LVL>
struct GqBvhVisit {
  // node `i` of level LVL; its children are 4i .. 4i+3 of level LVL-1
  static __device__ __forceinline__ void run(const float4* __restrict__ base, const GqBvhArgs& g, gq3 p, int i, GqBvhBest& b) {
    const float4* cb = base + g.lvl_off[LVL - 1] + 8 * i;
    float lb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lb[k] = gq_bvh_box_lb(cb[2 * k], cb[2 * k + 1], p);
#pragma unroll 1  // ONE inlined copy of the next level per level (an unrolled loop would make 4^DEPTH of them)
    for (int r = 0; r < 4; ++r) {  // nearest pending child first
      const float m = fminf(fminf(lb[0], lb[1]), fminf(lb[2], lb[3]));
      if (!(m <= gq_bvh_thr(b.d2)) || m == GQ_INF_F) break;  // (empty padding boxes have an infinite bound)
      const int k = (m == lb[0]) ? 0 : (m == lb[1]) ? 1 : (m == lb[2]) ? 2 : 3;
#pragma unroll
      for (int q = 0; q < 4; ++q) lb[q] = (q == k) ? GQ_INF_F : lb[q];
      GqBvhVisit<LVL - 1>::run(base, g, p, 4 * i + k, b);
    }
  }
};
template <>
struct GqBvhVisit<0> {
  // leaf i: faces 4i .. 4i+3
  static __device__ __forceinline__ void run(const float4* __restrict__ base, const GqBvhArgs& g, gq3 p, int i, GqBvhBest& b) {
    const float4* fb = base + g.fbox_off + 8 * i;
    float lb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lb[k] = gq_bvh_box_lb(fb[2 * k], fb[2 * k + 1], p);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (lb[k] <= gq_bvh_thr(b.d2) && lb[k] < GQ_INF_F) {
        const int f = 4 * i + k;
        const GqFace fc = *reinterpret_cast<const GqFace*>(base + g.rec_off + 6 * f);
        const float d2 = gq_tri_rank(fc, p - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z));
        const int orig = __float_as_int(fc.r5.z);
        if (d2 < b.d2 || (d2 == b.d2 && orig < b.orig)) {
          b.d2 = d2;
          b.idx = f;
          b.orig = orig;
        }
      }
    }
  }
};

// persistent blocks: the blob is staged once per block (LDS = true), then the block strides over chunks of blockDim.x points.
// Blocks have 512 threads while two or more of them fit a CU's 160 KB of LDS beside their blobs, 1024 threads (one block per
// CU, the same 16 wavefronts) for the larger blobs up to GQ_BVH_LDS_MAX.
#ifndef GQ_BVH_MIN_BLOCKS
#define GQ_BVH_MIN_BLOCKS 1  // min wavefronts per SIMD; A/B: 6 caps the kernel at 80 VGPRs (three 512-thread blocks per CU)
#endif
#define GQ_BVH_LDS_MAX (156 * 1024)
template <int DEPTH, bool LDS>
__global__ __launch_bounds__(1024, GQ_BVH_MIN_BLOCKS) void gq_sdf_bvh_kernel(GqBvhArgs g) {
  extern __shared__ float4 gq_bvh_sh[];
  const float4* base = g.blob;
  const unsigned nthr = blockDim.x;
  if (LDS) {
    for (unsigned i = threadIdx.x; i < g.words; i += nthr) gq_bvh_sh[i] = g.blob[i];
    __syncthreads();
    base = gq_bvh_sh;
  }
  const long long nchunk = (g.N + nthr - 1) / nthr;
  for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    const long long q = ch * nthr + threadIdx.x;
    const bool ok = q < g.N;
    const long long qq = ok ? q : g.N - 1;
    const gq3 p = gq_mk(g.points[qq * 3 + 0], g.points[qq * 3 + 1], g.points[qq * 3 + 2]);
    GqBvhBest b{GQ_INF_F, 0, 0x7fffffff};
    // virtual root: its children are the (<= 4, padded) nodes of the top level
    const float4* cb = base + g.lvl_off[DEPTH - 1];
    float lb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lb[k] = gq_bvh_box_lb(cb[2 * k], cb[2 * k + 1], p);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      const float m = fminf(fminf(lb[0], lb[1]), fminf(lb[2], lb[3]));
      if (!(m <= gq_bvh_thr(b.d2)) || m == GQ_INF_F) break;
      const int k = (m == lb[0]) ? 0 : (m == lb[1]) ? 1 : (m == lb[2]) ? 2 : 3;
#pragma unroll
      for (int qk = 0; qk < 4; ++qk) lb[qk] = (qk == k) ? GQ_INF_F : lb[qk];
      GqBvhVisit<DEPTH - 1>::run(base, g, p, k, b);
    }
    if (!ok) continue;
    const GqFace fc = *reinterpret_cast<const GqFace*>(base + g.rec_off + 6 * b.idx);
    const GqSdfOut o = gq_tri_finish(fc, p);
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

// The same traversal with the queries of a 2048-point chunk first ORDERED BY DIRECTION from the mesh centre inside the block
// (LDS counting sort over 96 direction bins: cube face x 4 x 4 cells): the 64 lanes of a wavefront then walk nearly the same
// nodes, so their LDS reads coalesce into broadcasts instead of conflicting and the wavefront's cost (its slowest lane)
// approaches the mean.  The order in which equal-bin queries land on lanes depends on LDS atomics, the result of every
// query does not.  Results are written back to the query's own slot.  OFF by default: see gq_bvh_sorted_ below.
#define GQ_BVH_CHUNK 2048
#define GQ_BVH_BINS 96
__device__ __forceinline__ int gq_bvh_dir_bin(gq3 d) {
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  int face;
  float u, v, m;
  if (ax >= ay && ax >= az) { face = d.x < 0.0f ? 1 : 0; m = ax; u = d.y; v = d.z; }
  else if (ay >= az) { face = d.y < 0.0f ? 3 : 2; m = ay; u = d.x; v = d.z; }
  else { face = d.z < 0.0f ? 5 : 4; m = az; u = d.x; v = d.y; }
  const float inv = m > 0.0f ? 2.0f / m : 0.0f;  // u / m in [-1, 1] -> cell 0..3
  int iu = (int)fminf(fmaxf(fmaf(u, inv, 2.0f), 0.0f), 3.0f);
  int iv = (int)fminf(fmaxf(fmaf(v, inv, 2.0f), 0.0f), 3.0f);
  if (iv & 1) iu = 3 - iu;  // boustrophedon inside the face: consecutive bins are neighbours
  return face * 16 + iv * 4 + iu;
}

template <int DEPTH, bool LDS>
__global__ __launch_bounds__(512) void gq_sdf_bvh_sorted_kernel(GqBvhArgs g, float cx, float cy, float cz) {
  extern __shared__ float4 gq_bvh_sh[];
  __shared__ unsigned s_hist[GQ_BVH_BINS + 32];
  __shared__ unsigned short s_perm[GQ_BVH_CHUNK];
  const float4* base = g.blob;
  if (LDS) {
    for (unsigned i = threadIdx.x; i < g.words; i += 512) gq_bvh_sh[i] = g.blob[i];
    base = gq_bvh_sh;
  }
  const int tid = (int)threadIdx.x;
  const long long nchunk = (g.N + GQ_BVH_CHUNK - 1) / GQ_BVH_CHUNK;
  for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    const long long q0 = ch * GQ_BVH_CHUNK;
    if (tid < GQ_BVH_BINS + 32) s_hist[tid] = 0u;
    __syncthreads();
    int bin[4];
    unsigned pos[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long q = q0 + r * 512 + tid;
      bin[r] = -1;
      if (q < g.N) {
        const gq3 d = gq_mk(g.points[q * 3 + 0] - cx, g.points[q * 3 + 1] - cy, g.points[q * 3 + 2] - cz);
        bin[r] = gq_bvh_dir_bin(d);
        pos[r] = atomicAdd(&s_hist[bin[r]], 1u);
      }
    }
    __syncthreads();
    if (tid < GQ_WAVE) {  // exclusive scan of the 96 counters by one wavefront (two per lane)
      const unsigned a = s_hist[2 * tid], b = (2 * tid + 1 < GQ_BVH_BINS + 32) ? s_hist[2 * tid + 1] : 0u;
      unsigned incl = a + b;
#pragma unroll
      for (int o = 1; o < GQ_WAVE; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, GQ_WAVE);
        if (tid >= o) incl += t;
      }
      const unsigned excl = incl - (a + b);
      s_hist[2 * tid] = excl;
      if (2 * tid + 1 < GQ_BVH_BINS + 32) s_hist[2 * tid + 1] = excl + a;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (bin[r] >= 0) s_perm[s_hist[bin[r]] + pos[r]] = (unsigned short)(r * 512 + tid);
    __syncthreads();
    const int n_here = (int)((g.N - q0) < GQ_BVH_CHUNK ? (g.N - q0) : GQ_BVH_CHUNK);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      const int slot = r * 512 + tid;
      const bool ok = slot < n_here;
      const long long q = q0 + (ok ? (int)s_perm[slot] : 0);
      const gq3 p = gq_mk(g.points[q * 3 + 0], g.points[q * 3 + 1], g.points[q * 3 + 2]);
      GqBvhBest b{GQ_INF_F, 0, 0x7fffffff};
      const float4* cb = base + g.lvl_off[DEPTH - 1];
      float lb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) lb[k] = gq_bvh_box_lb(cb[2 * k], cb[2 * k + 1], p);
#pragma unroll 1
      for (int rr = 0; rr < 4; ++rr) {
        const float m = fminf(fminf(lb[0], lb[1]), fminf(lb[2], lb[3]));
        if (!(m <= gq_bvh_thr(b.d2)) || m == GQ_INF_F) break;
        const int k = (m == lb[0]) ? 0 : (m == lb[1]) ? 1 : (m == lb[2]) ? 2 : 3;
#pragma unroll
        for (int qk = 0; qk < 4; ++qk) lb[qk] = (qk == k) ? GQ_INF_F : lb[qk];
        GqBvhVisit<DEPTH - 1>::run(base, g, p, k, b);
      }
      if (!ok) continue;
      const GqFace fc = *reinterpret_cast<const GqFace*>(base + g.rec_off + 6 * b.idx);
      const GqSdfOut o = gq_tri_finish(fc, p);
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
    __syncthreads();  // s_perm / s_hist are reused by the next chunk
  }
}

static inline uint32_t gq_bvh_spread10(uint32_t v) {
  v &= 0x3ff;
  v = (v | (v << 16)) & 0x030000ff;
  v = (v | (v << 8)) & 0x0300f00f;
  v = (v | (v << 4)) & 0x030c30c3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}

__global__ void gq_bvh_rec_kernel(const float* __restrict__ fv, const int32_t* __restrict__ perm, int F, int Fp, float4* __restrict__ rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Fp) return;
  GqFace f;
  if (i < F) {
    const float* v = fv + (size_t)perm[i] * 9;
    f = gq_make_face(gq_mk(v[0], v[1], v[2]), gq_mk(v[3], v[4], v[5]), gq_mk(v[6], v[7], v[8]), perm[i]);
  } else {  // padding: never passes its (empty) face box; a far-away point in any case
    const gq3 far = gq_mk(1e18f, 1e18f, 1e18f);
    f = gq_make_face(far, far, far, 0x7fffffff);
  }
  float4* o = rec + (size_t)6 * i;
  o[0] = f.r0; o[1] = f.r1; o[2] = f.r2; o[3] = f.r3; o[4] = f.r4; o[5] = f.r5;
}

// gq_debug_set_bvh_sorted: 1 = order every 2048-query chunk by direction first (A/B runs).  Measured SLOWER on the reference's
// call shape (14 Allegro links x 640 000 queries: 2.42 ms against 1.60 ms in plain order, profiles/r03_plugin_surface_*.json):
// the counting sort, its barriers and the scattered point / result accesses cost more than the coherence returns.
static int gq_bvh_sorted_ = 0;

extern "C" {

int gq_debug_set_bvh_sorted(int on) {
  gq_bvh_sorted_ = on;
  return GQ_OK;
}

int gq_bvh_create(const float* face_verts_host, int64_t n_faces, gqBvh** out) {
  GQ_REQUIRE(face_verts_host && out && n_faces > 0 && n_faces <= 65536, "bvh_create: 1..65536 faces, got %lld", (long long)n_faces);
  const int F = (int)n_faces;
  // Morton order of the face centroids
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  for (int64_t i = 0; i < (int64_t)F * 9; ++i) {
    const int c = (int)(i % 3);
    lo[c] = std::min(lo[c], face_verts_host[i]);
    hi[c] = std::max(hi[c], face_verts_host[i]);
  }
  std::vector<std::pair<uint32_t, int32_t>> keys(F);
  for (int i = 0; i < F; ++i) {
    const float* v = face_verts_host + (size_t)i * 9;
    uint32_t code = 0;
    for (int c = 0; c < 3; ++c) {
      const float ctr = (v[c] + v[3 + c] + v[6 + c]) * (1.0f / 3.0f), ext = hi[c] - lo[c];
      float t = ext > 0.0f ? (ctr - lo[c]) / ext : 0.0f;
      t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
      code |= gq_bvh_spread10((uint32_t)(t * 1023.0f)) << c;
    }
    keys[i] = {code, i};
  }
  std::stable_sort(keys.begin(), keys.end());
  std::vector<int32_t> perm(F);
  for (int i = 0; i < F; ++i) perm[i] = keys[i].second;
  gqBvh* b = new gqBvh();
  b->F = F;
  for (int c = 0; c < 3; ++c) b->centre[c] = 0.5f * (lo[c] + hi[c]);
  // levels
  int n = (F + 3) / 4, depth = 0;
  std::vector<int> ln;
  for (;;) {
    ln.push_back((n + 3) / 4 * 4);  // padded
    ++depth;
    if (n <= 4) break;
    n = (n + 3) / 4;
  }
  GQ_REQUIRE(depth <= 8, "bvh_create: too deep");
  b->depth = depth;
  b->Fp = 4 * ln[0];
  // (padding nodes have empty boxes and are never entered, so their children need not exist: level k-1 holds exactly
  // 4 x the real nodes of level k, rounded up to a multiple of 4)
  size_t w = 0;
  for (int k = depth - 1; k >= 0; --k) {  // top level first (read by every query)
    b->lvl_off[k] = (int)w;
    b->lvl_n[k] = ln[k];
    w += (size_t)2 * ln[k];
  }
  b->fbox_off = (int)w;
  w += (size_t)2 * b->Fp;
  b->rec_off = (int)w;
  w += (size_t)6 * b->Fp;
  b->words = w;
  std::vector<float> host(w * 4, 0.0f);
  const float inf = __builtin_inff();
  auto put_box = [&](size_t word, const float* blo, const float* bhi) {
    float* o = &host[word * 4];
    o[0] = blo[0]; o[1] = blo[1]; o[2] = blo[2]; o[3] = bhi[0];
    o[4] = bhi[1]; o[5] = bhi[2]; o[6] = 0.0f; o[7] = 0.0f;
  };
  const float elo[3] = {inf, inf, inf}, ehi[3] = {-inf, -inf, -inf};
  // face boxes
  std::vector<float> blo((size_t)b->Fp * 3, inf), bhi((size_t)b->Fp * 3, -inf);
  for (int i = 0; i < F; ++i) {
    const float* v = face_verts_host + (size_t)perm[i] * 9;
    for (int k = 0; k < 9; ++k) {
      const int c = k % 3;
      blo[(size_t)i * 3 + c] = std::min(blo[(size_t)i * 3 + c], v[k]);
      bhi[(size_t)i * 3 + c] = std::max(bhi[(size_t)i * 3 + c], v[k]);
    }
  }
  for (int i = 0; i < b->Fp; ++i) put_box((size_t)b->fbox_off + 2 * i, i < F ? &blo[(size_t)i * 3] : elo, i < F ? &bhi[(size_t)i * 3] : ehi);
  // node levels bottom-up
  std::vector<float> clo = blo, chi = bhi;
  int cn = b->Fp;
  for (int k = 0; k < depth; ++k) {
    std::vector<float> nlo((size_t)ln[k] * 3, inf), nhi((size_t)ln[k] * 3, -inf);
    for (int i = 0; i < ln[k]; ++i)
      for (int q = 0; q < 4; ++q) {
        const int ch = 4 * i + q;
        if (ch >= cn) continue;
        for (int c = 0; c < 3; ++c) {
          nlo[(size_t)i * 3 + c] = std::min(nlo[(size_t)i * 3 + c], clo[(size_t)ch * 3 + c]);
          nhi[(size_t)i * 3 + c] = std::max(nhi[(size_t)i * 3 + c], chi[(size_t)ch * 3 + c]);
        }
      }
    for (int i = 0; i < ln[k]; ++i) put_box((size_t)b->lvl_off[k] + 2 * i, &nlo[(size_t)i * 3], &nhi[(size_t)i * 3]);
    clo.swap(nlo);
    chi.swap(nhi);
    cn = ln[k];
  }
  float* fv_dev = nullptr;
  int32_t* perm_dev = nullptr;
  GQ_CHECK_HIP(hipMalloc(&b->blob, w * 16));
  GQ_CHECK_HIP(hipMalloc(&fv_dev, (size_t)F * 9 * 4));
  GQ_CHECK_HIP(hipMalloc(&perm_dev, (size_t)F * 4));
  GQ_CHECK_HIP(hipMemcpy(b->blob, host.data(), w * 16, hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(fv_dev, face_verts_host, (size_t)F * 9 * 4, hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(perm_dev, perm.data(), (size_t)F * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gq_bvh_rec_kernel, dim3((unsigned)((b->Fp + 255) / 256)), dim3(256), 0, 0, fv_dev, perm_dev, F, b->Fp,
                     b->blob + b->rec_off);
  GQ_LAUNCH_CHECK();
  GQ_CHECK_HIP(hipDeviceSynchronize());
  GQ_CHECK_HIP(hipFree(fv_dev));
  GQ_CHECK_HIP(hipFree(perm_dev));
  *out = b;
  return GQ_OK;
}

int gq_bvh_destroy(gqBvh* b) {
  if (!b) return GQ_OK;
  (void)hipFree(b->blob);
  delete b;
  return GQ_OK;
}

int gq_sdf_forward_bvh(const gqBvh* b, const float* points, int64_t n_points, float* dist_sq, int32_t* sign, float* normal,
                       float* closest, void* stream) {
  if (n_points == 0) return GQ_OK;
  GQ_REQUIRE(b && points && dist_sq && sign && closest && n_points > 0, "sdf_forward_bvh: bad arguments");
  GqBvhArgs a{};
  a.blob = b->blob;
  a.words = (unsigned)b->words;
  for (int k = 0; k < 8; ++k) a.lvl_off[k] = b->lvl_off[k];
  a.fbox_off = b->fbox_off;
  a.rec_off = b->rec_off;
  a.points = points;
  a.N = n_points;
  a.dist_sq = dist_sq;
  a.sign = sign;
  a.normal = normal;
  a.closest = closest;
  const size_t bytes = b->words * 16;
  const bool sorted = gq_bvh_sorted_ != 0 && n_points >= 4 * GQ_BVH_CHUNK;
  // LDS-resident while the blob fits: two or more 512-thread blocks per CU up to 80 KB, one 1024-thread block per CU above
  // (a hierarchy walked from global memory costs about twice as much per query: 140 us at 456 faces in LDS against 256 us at
  // 502 faces from L2 on the reference's per-link call shape, profiles/r03_ab_sdf_routes.txt); the A/B sorted variant keeps
  // its 64 KB limit
  const bool lds = b->depth <= 5 && bytes <= (sorted ? (size_t)64 * 1024 : (size_t)GQ_BVH_LDS_MAX);
#ifndef GQ_BVH_SMALL_THREADS
#define GQ_BVH_SMALL_THREADS 512u  // A/B: 640 (two blocks = five wavefronts per SIMD for the <= 3-level hierarchies) is 20 % SLOWER
#endif                             // on the 188 ... 232-face Allegro links (profiles/r03_ab_bvh_occupancy.txt): LDS-bound, not latency-bound
  const unsigned threads = (!sorted && lds && bytes > 80 * 1024) ? 1024u : ((!sorted && lds && b->depth <= 3) ? GQ_BVH_SMALL_THREADS : 512u);
  const long long per_chunk = sorted ? GQ_BVH_CHUNK : (long long)threads;
  const long long nchunk = (n_points + per_chunk - 1) / per_chunk;
  const size_t stat = sorted ? 5 * 1024 : 0;  // static LDS of the sorted kernel (histogram + permutation)
  const int per_cu = lds ? (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / (bytes + stat))) : 4;
  const unsigned grid = (unsigned)std::min<long long>(nchunk, 256ll * per_cu);
  hipStream_t st = (hipStream_t)stream;
#define GQ_BVH_LAUNCH(D, L)                                                                                                   \
  do {                                                                                                                        \
    if (sorted) {                                                                                                             \
      hipLaunchKernelGGL((gq_sdf_bvh_sorted_kernel<D, L>), dim3(grid), dim3(512), (L) ? bytes : 0, st, a, b->centre[0],       \
                         b->centre[1], b->centre[2]);                                                                         \
    } else {                                                                                                                  \
      if ((L) && bytes > 64 * 1024) { /* more dynamic LDS than the 64 KB a kernel may use without asking */                    \
        static bool raised = false;                                                                                           \
        if (!raised) {                                                                                                        \
          GQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gq_sdf_bvh_kernel<D, L>),                           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GQ_BVH_LDS_MAX));                      \
          raised = true;                                                                                                      \
        }                                                                                                                     \
      }                                                                                                                       \
      hipLaunchKernelGGL((gq_sdf_bvh_kernel<D, L>), dim3(grid), dim3(threads), (L) ? bytes : 0, st, a);                       \
    }                                                                                                                         \
  } while (0)
  switch (b->depth) {
    case 1: if (lds) GQ_BVH_LAUNCH(1, true); else GQ_BVH_LAUNCH(1, false); break;
    case 2: if (lds) GQ_BVH_LAUNCH(2, true); else GQ_BVH_LAUNCH(2, false); break;
    case 3: if (lds) GQ_BVH_LAUNCH(3, true); else GQ_BVH_LAUNCH(3, false); break;
    case 4: if (lds) GQ_BVH_LAUNCH(4, true); else GQ_BVH_LAUNCH(4, false); break;
    case 5: if (lds) GQ_BVH_LAUNCH(5, true); else GQ_BVH_LAUNCH(5, false); break;
    case 6: GQ_BVH_LAUNCH(6, false); break;
    case 7: GQ_BVH_LAUNCH(7, false); break;
    default: GQ_BVH_LAUNCH(8, false); break;
  }
#undef GQ_BVH_LAUNCH
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
