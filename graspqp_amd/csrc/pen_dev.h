// Device bodies of the E_pen hot path (penetration-only hand query and its backward), shared by the stand-alone
// kernels of sdf.hip and the fused stage kernels of stage.hip.  A body takes its block coordinates and a pointer to
// the block's dynamic LDS instead of reading blockIdx / declaring static __shared__ arrays, so that several bodies
// can live in one kernel and share one LDS allocation.
#pragma once
#include "tri.h"
#include "wave.h"

// ---- mesh-set handle: concatenated face records of n_mesh meshes on the device -----------------------------------
struct gqMeshSet {
  GqFace* rec;         // records, faces Morton-sorted inside each mesh
  int32_t* off_dev;    // (n_mesh+1) face offsets
  int32_t* off_host;
  float* aabb_dev;     // (n_mesh, 8) box of each mesh in its own frame
  float* sub_aabb_dev; // (n_sub, 8) boxes of 16-face sub-clusters
  int32_t* sub_off_dev;   // (n_mesh+1)
  float* cl_aabb_dev;  // (n_cl, 16) oriented boxes of the 64-face clusters (gq_cluster_bound)
  int32_t* cl_off_dev;    // (n_mesh+1)
  uint32_t* occ_dev;   // (n_mesh, 1024) occupancy bits or null (gq_meshset_build_occupancy)
  float* occ_invz_dev; // (n_mesh)
  uint32_t* cand_off_dev;  // (n_mesh*32768 + 1) per-voxel candidate lists or null (gq_meshset_build_occupancy)
  uint16_t* cand_idx_dev;
  int64_t n_cand;
  int n_mesh;
  int64_t n_faces;
};


// ---- uniform grid over the surface points of every object (set-up data of the cell-driven penetration query) --------
struct gqPointGrid {
  float* box_dev;        // (n_obj,8): lo.xyz, -, cells per metre x y z, -
  int32_t* start_dev;    // (n_obj, G^3 + 1)
  uint16_t* pts_dev;     // (n_obj, P)
  int n_obj, P, G;
};

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
  const float* occ_invz;   // (L) 32 / z-extent of the link AABB (x, y scales ride in the pads of aabb)
  const uint32_t* occ;     // (L, 32*32) words: bit ix of word iz*32+iy set <=> voxel may contain interior/surface
  const uint32_t* cand_off;  // (L*32768 + 1) candidate-list offsets per voxel, or null (gq_cand_fill_kernel)
  const uint16_t* cand_idx;  // face indices (local to the link mesh) that can be closest to some point of the voxel
  int B, P, L, D, batch_each;
  float* dis;     // (B, P)
  int32_t* link;  // (B, P)
  float* gvec;    // (B, P, 3)
  uint64_t* span;  // optional [min start, max end] of the launch in 100 MHz s_memrealtime ticks
  // optional uniform grid over every object's surface points (gqPointGrid): the cell-driven query gq_pen_cells_body
  const float* grid_box;       // (n_obj,8): lo.xyz, -, cells per metre x y z, -
  const int32_t* grid_start;   // (n_obj, G^3 + 1) prefix offsets into grid_pts
  const uint16_t* grid_pts;    // (n_obj, P) point indices grouped by cell
  int G;
  // optional (n_obj, ceil(P/256), 4): bounding sphere (centre, radius; object frame) of every 256-point slice of the
  // surface points -- lets a block drop, before anything else, the links whose box cannot reach its slice
  const float* patch;
  unsigned long long* dbg;  // optional counters (12 words, gq_debug_set_pen_counters): [0] needing (point,link) pairs,
                            // [1] (wave,link) evaluations, [2] (wave,sub-cluster) evaluations, [3] waves (AABB / queue
                            // kernels); gq_pen_grid_body: [4] entries = (point,link) pairs that reach a non-empty
                            // voxel, [5] executed point-triangle rankings, [6] entries ranked inline (LDS capacity
                            // overflow), [7] blocks
};

// launch time span in 100 MHz s_memrealtime ticks, sharded 64 ways so the atomics of 1e3 blocks do not pile up on one
// address: span[2*s] = min start, span[2*s+1] = max end of the blocks with (linear block id % 64) == s
// -DGQ_BLOCK_TIMES (development builds only, tools/block_timeline.py): the span buffer is followed by eight words per
// block -- start, end, end of scan / ranking / finish, entries | items << 32 -- so the caller passes 128 + 8 * blocks words
#ifdef GQ_BLOCK_TIMES
__device__ __forceinline__ uint64_t gq_hw_id() {  // HW_ID (wave / SIMD / CU / SH / SE of gfx9) | XCC_ID << 32
  return (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
}
#endif
__device__ __forceinline__ void gq_span_open(uint64_t* span, unsigned block_id) {
  const unsigned s = block_id & 63u;
  const unsigned long long t = (unsigned long long)__builtin_amdgcn_s_memrealtime();
  atomicMin((unsigned long long*)&span[2 * s], t);
#ifdef GQ_BLOCK_TIMES
  span[128 + 8 * (size_t)block_id] = t;
  span[128 + 8 * (size_t)block_id + 6] = gq_hw_id();
#endif
}
__device__ __forceinline__ void gq_span_close(uint64_t* span, unsigned block_id) {
  const unsigned s = block_id & 63u;
  const unsigned long long t = (unsigned long long)__builtin_amdgcn_s_memrealtime();
  atomicMax((unsigned long long*)&span[2 * s + 1], t);
#ifdef GQ_BLOCK_TIMES
  span[128 + 8 * (size_t)block_id + 1] = t;
#endif
}

// ---- penetration-only query with per-voxel candidate faces (the hot path of E_pen) -------------------------------
// One block = 256 surface points of one row, no global queues, no global atomics.
//   A  every thread walks the links for its point: bounding sphere of the link box (LDS), link frame, AABB, voxel of the
//      32^3 grid -- no global memory.  Only 2e3 .. 3e4 of the 9e6 (point, link) pairs of a config-2 launch survive; a
//      survivor becomes an ENTRY in LDS that remembers its voxel.
//   A2 the block's threads share the entries: candidate list of the entry's voxel (independent look-ups, one round trip
//      for the block); an empty list retires the entry, otherwise its candidate faces become ITEMS (entry, j) in LDS.
//   B  the block's threads share the items evenly: one (point, face) ranking each, all lanes busy, independent gathers
//      in flight; the minimum per entry is taken with a 64-bit LDS atomicMin on (distance, original face index, face),
//      which is independent of the processing order.
//   C  one thread per entry finishes the winner exactly (closest point, sign); penetrating entries compete per point
//      with a 64-bit LDS atomicMax on (dis, 255 - link, entry) -- the max over links with torch's first-index tie rule.
//   D  the point's own thread writes dis (every point) and link / gradient (penetrating points only).
// The candidate list of a voxel holds every face that is closest for SOME point of the voxel (gq_cand_kernel), so the
// result equals the brute-force scan of the whole link mesh.  Entries beyond the LDS capacity are looked up and ranked
// inline by the thread that found them, items beyond it by the thread that looked the entry up (same arithmetic, just
// slower).
#define GQ_PG_ECAP 512
#define GQ_PG_ICAP 4096
struct GqPgEntry {
  float x, y, z;      // point in the link frame
  uint32_t c0;        // first candidate
  uint16_t pt, link;  // local point index, link
};
__device__ __forceinline__ unsigned long long gq_rank_key(float d2, unsigned orig_local, unsigned f_local) {
  unsigned b = __float_as_uint(d2);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // order-preserving map of floats onto unsigned
  return ((unsigned long long)b << 32) | ((unsigned long long)(orig_local & 0xffffu) << 16) | (f_local & 0xffffu);
}
// LDS of one block (bytes): entries, entry keys, items (re-used for the finished entries), point keys, counters, links
__host__ __device__ inline size_t gq_pen_grid_lds_bytes(int L, int ecap = GQ_PG_ECAP, int icap = GQ_PG_ICAP, int ppt = 1) {
  return (size_t)ecap * (sizeof(GqPgEntry) + 8) + (size_t)icap * 4 + (size_t)ppt * 256 * 8 + 32 + (size_t)L * 28 * 4;
}
// block = 256 threads = surface points [256 PPT bx, 256 PPT (bx + 1)) of `row`: PPT points per thread (slice h of the
// block = points 256 (PPT bx + h) + tid).
// ECAP / ICAP: LDS capacities (entries / items, ICAP >= 4 ECAP); what does not fit is ranked inline, so they trade
// occupancy (LDS per block) against the speed of blocks whose whole slice sits inside the hand.
// PPT = 2 halves the blocks and wavefronts of a launch: the prologue (link table, bounding spheres), the per-link loop
// overhead and the three list phases are shared by twice the points, at the same LDS lists.  The results do not depend
// on it (the list phases are order-independent; overflow is ranked inline with the same arithmetic).
template <bool EVAL, int ECAP = GQ_PG_ECAP, int ICAP = GQ_PG_ICAP, int PPT = 1>
__device__ __forceinline__ void gq_pen_grid_body(const GqPenArgs& g, int bx, int row, char* lds) {
  static_assert(ICAP >= 4 * ECAP, "the item area is re-used for 4 floats per entry");
  static_assert(PPT == 1 || PPT == 2, "one or two points per thread");
  unsigned long long* s_ekey = reinterpret_cast<unsigned long long*>(lds);
  unsigned long long* s_pkey = s_ekey + ECAP;
  GqPgEntry* s_ent = reinterpret_cast<GqPgEntry*>(s_pkey + 256 * PPT);
  uint32_t* s_item = reinterpret_cast<uint32_t*>(s_ent + ECAP);  // entry << 16 | j
  float* s_ecl = reinterpret_cast<float*>(s_item);  // after phase B: closest point (link frame) + dis per entry
  int* s_cnt = reinterpret_cast<int*>(s_item + ICAP);  // 4 counters + 2 x the 64-bit mask of the links in reach of a slice
  float* s_link = reinterpret_cast<float*>(s_cnt + 8);  // L x 24: link transform (12) + padded AABB (8) + occupancy
                                                        // z scale (1) + pad, then L x 4: bounding sphere of the link
                                                        // box in the hand frame (centre, r^2)
  float* s_sph = s_link + g.L * 24;
  const int tid = threadIdx.x;
  const int n_slices = (g.P + 255) / 256;
  const unsigned block_id = (unsigned)(bx + row * ((n_slices + PPT - 1) / PPT));
  if (g.span && tid == 0) gq_span_open(g.span, block_id);
  const int obj = row / g.batch_each;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  int pt[PPT];
  bool ok[PPT];
  gq3 xh[PPT];
#pragma unroll
  for (int h = 0; h < PPT; ++h) {
    pt[h] = (bx * PPT + h) * 256 + tid;
    ok[h] = pt[h] < g.P;
    const float* sp = g.surf + ((size_t)obj * g.P + (ok[h] ? pt[h] : 0)) * 3;
    xh[h] = gq_mtv(R, gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]));
  }
  for (int i = tid; i < g.L * 24; i += 256) {  // link transforms / boxes: independent loads, one round trip
    const int l = i / 24, k = i % 24;
    float v = 0.0f;
    if (k < 12) v = g.link_T[((size_t)row * g.L + l) * 12 + k];
    else if (k < 20) v = g.aabb[l * 8 + (k - 12)];
    else if (k == 20) v = g.occ_invz[l];
    s_link[i] = v;
  }
  float4 pslice[PPT];
#pragma unroll
  for (int h = 0; h < PPT; ++h) {
    pslice[h] = make_float4(0, 0, 0, -1.0f);
    const int sl = bx * PPT + h;
    if (g.patch && tid < g.L && sl < n_slices) pslice[h] = *reinterpret_cast<const float4*>(g.patch + ((size_t)obj * n_slices + sl) * 4);
  }
  const bool has_faces = tid < g.L && g.off[tid + 1] > g.off[tid];
  if (tid < 4) s_cnt[tid] = 0;  // [0] entries, [1] items, [2] inline-ranked entries, [3] rankings (diagnostics)
#pragma unroll
  for (int h = 0; h < PPT; ++h) s_pkey[h * 256 + tid] = 0ull;
  __syncthreads();
  if (tid < g.L) {  // bounding sphere of every link box in the hand frame + which links can reach the block's slices at all
    const int l = tid;
    const float* T = s_link + l * 24;
    const float* bb = T + 12;
    const gq3 c = gq_mk(0.5f * (bb[0] + bb[4]), 0.5f * (bb[1] + bb[5]), 0.5f * (bb[2] + bb[6]));
    const gq3 hh = gq_mk(0.5f * (bb[4] - bb[0]), 0.5f * (bb[5] - bb[1]), 0.5f * (bb[6] - bb[2]));
    const gq3 sc = gq_mk(T[0] * c.x + T[1] * c.y + T[2] * c.z + T[3], T[4] * c.x + T[5] * c.y + T[6] * c.z + T[7],
                         T[8] * c.x + T[9] * c.y + T[10] * c.z + T[11]);
    const float r2 = has_faces ? gq_dot(hh, hh) * 1.001f + 1e-12f : -1.0f;
    s_sph[l * 4 + 0] = sc.x;
    s_sph[l * 4 + 1] = sc.y;
    s_sph[l * 4 + 2] = sc.z;
    s_sph[l * 4 + 3] = r2;
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      bool reach = r2 >= 0.0f && (bx * PPT + h) < n_slices;
      if (g.patch && reach) {
        // a point of the slice can only be inside the link's (padded) box if the slice's bounding sphere touches the box:
        // distance of the slice centre to the box, in the link frame (tighter than sphere against sphere for the long,
        // thin finger links)
        const gq3 pc = gq_mtv(R, gq_mk(pslice[h].x - hp[0], pslice[h].y - hp[1], pslice[h].z - hp[2]));  // slice centre, hand frame
        const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
        const gq3 pl = gq_mtv(Rl, pc - gq_mk(T[3], T[7], T[11]));
        reach = gq_aabb_dist2(bb, pl) <= pslice[h].w * pslice[h].w * 1.0002f + 1e-10f;
      }
      const unsigned long long m = __ballot(reach);  // links 0 .. L-1 sit in wavefront 0 (L <= 64)
      if (tid == 0) *reinterpret_cast<unsigned long long*>(s_cnt + 4 + 2 * h) = m;
    }
  }
  __syncthreads();
  unsigned long long lmask[PPT], lany = 0ull;
#pragma unroll
  for (int h = 0; h < PPT; ++h) {
    lmask[h] = *reinterpret_cast<const unsigned long long*>(s_cnt + 4 + 2 * h);
    lany |= lmask[h];
  }
  if (lany == 0ull) {  // no link can touch these slices: nothing penetrates
#pragma unroll
    for (int h = 0; h < PPT; ++h)
      if (ok[h]) g.dis[(size_t)row * g.P + pt[h]] = -1e30f;
    if (g.dbg && tid == 0) atomicAdd(&g.dbg[7], 1ull);
    if (g.span && tid == 0) gq_span_close(g.span, block_id);
    return;
  }
  // ---- A: scan -------------------------------------------------------------------------------------------------
  float in_dis[PPT];  // result of entries this thread had to rank inline (capacity overflow)
  int in_link[PPT];
  gq3 in_cl[PPT], in_xl[PPT];
#pragma unroll
  for (int h = 0; h < PPT; ++h) {
    in_dis[h] = 0.0f;
    in_link[h] = -1;
    in_cl[h] = in_xl[h] = gq_mk(0, 0, 0);
  }
  if (g.dbg && (tid & 63) == 0) {  // [8] (wavefront, link) sphere tests executed, [11] wavefronts that scan
    atomicAdd(&g.dbg[8], (unsigned long long)__builtin_popcountll(lany));
    atomicAdd(&g.dbg[11], 1ull);
  }
  for (unsigned long long rest = lany; rest != 0ull; rest &= rest - 1ull) {  // links in reach of the slices, ascending
    const int l = __builtin_ctzll(rest);
    // bounding sphere first (one LDS read, 7 VALU ops per point); the surface points are Morton-ordered, so a wavefront is
    // a compact patch of the object and most (wavefront, link) pairs end here
    const float4 sph = *reinterpret_cast<const float4*>(s_sph + l * 4);
    bool near[PPT], any_near = false;
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      const gq3 dc = xh[h] - gq_mk(sph.x, sph.y, sph.z);
      near[h] = ok[h] && ((lmask[h] >> l) & 1ull) && gq_dot(dc, dc) <= sph.w;
      any_near |= near[h];
    }
    if (__ballot(any_near) == 0ull) continue;
    if (g.dbg && (tid & 63) == 0) atomicAdd(&g.dbg[9], 1ull);  // (wavefront, link) pairs with a point inside the sphere
    const float* T = s_link + l * 24;
    const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const float* bb = T + 12;
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      if (!near[h]) continue;
      const gq3 xl = gq_mtv(Rl, xh[h] - gq_mk(T[3], T[7], T[11]));
      if (!(gq_aabb_dist2(bb, xl) <= 0.0f)) continue;
      if (g.dbg) atomicAdd(&g.dbg[10], 1ull);  // (point, link) pairs inside the link box
      const float ux = (xl.x - bb[0]) * bb[3], uy = (xl.y - bb[1]) * bb[7], uz = (xl.z - bb[2]) * T[20];
      const int ix = min(max((int)ux, 0), 31), iy = min(max((int)uy, 0), 31), iz = min(max((int)uz, 0), 31);
      if (!EVAL) {
        if (!((g.occ[(size_t)l * 1024 + iz * 32 + iy] >> ix) & 1u)) continue;
        continue;
      }
      // No global memory in this loop: the pair becomes an entry that remembers its voxel; the candidate lists of all
      // entries of the block are looked up afterwards, by all threads at once (A2).  A lane inside k link boxes used to
      // wait for k dependent look-ups here, and with it its whole wavefront (the scan was 4 .. 14 us of a block's 5 .. 18).
      const uint32_t vox = (uint32_t)(iz * 1024 + iy * 32 + ix);
      const int e = atomicAdd(&s_cnt[0], 1);
      if (e < ECAP) {
        GqPgEntry en;
        en.x = xl.x; en.y = xl.y; en.z = xl.z;
        en.c0 = vox;
        en.pt = (uint16_t)(h * 256 + tid);
        en.link = (uint16_t)l;
        s_ent[e] = en;
      } else {  // no entry slot left: look the candidates up and rank them here
        const size_t v = (size_t)l * 32768 + (size_t)vox;
        const uint32_t c0 = g.cand_off[v], len = g.cand_off[v + 1] - c0;
        if (len == 0u) continue;
        if (g.dbg) {
          atomicAdd(&s_cnt[3], (int)len);
          atomicAdd(&s_cnt[2], 1);
          atomicAdd(&g.dbg[4], 1ull);
        }
        const int f0 = g.off[l];
        float bd = GQ_INF_F;
        unsigned bo = 0xffffffffu;
        int bi = -1;
        for (uint32_t c = c0; c < c0 + len; ++c) {
          const int f = f0 + (int)g.cand_idx[c];
          const GqFace fc = g.rec[f];
          const float d2 = gq_tri_rank(fc, xl - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z));
          const unsigned orig = (unsigned)__float_as_int(fc.r5.z);
          if (d2 < bd || (d2 == bd && orig < bo)) {
            bd = d2;
            bo = orig;
            bi = f;
          }
        }
        const GqSdfOut o = gq_tri_finish(g.rec[bi], xl);
        const float dis = sqrtf(o.dist2 + 1e-8f);
        if (o.sign < 0 && dis > in_dis[h]) {
          in_dis[h] = dis;
          in_link[h] = l;
          in_cl[h] = o.closest;
          in_xl[h] = xl;
        }
      }
    }
  }
  __syncthreads();
  // ---- A2: candidate lists of the entries -- independent look-ups, one round trip for the whole block ----------------
  {
    const int n_pre = min(s_cnt[0], ECAP);
    for (int e = tid; e < n_pre; e += 256) {
      const GqPgEntry en = s_ent[e];
      const size_t v = (size_t)en.link * 32768 + (size_t)en.c0;
      const uint32_t c0 = g.cand_off[v], len = g.cand_off[v + 1] - c0;
      if (len == 0u) {  // voxel neither touches a face nor is inside: the point is outside this link
        s_ent[e].c0 = 0xffffffffu;
        continue;
      }
      if (g.dbg) {
        atomicAdd(&s_cnt[3], (int)len);
        atomicAdd(&g.dbg[4], 1ull);
      }
      s_ent[e].c0 = c0;
      int ib = ICAP;
      if (len <= 0xffffu) ib = atomicAdd(&s_cnt[1], (int)len);
      if (ib + (int)len <= ICAP) {
        s_ekey[e] = ~0ull;
        for (uint32_t j = 0; j < len; ++j) s_item[ib + j] = ((uint32_t)e << 16) | j;
      } else {  // no room in the item list: rank the candidates here; the entry is finished in C like any other
        if (g.dbg) atomicAdd(&s_cnt[2], 1);
        // the part of the item list this entry reserved but does not use must not be read as items
        for (int i = ib; i < ICAP && i < ib + (int)len; ++i) s_item[i] = 0xffffffffu;
        const int f0 = g.off[en.link];
        unsigned long long key = ~0ull;
        for (uint32_t j = 0; j < len; ++j) {
          const unsigned fl = g.cand_idx[c0 + j];
          const GqFace fc = g.rec[f0 + (int)fl];
          const float d2 = gq_tri_rank(fc, gq_mk(en.x - fc.r0.x, en.y - fc.r0.y, en.z - fc.r0.z));
          const unsigned orig = (unsigned)__float_as_int(fc.r5.z) - (unsigned)f0;
          const unsigned long long k = gq_rank_key(d2, orig, fl);
          key = k < key ? k : key;
        }
        s_ekey[e] = key;
      }
    }
  }
  __syncthreads();
  if (g.dbg && tid == 0) {
    atomicAdd(&g.dbg[5], (unsigned long long)s_cnt[3]);
    atomicAdd(&g.dbg[6], (unsigned long long)s_cnt[2]);
    atomicAdd(&g.dbg[7], 1ull);
  }
  const int n_ent = min(s_cnt[0], ECAP), n_item = min(s_cnt[1], ICAP);
#ifdef GQ_BLOCK_TIMES
  if (g.span && tid == 0) {
    g.span[128 + 8 * (size_t)block_id + 2] = __builtin_amdgcn_s_memrealtime();
    g.span[128 + 8 * (size_t)block_id + 5] = (uint64_t)(unsigned)s_cnt[0] | ((uint64_t)(unsigned)s_cnt[1] << 32);
  }
#endif
  // ---- B: one (entry, candidate) ranking per thread and step ----------------------------------------------------
  for (int i = tid; i < n_item; i += 256) {
    const uint32_t it = s_item[i];
    if (it == 0xffffffffu) continue;  // reserved by an entry that was ranked inline
    const int e = (int)(it >> 16);
    const GqPgEntry en = s_ent[e];
    const int f0 = g.off[en.link];
    const unsigned fl = g.cand_idx[en.c0 + (it & 0xffffu)];
    const GqFace fc = g.rec[f0 + (int)fl];
    const float d2 = gq_tri_rank(fc, gq_mk(en.x - fc.r0.x, en.y - fc.r0.y, en.z - fc.r0.z));
    const unsigned orig = (unsigned)__float_as_int(fc.r5.z) - (unsigned)f0;  // original index inside the mesh
    atomicMin(&s_ekey[e], gq_rank_key(d2, orig, fl));
  }
  __syncthreads();
#ifdef GQ_BLOCK_TIMES
  if (g.span && tid == 0) g.span[128 + 8 * (size_t)block_id + 3] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- C: finish the winner of every entry ----------------------------------------------------------------------
  for (int e = tid; e < n_ent; e += 256) {
    const GqPgEntry en = s_ent[e];
    if (en.c0 == 0xffffffffu) continue;  // its voxel has no candidate faces: outside this link
    const int f = g.off[en.link] + (int)(s_ekey[e] & 0xffffull);
    const gq3 xl = gq_mk(en.x, en.y, en.z);
    const GqSdfOut o = gq_tri_finish(g.rec[f], xl);
    if (o.sign < 0) {  // inside the link: dis = +sqrt(d^2 + 1e-8) > 0
      const float dis = sqrtf(o.dist2 + 1e-8f);
      s_ecl[e * 4 + 0] = o.closest.x;
      s_ecl[e * 4 + 1] = o.closest.y;
      s_ecl[e * 4 + 2] = o.closest.z;
      s_ecl[e * 4 + 3] = dis;
      atomicMax(&s_pkey[en.pt], ((unsigned long long)__float_as_uint(dis) << 32) |
                                    ((unsigned long long)(255 - (int)en.link) << 16) | (unsigned long long)e);
    }
  }
  __syncthreads();
#ifdef GQ_BLOCK_TIMES
  if (g.span && tid == 0) g.span[128 + 8 * (size_t)block_id + 4] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- D: outputs ---------------------------------------------------------------------------------------------------
#pragma unroll
  for (int h = 0; h < PPT; ++h) {
    if (!ok[h]) continue;
    // dis for every point (coalesced 4 B); link and gradient only where a link is penetrated -- nothing downstream
    // reads them elsewhere (energy.py:59-61 zeroes dis <= 0), the caller provides zero-initialised buffers
    float best_dis = in_dis[h];
    int best_link = in_link[h];
    gq3 best_cl = in_cl[h], best_xl = in_xl[h];
    const unsigned long long pk = s_pkey[h * 256 + tid];
    if (pk != 0ull) {
      const int e = (int)(pk & 0xffffull);
      const float dis = s_ecl[e * 4 + 3];
      const int l = (int)s_ent[e].link;
      if (dis > best_dis || (dis == best_dis && l < best_link)) {
        best_dis = dis;
        best_link = l;
        best_cl = gq_mk(s_ecl[e * 4], s_ecl[e * 4 + 1], s_ecl[e * 4 + 2]);
        best_xl = gq_mk(s_ent[e].x, s_ent[e].y, s_ent[e].z);
      }
    }
    const size_t o = (size_t)row * g.P + pt[h];
    g.dis[o] = best_link >= 0 ? best_dis : -1e30f;
    if (best_link >= 0) {
      const float* T = s_link + best_link * 24;
      const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
      const gq3 gh = gq_mv(Rl, (1.0f / best_dis) * (best_xl - best_cl));
      g.link[o] = best_link;
      g.gvec[o * 3 + 0] = gh.x;
      g.gvec[o * 3 + 1] = gh.y;
      g.gvec[o * 3 + 2] = gh.z;
    }
  }
  if (g.span) {
    __syncthreads();
    if (tid == 0) gq_span_close(g.span, block_id);
  }
}

// ---- the same query driven by the LINKS instead of the points ------------------------------------------------------------
// gq_pen_grid_body tests every (point, link) pair of a row -- 2500 x 14 sphere tests of which ~0.1 % survive.  Here a row's
// block walks, per link, only the cells of a coarse uniform grid over the object's surface points (gqPointGrid, built
// once per object) that the link's box can touch: the link AABB's eight corners are taken to the world frame, their
// bounding box gives a cell range, and only the points filed under those cells are tested.  The work is proportional to
// the overlaps, not to points x links.  From the point test on (link frame, AABB, occupancy voxel, candidate faces,
// ranking, finish, max over links) it is the code of gq_pen_grid_body, so the two produce the same dis / link / gvec.
// One block of 256 threads = one row; entries / items / per-point keys live in LDS.  If a row has more than GQ_PC_ECAP
// (point, link) overlaps (a hand deep inside the object) the block falls back to gq_pen_grid_body for that row.
#define GQ_PC_ECAP 1024
#define GQ_PC_ICAP 4096
#define GQ_PC_PMAX 4096
__host__ __device__ inline size_t gq_pen_cells_lds_bytes(int L, int P) {
  const size_t mine = (size_t)GQ_PC_ECAP * (sizeof(GqPgEntry) + 8) + (size_t)P * 8 + (size_t)GQ_PC_ICAP * 4 + 32 +
                      (size_t)L * 24 * 4 + (size_t)L * 8 * 4 + (size_t)(L + 1) * 4;
  const size_t old = gq_pen_grid_lds_bytes(L);
  return (mine > old ? mine : old) + 16;
}
__device__ __forceinline__ void gq_pen_cells_body(const GqPenArgs& g, int row, char* lds) {
  unsigned long long* s_ekey = reinterpret_cast<unsigned long long*>(lds);
  unsigned long long* s_pkey = s_ekey + GQ_PC_ECAP;
  GqPgEntry* s_ent = reinterpret_cast<GqPgEntry*>(s_pkey + g.P);
  uint32_t* s_item = reinterpret_cast<uint32_t*>(s_ent + GQ_PC_ECAP);
  float* s_ecl = reinterpret_cast<float*>(s_item);  // after phase B: closest point + dis per entry (4 floats; ICAP >= ECAP x 4)
  int* s_cnt = reinterpret_cast<int*>(s_item + GQ_PC_ICAP);  // [0] entries [1] items [2] overflow flag [3..7] diagnostics
  float* s_link = reinterpret_cast<float*>(s_cnt + 8);
  int* s_rng = reinterpret_cast<int*>(s_link + g.L * 24);  // per link: ix0 iy0 iz0 nx ny nz - -
  int* s_pref = s_rng + g.L * 8;                            // (L+1) prefix of cells per link
  const int tid = threadIdx.x;
  const int obj = row / g.batch_each;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  if (g.span && tid == 0) gq_span_open(g.span, (unsigned)row);
  for (int i = tid; i < g.L * 24; i += 256) {
    const int l = i / 24, k = i % 24;
    float v = 0.0f;
    if (k < 12) v = g.link_T[((size_t)row * g.L + l) * 12 + k];
    else if (k < 20) v = g.aabb[l * 8 + (k - 12)];
    else if (k == 20) v = g.occ_invz[l];
    s_link[i] = v;
  }
  if (tid < 8) s_cnt[tid] = 0;
  for (int pt = tid; pt < g.P; pt += 256) s_pkey[pt] = 0ull;
  const int G = g.G;
  const float* gb = g.grid_box + (size_t)obj * 8;
  if (tid < g.L) {  // cell range of link `tid`: world bounding box of the eight corners of its AABB
    const int l = tid;
    const float* T = g.link_T + ((size_t)row * g.L + l) * 12;
    const float* bb = g.aabb + l * 8;
    float lo[3] = {GQ_INF_F, GQ_INF_F, GQ_INF_F}, hi[3] = {-GQ_INF_F, -GQ_INF_F, -GQ_INF_F};
    for (int c = 0; c < 8; ++c) {
      const gq3 q = gq_mk((c & 1) ? bb[4] : bb[0], (c & 2) ? bb[5] : bb[1], (c & 4) ? bb[6] : bb[2]);
      const gq3 qh = gq_mk(T[0] * q.x + T[1] * q.y + T[2] * q.z + T[3], T[4] * q.x + T[5] * q.y + T[6] * q.z + T[7],
                           T[8] * q.x + T[9] * q.y + T[10] * q.z + T[11]);
      const gq3 qw = gq_mv(R, qh) + gq_mk(hp[0], hp[1], hp[2]);
      lo[0] = fminf(lo[0], qw.x); hi[0] = fmaxf(hi[0], qw.x);
      lo[1] = fminf(lo[1], qw.y); hi[1] = fmaxf(hi[1], qw.y);
      lo[2] = fminf(lo[2], qw.z); hi[2] = fmaxf(hi[2], qw.z);
    }
    int c0[3], n[3];
    bool any = g.off[l + 1] > g.off[l];
    for (int k = 0; k < 3; ++k) {
      // a point p is filed under floor((p - lo_grid) * cells_per_metre); one thousandth of a cell of slack on both sides
      // covers the rounding of the corner transforms (the point test itself is exact and repeated below)
      const float a = (lo[k] - gb[k]) * gb[4 + k] - 1e-3f, b = (hi[k] - gb[k]) * gb[4 + k] + 1e-3f;
      const int i0 = max((int)floorf(a), 0), i1 = min((int)floorf(b), G - 1);
      c0[k] = i0;
      n[k] = i1 - i0 + 1;
      any = any && (b >= 0.0f) && (a < (float)G) && n[k] > 0;
    }
    int* r = s_rng + l * 8;
    r[0] = c0[0]; r[1] = c0[1]; r[2] = c0[2];
    r[3] = any ? n[0] : 0; r[4] = any ? n[1] : 0; r[5] = any ? n[2] : 0;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < g.L; ++l) {
      s_pref[l] = acc;
      acc += s_rng[l * 8 + 3] * s_rng[l * 8 + 4] * s_rng[l * 8 + 5];
    }
    s_pref[g.L] = acc;
  }
  __syncthreads();
  const int n_pairs = s_pref[g.L];
  const int32_t* cstart = g.grid_start + (size_t)obj * (G * G * G + 1);
  const uint16_t* cpts = g.grid_pts + (size_t)obj * g.P;
  // ---- A: (link, cell) pairs -> points -> entries -------------------------------------------------------------------------
  for (int pair = tid; pair < n_pairs; pair += 256) {
    int l = 0;
    while (s_pref[l + 1] <= pair) ++l;
    const int* r = s_rng + l * 8;
    int q = pair - s_pref[l];
    const int cx = r[0] + q % r[3];
    q /= r[3];
    const int cy = r[1] + q % r[4], cz = r[2] + q / r[4];
    const int cid = (cz * G + cy) * G + cx;
    const int k0 = cstart[cid], k1 = cstart[cid + 1];
    if (k1 <= k0) continue;
    const float* T = s_link + l * 24;
    const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const float* bb = T + 12;
    for (int k = k0; k < k1; ++k) {
      const int pt = (int)cpts[k];
      const float* sp = g.surf + ((size_t)obj * g.P + pt) * 3;
      const gq3 xh = gq_mtv(R, gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]));
      const gq3 xl = gq_mtv(Rl, xh - gq_mk(T[3], T[7], T[11]));
      if (!(gq_aabb_dist2(bb, xl) <= 0.0f)) continue;
      const float ux = (xl.x - bb[0]) * bb[3], uy = (xl.y - bb[1]) * bb[7], uz = (xl.z - bb[2]) * T[20];
      const int ix = min(max((int)ux, 0), 31), iy = min(max((int)uy, 0), 31), iz = min(max((int)uz, 0), 31);
      if (!((g.occ[(size_t)l * 1024 + iz * 32 + iy] >> ix) & 1u)) continue;
      const size_t v = (size_t)l * 32768 + (size_t)(iz * 1024 + iy * 32 + ix);
      const uint32_t c0 = g.cand_off[v], len = g.cand_off[v + 1] - c0;
      if (len == 0u) continue;
      if (g.dbg) atomicAdd(&s_cnt[3], (int)len);
      const int e = atomicAdd(&s_cnt[0], 1);
      if (e >= GQ_PC_ECAP) {  // too many overlaps for the LDS lists: the whole row is redone by the dense query
        s_cnt[2] = 1;
        continue;
      }
      int ib = GQ_PC_ICAP;
      if (len <= 0xffffu) ib = atomicAdd(&s_cnt[1], (int)len);
      GqPgEntry en;
      en.x = xl.x; en.y = xl.y; en.z = xl.z;
      en.c0 = c0;
      en.pt = (uint16_t)pt;
      en.link = (uint16_t)l;
      s_ent[e] = en;
      if (ib + (int)len <= GQ_PC_ICAP) {
        s_ekey[e] = ~0ull;
        for (uint32_t j = 0; j < len; ++j) s_item[ib + j] = ((uint32_t)e << 16) | j;
      } else {  // no room for its items: rank the candidates here, phase C finishes the winner like any other entry
        if (g.dbg) atomicAdd(&s_cnt[4], 1);
        for (int i = ib; i < GQ_PC_ICAP && i < ib + (int)len; ++i) s_item[i] = 0xffffffffu;
        const int f0 = g.off[l];
        unsigned long long bk = ~0ull;
        for (uint32_t c = 0; c < len; ++c) {
          const unsigned fl = g.cand_idx[c0 + c];
          const GqFace fc = g.rec[f0 + (int)fl];
          const float d2 = gq_tri_rank(fc, gq_mk(xl.x - fc.r0.x, xl.y - fc.r0.y, xl.z - fc.r0.z));
          const unsigned long long key = gq_rank_key(d2, (unsigned)__float_as_int(fc.r5.z) - (unsigned)f0, fl);
          bk = key < bk ? key : bk;
        }
        s_ekey[e] = bk;
      }
    }
  }
  __syncthreads();
  if (g.dbg && tid == 0) {
    atomicAdd(&g.dbg[4], (unsigned long long)s_cnt[0]);
    atomicAdd(&g.dbg[5], (unsigned long long)s_cnt[3]);
    atomicAdd(&g.dbg[6], (unsigned long long)s_cnt[4] + (s_cnt[2] ? (unsigned long long)s_cnt[0] : 0ull));
    atomicAdd(&g.dbg[7], 1ull);
    atomicAdd(&g.dbg[0], (unsigned long long)n_pairs);
  }
  if (s_cnt[2]) {  // block-uniform
    GqPenArgs h = g;
    h.span = nullptr;
    h.dbg = nullptr;
    __syncthreads();
    for (int bx = 0; bx < (g.P + 255) / 256; ++bx) {
      gq_pen_grid_body<true>(h, bx, row, lds);
      __syncthreads();
    }
    if (g.span && tid == 0) gq_span_close(g.span, (unsigned)row);
    return;
  }
  const int n_ent = s_cnt[0], n_item = min(s_cnt[1], GQ_PC_ICAP);
  // ---- B: one (entry, candidate) ranking per thread and step --------------------------------------------------------------
  for (int i = tid; i < n_item; i += 256) {
    const uint32_t it = s_item[i];
    if (it == 0xffffffffu) continue;
    const int e = (int)(it >> 16);
    const GqPgEntry en = s_ent[e];
    const int f0 = g.off[en.link];
    const unsigned fl = g.cand_idx[en.c0 + (it & 0xffffu)];
    const GqFace fc = g.rec[f0 + (int)fl];
    const float d2 = gq_tri_rank(fc, gq_mk(en.x - fc.r0.x, en.y - fc.r0.y, en.z - fc.r0.z));
    const unsigned orig = (unsigned)__float_as_int(fc.r5.z) - (unsigned)f0;
    atomicMin(&s_ekey[e], gq_rank_key(d2, orig, fl));
  }
  __syncthreads();
  // ---- C: finish the winner of every entry; penetrating entries compete per point ----------------------------------------
  for (int e = tid; e < n_ent; e += 256) {
    const GqPgEntry en = s_ent[e];
    const int f = g.off[en.link] + (int)(s_ekey[e] & 0xffffull);
    const gq3 xl = gq_mk(en.x, en.y, en.z);
    const GqSdfOut o = gq_tri_finish(g.rec[f], xl);
    if (o.sign < 0) {
      const float dis = sqrtf(o.dist2 + 1e-8f);
      s_ecl[e * 4 + 0] = o.closest.x;
      s_ecl[e * 4 + 1] = o.closest.y;
      s_ecl[e * 4 + 2] = o.closest.z;
      s_ecl[e * 4 + 3] = dis;
      atomicMax(&s_pkey[en.pt], ((unsigned long long)__float_as_uint(dis) << 32) |
                                    ((unsigned long long)(255 - (int)en.link) << 16) | (unsigned long long)e);
    }
  }
  __syncthreads();
  // ---- D: outputs (dis for every point; link / gradient only where a link is penetrated) ---------------------------------
  for (int pt = tid; pt < g.P; pt += 256) {
    const unsigned long long pk = s_pkey[pt];
    const size_t o = (size_t)row * g.P + pt;
    if (pk == 0ull) {
      g.dis[o] = -1e30f;
      continue;
    }
    const int e = (int)(pk & 0xffffull);
    const float dis = s_ecl[e * 4 + 3];
    const int l = (int)s_ent[e].link;
    const float* T = s_link + l * 24;
    const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const gq3 xl = gq_mk(s_ent[e].x, s_ent[e].y, s_ent[e].z);
    const gq3 cl = gq_mk(s_ecl[e * 4], s_ecl[e * 4 + 1], s_ecl[e * 4 + 2]);
    const gq3 gh = gq_mv(Rl, (1.0f / dis) * (xl - cl));
    g.dis[o] = dis;
    g.link[o] = l;
    g.gvec[o * 3 + 0] = gh.x;
    g.gvec[o * 3 + 1] = gh.y;
    g.gvec[o * 3 + 2] = gh.z;
  }
  if (g.span) {
    __syncthreads();
    if (tid == 0) gq_span_close(g.span, (unsigned)row);
  }
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
  const float* dis;  // when w == nullptr: w = w_pen * [dis > 0]  (energy.py:59-61), and e_pen[row] = sum relu(dis)
  float w_pen;
  float* e_pen;
  uint64_t* span;      // optional: the forward query's 64 x {min start, max end} shards ...
  uint64_t* span_acc;  // ... folded into {sum of spans, launches} and re-armed here (the query is over by now)
};

// One block per row, surface points in rounds of up to 4096.  Phase A: every thread reads the weights of its 16 points
// (loads in flight together), the contributing ones (w != 0) are compacted IN POINT ORDER into an LDS list of
// (link, w*G, x_h) records (ballot prefix inside a wave, (slice, wave) counts across the block) -- the order is
// independent of scheduling.  Phase B: thread a < L*6 + 12 owns one accumulator and folds the list in order.
#define GQ_PENB_K 16
#define GQ_PENB_LIST 1024
__host__ __device__ inline size_t gq_pen_bwd_lds_bytes() {
  return (size_t)GQ_PENB_LIST * 6 * 4 + GQ_PENB_K * 4 * 4 + GQ_PENB_LIST;
}
// block = 256 threads = one row.  K = 256-point slices taken up per round (K <= GQ_PENB_K): every slice costs eight
// registers per thread whether it exists or not, so the launcher picks the smallest K that covers the surface points in
// one round (K = 10 for the 2500 points of the reference: 120 instead of 162 VGPRs for the stand-alone kernel).  The
// order of all sums is independent of K.
template <int K = GQ_PENB_K>
__device__ __forceinline__ void gq_pen_bwd_body(const GqPenBwdArgs& g, int row, char* lds) {
  static_assert(K >= 4 && K <= GQ_PENB_K, "a round must fit the LDS list and the counter table");
  float* s_rec = reinterpret_cast<float*>(lds);
  int* s_cnt = reinterpret_cast<int*>(s_rec + GQ_PENB_LIST * 6);
  unsigned char* s_lnk = reinterpret_cast<unsigned char*>(s_cnt + GQ_PENB_K * 4);
  const int tid = threadIdx.x, lane = gq_lane(), wv = tid / GQ_WAVE;
  const int obj = row / g.batch_each;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  if (g.span && row == 0 && wv == 0) {
    unsigned long long t0 = g.span[2 * lane], t1 = g.span[2 * lane + 1];
    if (t1 == 0ull) t0 = ~0ull;  // shard saw no block
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a0 = __shfl_xor(t0, o, GQ_WAVE), a1 = __shfl_xor(t1, o, GQ_WAVE);
      t0 = a0 < t0 ? a0 : t0;
      t1 = a1 > t1 ? a1 : t1;
    }
    g.span[2 * lane] = ~0ull;
    g.span[2 * lane + 1] = 0ull;
    if (lane == 0 && t1 > t0) {
      g.span_acc[0] += t1 - t0;
      g.span_acc[1] += 1ull;
    }
  }
  float e_acc = 0.0f;
  for (int base = 0; base < g.P;) {
    float w[K], dpos[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int pt = base + k * 256 + tid;
      dpos[k] = 0.0f;
      if (g.w) {
        w[k] = (pt < g.P) ? g.w[(size_t)row * g.P + pt] : 0.0f;
      } else {
        const float d = (pt < g.P) ? g.dis[(size_t)row * g.P + pt] : 0.0f;
        w[k] = d > 0.0f ? g.w_pen : 0.0f;
        dpos[k] = d > 0.0f ? d : 0.0f;
      }
    }
    unsigned long long m[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      m[k] = __ballot(w[k] != 0.0f);
      if (lane == 0) s_cnt[k * 4 + wv] = __popcll(m[k]);
    }
    __syncthreads();
    // as many 256-point slices as fit into the LDS list (a slice holds <= 256 entries, so at least four always do);
    // the remaining slices are taken up again by the next round.  The (slice, wave) counts are read once, 16 B each.
    int4 cnt4[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cnt4[k] = *reinterpret_cast<const int4*>(s_cnt + k * 4);
    int kfit = 0, run = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tot = cnt4[k].x + cnt4[k].y + cnt4[k].z + cnt4[k].w;
      if (kfit == k && run + tot <= GQ_PENB_LIST) {
        kfit = k + 1;
        run += tot;
      }
    }
    run = 0;  // entries before slice k
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k < kfit) {
        const int off = run + (wv > 0 ? cnt4[k].x : 0) + (wv > 1 ? cnt4[k].y : 0) + (wv > 2 ? cnt4[k].z : 0);
        run += cnt4[k].x + cnt4[k].y + cnt4[k].z + cnt4[k].w;
        e_acc += dpos[k];
        if (w[k] != 0.0f) {
          const int pt = base + k * 256 + tid;
          const size_t o = (size_t)row * g.P + pt;
          const int i = off + __popcll(m[k] & ((1ull << lane) - 1ull));
          const float* sp = g.surf + ((size_t)obj * g.P + pt) * 3;
          const gq3 xh = gq_mtv(R, gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]));
          s_rec[i * 6 + 0] = w[k] * g.gvec[o * 3];
          s_rec[i * 6 + 1] = w[k] * g.gvec[o * 3 + 1];
          s_rec[i * 6 + 2] = w[k] * g.gvec[o * 3 + 2];
          s_rec[i * 6 + 3] = xh.x;
          s_rec[i * 6 + 4] = xh.y;
          s_rec[i * 6 + 5] = xh.z;
          s_lnk[i] = (unsigned char)g.link[o];
        }
      }
    }
    const bool first = base == 0;
    base += kfit * 256;
    __syncthreads();
    const int n = run;  // block-uniform
    // fold, wave-parallel and in a fixed order: a wavefront takes a group of 4 links (24 accumulators) or the group of
    // the 12 global sums; lane j adds entries j, j+64, ... into registers, then the fixed DPP tree adds the lanes.
    //   wrench f_l -= G, m_l -= x_h x G ; gsum += G ; K += x_h (x) G      (G = r[0..2], x_h = r[3..5])
    const int n_lgroups = (g.L + 3) / 4;
    // the 12 global sums ride in accumulators 12..23 of the last link group when that group has at most two links
    // (Allegro: 14 links -> 4 groups, one per wavefront); otherwise they are a group of their own
    const bool ride = g.L - 4 * (n_lgroups - 1) <= 2;
    for (int grp = wv; grp < (ride ? n_lgroups : n_lgroups + 1); grp += 4) {  // wave-uniform
      float part[24];
#pragma unroll
      for (int q = 0; q < 24; ++q) part[q] = 0.0f;
      const int l0 = grp * 4;
      const bool ride_here = ride && grp == n_lgroups - 1;
      for (int i = lane; i < n; i += GQ_WAVE) {
        const float* r = s_rec + i * 6;
        const gq3 G = gq_mk(r[0], r[1], r[2]), x = gq_mk(r[3], r[4], r[5]);
        if (grp < n_lgroups) {
          const gq3 mm = gq_cross(x, G);
          const int dl = (int)s_lnk[i] - l0;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float sel = (dl == q4) ? 1.0f : 0.0f;
            part[q4 * 6 + 0] -= sel * G.x;
            part[q4 * 6 + 1] -= sel * G.y;
            part[q4 * 6 + 2] -= sel * G.z;
            part[q4 * 6 + 3] -= sel * mm.x;
            part[q4 * 6 + 4] -= sel * mm.y;
            part[q4 * 6 + 5] -= sel * mm.z;
          }
          if (ride_here) {
            part[12] += G.x; part[13] += G.y; part[14] += G.z;
            part[15] += x.x * G.x; part[16] += x.x * G.y; part[17] += x.x * G.z;
            part[18] += x.y * G.x; part[19] += x.y * G.y; part[20] += x.y * G.z;
            part[21] += x.z * G.x; part[22] += x.z * G.y; part[23] += x.z * G.z;
          }
        } else {
          part[0] += G.x; part[1] += G.y; part[2] += G.z;
          part[3] += x.x * G.x; part[4] += x.x * G.y; part[5] += x.x * G.z;
          part[6] += x.y * G.x; part[7] += x.y * G.y; part[8] += x.y * G.z;
          part[9] += x.z * G.x; part[10] += x.z * G.y; part[11] += x.z * G.z;
        }
      }
      const int nq = (grp < n_lgroups) ? 24 : 12;
      gq_wave_sums_f<24>(part);  // all 24 lane sums at once (pairwise folding, fixed order)
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        if (q < nq) {
          const float tot = part[q];
          if (lane == 0) {
            if (grp < n_lgroups) {
              const int l = l0 + q / 6;
              if (l < g.L) {
                float* dst = g.wrench + ((size_t)row * g.L + l) * 6 + (q % 6);
                *dst = (first ? 0.0f : *dst) + tot;
              } else if (ride_here && q >= 12) {
                float* dst = g.gRt + (size_t)row * 12 + (q - 12);
                *dst = (first ? 0.0f : *dst) + tot;
              }
            } else {
              float* dst = g.gRt + (size_t)row * 12 + q;
              *dst = (first ? 0.0f : *dst) + tot;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (g.e_pen) {  // fixed order: thread-local (k ascending), DPP tree per wave, waves 0..3
    const float ws = gq_dpp_sum(e_acc);
    if (lane == 0) s_rec[wv] = ws;
    __syncthreads();
    if (tid == 0) g.e_pen[row] = ((s_rec[0] + s_rec[1]) + s_rec[2]) + s_rec[3];
  }
}


// ---- host side: argument blocks of the penetration-only query and its fused-E_pen backward ---------------------------
static inline int gq_pen_fill(const gqMeshSet* links, const float* surface_points, int64_t n_obj, int64_t n_surface,
                              int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                              const float* link_T, float* dis, int32_t* link, float* gvec, uint64_t* span, GqPenArgs* out,
                              const gqPointGrid* grid = nullptr) {
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
  a.occ = links->occ_dev;
  a.occ_invz = links->occ_invz_dev;
  a.cand_off = links->cand_off_dev;
  a.cand_idx = links->cand_idx_dev;
  a.B = (int)(n_obj * batch_each);
  a.P = (int)n_surface;
  a.L = links->n_mesh;
  a.D = pose_dim;
  a.batch_each = (int)batch_each;
  a.dis = dis;
  a.link = link;
  a.gvec = gvec;
  a.span = span;
  if (grid) {
    GQ_REQUIRE(grid->n_obj == (int)n_obj && grid->P == (int)n_surface, "hand_pen_forward: the point grid was built for "
               "%d objects x %d points, not %lld x %lld", grid->n_obj, grid->P, (long long)n_obj, (long long)n_surface);
    GQ_REQUIRE(n_surface <= GQ_PC_PMAX, "hand_pen_forward: the cell-driven query handles at most %d surface points", GQ_PC_PMAX);
    a.grid_box = grid->box_dev;
    a.grid_start = grid->start_dev;
    a.grid_pts = grid->pts_dev;
    a.G = grid->G;
  }
  GQ_REQUIRE(a.B <= 65535 || grid, "hand_pen_forward: B=%d exceeds grid.y limit", a.B);
  *out = a;
  return GQ_OK;
}
static inline int gq_pen_bwd_fill(int n_links, const float* surface_points, int64_t n_obj, int64_t n_surface,
                                  int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                                  const float* grad_dis, const int32_t* link, const float* gvec, float* link_wrench,
                                  float* gRt, const float* dis, float w_pen, float* e_pen, uint64_t* span,
                                  uint64_t* span_acc, GqPenBwdArgs* out) {
  GQ_REQUIRE(surface_points && hand_pose && Rg && link && gvec && link_wrench && gRt, "hand_pen_backward: null");
  GQ_REQUIRE(grad_dis || (dis && e_pen), "hand_pen_backward: need grad_dis, or dis + e_pen for the fused E_pen form");
  GQ_REQUIRE(!span || span_acc, "hand_pen_backward: span without span_acc");
  GQ_REQUIRE(n_links > 0 && n_links <= 160 && n_obj > 0 && n_surface > 0 && batch_each > 0, "hand_pen_backward: sizes");
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
  a.dis = dis;
  a.w_pen = w_pen;
  a.e_pen = grad_dis ? nullptr : e_pen;
  a.span = span;
  a.span_acc = span_acc;
  *out = a;
  return GQ_OK;
}
