// Hand kinematics for the grasp loop: rot6d -> R, URDF tree FK, contact points / normals, penetration spheres,
// and the analytic backward pass (geometric Jacobian via per-node wrench accumulation) that replaces autograd
// through pytorch_kinematics (reference hand_model.py:762-766, 787-873, 1220-1267; utils/transforms.py:5-13).
//
// Work shape: one thread per grasp candidate (row).  Per row the work is ~2 kflop of strictly sequential 3x3
// products down a 16-24 joint tree, i.e. latency- not throughput-bound; the per-row scratch (node transforms,
// node wrenches) lives in a caller-provided workspace so nothing is dynamically indexed in registers.
#include "kin_dev.h"
#include "loop_dev.h"
#include "sdf_dev.h"
#include "wave.h"

#ifndef GQ_FK_QUERY_TOPK
// clusters per round of the contact queries that ride in the FK forward block.  2 (round 3): two instead of four face records in
// flight take the kernel from 168 VGPRs + 144 B of scratch to 156 VGPRs without scratch, same answers (the search is exact),
// +0.1 .. +0.6 % on configs[1] (profiles/r03_ab_fk_topk2.txt); the stand-alone query kernel keeps 4
#define GQ_FK_QUERY_TOPK 2
#endif

struct GqFkArgs {
  gqHand h;
  const float* hand_pose;  // (B, D)  D = 9 + J
  const int64_t* idx;      // (B, n)
  int B, n, D;
  float* Rg;        // (B, 9)
  float* link_T;    // (B, L, 12)
  float* node_W;    // (B, J, 12) workspace / saved for backward
  float* cpts;      // (B, n, 3)
  float* cnrm;      // (B, n, 3)
  float* spheres;   // (B, S, 3) or null
  float* e_spen;    // (B) or null: self penetration of the spheres rides along (needs spheres)
  float* g_spheres; // (B, S, 3): spen_scale * dE_spen / d centre
  float spen_scale;
  int has_propose;   // MalaStar.try_step runs first in the same wavefront and WRITES hand_pose / idx
  GqProposeArgs pr;
  int has_sdf;       // the contact queries of the row are answered in the same launch
  GqWaveArgs sdf;
};

// One wavefront does the kinematics of the row (the barriers below are wavefront-level: blocks of this kernel are either
// a single wavefront, or -- with the object SDF attached -- wavefront 0 plus query wavefronts that wait at the one
// block barrier further down).
// Column means of clip(grad)^2 over ALL rows (optimizer.py:231) without a launch of their own: every block computes them
// redundantly (B <= 512 rows x D floats, L2 hits) with its query wavefronts, which have nothing to do before the
// kinematics are through.  Lane groups k = (wave - 1) * G + grp (G = 64 / D groups of D lanes per wavefront) take the
// units k, k + K, ... of the canonical order (loop_dev.h); wavefront 0 adds the 16 partial sums in unit order.
__device__ __forceinline__ void gq_colsq_partial(const GqProposeArgs& p, int wv, int nw, int lane, float* sPart) {
  const int D = p.D, G = GQ_WAVE / D, grp = lane / D, col = lane % D, K = (nw - 1) * G;
  if (grp < G)
    for (int u = (wv - 1) * G + grp; u < GQ_COLSQ_UNITS; u += K) sPart[u * D + col] = gq_colsq_unit(p.grad, p.B, D, p.clip, u, col);
}

__device__ __forceinline__ void gq_fk_forward_row(const GqFkArgs& g, int row, int lane, float* sW, float* sT, float* sC,
                                                  unsigned long long* sKey, float* sRad, float* sCP, float* sPose,
                                                  const float* sPart = nullptr) {
  const gqHand& h = g.h;
  // The constant hand tables this lane needs (its joint node, its link, its first sphere, its sphere group) are loaded
  // BEFORE anything else: a single wavefront per row hides no latency, and the barriers / fences below would otherwise
  // turn every table into a dependent memory round trip of its own.
  const int slot_now = (g.has_propose && g.pr.slot_ctr) ? g.pr.slot_ctr[0] : -1;  // requested first: loads depend on it
  int parent = -1, depth = -1, ntype = 0, lnode = -1;
  GqT pre = gq_t_identity(), loff = gq_t_identity();
  gq3 ax = gq_mk(0, 0, 0);
  if (lane < h.J) {
    parent = h.node_parent[lane];
    depth = h.node_depth[lane];
    ntype = h.node_type[lane];
    pre = gq_t_load(h.node_pre + lane * 12);
    ax = gq_mk(h.node_axis[lane * 3], h.node_axis[lane * 3 + 1], h.node_axis[lane * 3 + 2]);
  }
  if (lane < h.L) {
    lnode = h.link_node[lane];
    loff = gq_t_load(h.link_offset + lane * 12);
  }
  const float* hp = g.hand_pose + (size_t)row * g.D;
  int64_t my_idx = 0;
#ifdef GQ_FK_PREFETCH_BOTH
  bool have_tables = false;
  int clS = 0;
  gq3 cpS = gq_mk(0, 0, 0), cnS = gq_mk(0, 0, 0);
#endif
  const bool z_here = !sCP || (int)blockDim.x < 2 * GQ_WAVE;  // otherwise wavefront 1 computes the z-score while it waits
  if (g.has_propose) {  // the proposal of this row, then its forward kinematics: pose and indices are handed over in
    // LDS / registers, no wait for its own stores.  All operands of the proposal are requested BEFORE the barrier behind
    // which the column means become available (one memory round trip less on the row's critical path).
    const GqProposePre pp = gq_propose_prefetch(g.pr, row, lane, slot_now);
#ifdef GQ_FK_PREFETCH_BOTH
    // Diagnostic variant (round-2 miscompare): candidate tables of BOTH possible contacts (drawn / current
    // index) requested before barrier A, selected after the proposal.
    int clN = 0, clO = 0;
    gq3 cpN = gq_mk(0, 0, 0), cnN = gq_mk(0, 0, 0), cpO = gq_mk(0, 0, 0), cnO = gq_mk(0, 0, 0);
    if (lane < g.n) {
      const int a = (int)pp.nix, b = (int)pp.oix;
      clN = h.cand_link[a];
      cpN = gq_mk(h.cand_pos[a * 3], h.cand_pos[a * 3 + 1], h.cand_pos[a * 3 + 2]);
      cnN = gq_mk(h.cand_nrm[a * 3], h.cand_nrm[a * 3 + 1], h.cand_nrm[a * 3 + 2]);
      clO = h.cand_link[b];
      cpO = gq_mk(h.cand_pos[b * 3], h.cand_pos[b * 3 + 1], h.cand_pos[b * 3 + 2]);
      cnO = gq_mk(h.cand_nrm[b * 3], h.cand_nrm[b * 3 + 1], h.cand_nrm[b * 3 + 2]);
    }
#endif
    float g2v[2];
    if (sPart) {  // block barrier A: the query wavefronts have left their partial column sums
      __syncthreads();
      g2v[0] = lane < g.pr.D ? gq_colsq_finish(sPart, g.pr.B, g.pr.D, lane) : 0.0f;
      g2v[1] = 0.0f;  // g2_inline requires D <= 64
      if (row == 0 && lane < g.pr.D) const_cast<float*>(g.pr.g2)[lane] = g2v[0];  // kept observable (g2_scratch)
    } else {
      g2v[0] = lane < g.pr.D ? g.pr.g2[lane] : 0.0f;
      g2v[1] = lane + GQ_WAVE < g.pr.D ? g.pr.g2[lane + GQ_WAVE] : 0.0f;
    }
    gq_propose_finish(g.pr, pp, row, lane, g2v, sPose, &my_idx, z_here);
#ifdef GQ_FK_PREFETCH_BOTH
    {
      const bool sw = pp.us < g.pr.switch_p;  // the rule of gq_propose_finish
      have_tables = true;
      clS = sw ? clN : clO;
      cpS = sw ? cpN : cpO;
      cnS = sw ? cnN : cnO;
    }
#endif
    gq_wave_sync();
    hp = sPose;
    if (g.n > GQ_WAVE) __threadfence_block();  // contacts beyond the first 64 re-read their indices from memory
  } else if (lane < g.n) {
    my_idx = g.idx[(size_t)row * g.n + lane];
  }
  // contact candidates of the (just proposed) indices: in flight while the tree is walked
  int cl0 = 0;
  gq3 cp0 = gq_mk(0, 0, 0), cn0 = gq_mk(0, 0, 0);
#ifdef GQ_FK_PREFETCH_BOTH
  if (have_tables) {
    cl0 = clS;
    cp0 = cpS;
    cn0 = cnS;
  } else
#endif
  if (lane < g.n) {
    const int ci = (int)my_idx;
    cl0 = h.cand_link[ci];
    cp0 = gq_mk(h.cand_pos[ci * 3], h.cand_pos[ci * 3 + 1], h.cand_pos[ci * 3 + 2]);
    cn0 = gq_mk(h.cand_nrm[ci * 3], h.cand_nrm[ci * 3 + 1], h.cand_nrm[ci * 3 + 2]);
  }
  float R[9];
  gq_rot6d(hp + 3, R);
  if (lane < 9) g.Rg[(size_t)row * 9 + lane] = R[lane];
  const gq3 tg = gq_mk(hp[0], hp[1], hp[2]);
  GqT A = gq_t_identity();
  if (lane < h.J) {
    float q;
    if (h.coup) {  // coupled hand: theta_tree = C theta_act + c0 (hands/ability_hand.py:9-26, panda.py:6-14)
      q = h.coup0[lane];
      for (int a = 0; a < h.JA; ++a) q = fmaf(h.coup[lane * h.JA + a], hp[9 + a], q);
    } else {
      q = hp[9 + lane];
    }
    A = gq_t_mul(pre, gq_joint_motion(ntype, ax, q));
    if (parent < 0) gq_t_store(sW + lane * 12, A);
  }
  gq_wave_sync();
  for (int d = 1; d <= h.max_depth; ++d) {  // level by level down the tree
    if (depth == d) {
      A = gq_t_mul(gq_t_load(sW + parent * 12), A);
      gq_t_store(sW + lane * 12, A);
    }
    gq_wave_sync();
  }
  if (lane < h.J) gq_t_store(g.node_W + ((size_t)row * h.J + lane) * 12, A);
  if (lane < h.L) {
    GqT T = loff;
    if (lnode >= 0) T = gq_t_mul(gq_t_load(sW + lnode * 12), T);
    gq_t_store(sT + lane * 12, T);
    gq_t_store(g.link_T + ((size_t)row * h.L + lane) * 12, T);
  }
  gq_wave_sync();
  for (int c = lane; c < g.n; c += GQ_WAVE) {
    int cl = cl0;
    gq3 cp = cp0, cn = cn0;
    if (c >= GQ_WAVE) {
      const int ci = (int)g.idx[(size_t)row * g.n + c];
      cl = h.cand_link[ci];
      cp = gq_mk(h.cand_pos[ci * 3], h.cand_pos[ci * 3 + 1], h.cand_pos[ci * 3 + 2]);
      cn = gq_mk(h.cand_nrm[ci * 3], h.cand_nrm[ci * 3 + 1], h.cand_nrm[ci * 3 + 2]);
    }
    const GqT T = gq_t_load(sT + cl * 12);
    const gq3 pw = gq_mv(R, gq_t_apply(T, cp)) + tg;
    const gq3 nw = gq_mv(R, gq_t_rot(T, cn));
    float* o = g.cpts + ((size_t)row * g.n + c) * 3;
    o[0] = pw.x; o[1] = pw.y; o[2] = pw.z;
    if (sCP && c < GQ_WAVE) {
      sCP[c * 3] = pw.x; sCP[c * 3 + 1] = pw.y; sCP[c * 3 + 2] = pw.z;
    }
    o = g.cnrm + ((size_t)row * g.n + c) * 3;
    o[0] = nw.x; o[1] = nw.y; o[2] = nw.z;
  }
  if (g.spheres) {
    GqSpenArgs sa;
    sa.spheres = g.spheres;
    sa.e_spen = g.e_spen;
    sa.g_spheres = g.g_spheres;
    sa.spen_scale = g.spen_scale;
    gq_spheres_row(h, sa, sT, R, tg, row, lane, sC, sKey, sRad);
  }
}

__device__ __forceinline__ gq3 gq_fk_contact_point(const GqFkArgs& g, const float* sCP, int row, int c) {
  if (c < GQ_WAVE) return gq_mk(sCP[c * 3], sCP[c * 3 + 1], sCP[c * 3 + 2]);
  const float* q = g.cpts + ((size_t)row * g.n + c) * 3;  // > 64 contacts: written by wavefront 0 before the barrier
  return gq_mk(q[0], q[1], q[2]);
}

// block = one row.  has_sdf: blockDim = 64 * (number of query wavefronts); after the kinematics every wavefront answers
// contact queries c = wave, wave + nw, ... of the row against the row's object mesh (gq_sdf_wave_query).
// The kinematics alone (no contact queries in the launch: batches > 512 rows, the autograd route): one wavefront per
// block and no 170-register ceiling, so nothing spills and the query code is not part of the kernel at all.
__global__ __launch_bounds__(GQ_WAVE) void gq_fk_forward_row_kernel(GqFkArgs g) {
  __shared__ float sW[64 * 12];
  __shared__ float sT[64 * 12];
  __shared__ float sC[256 * 3];
  __shared__ unsigned long long sKey[64];
  __shared__ float sRad[256];
  __shared__ float sPose[128];
  gq_fk_forward_row(g, (int)blockIdx.x, gq_lane(), sW, sT, sC, sKey, sRad, nullptr, sPose);
}

__global__ __launch_bounds__(768) void gq_fk_forward_kernel(GqFkArgs g) {
  __shared__ float sW[64 * 12];
  __shared__ float sT[64 * 12];
  __shared__ float sC[256 * 3];             // world sphere centres (self penetration)
  __shared__ unsigned long long sKey[64];   // per sphere group: (pen, a, b) of the most penetrating pair
  __shared__ float sRad[256];
  __shared__ float sCP[GQ_WAVE * 3];        // world contact points handed to the query wavefronts
  __shared__ float sPose[128];              // the proposed pose (head of the row's kinematics)
  __shared__ float sPart[GQ_COLSQ_UNITS * GQ_WAVE];  // partial column sums of the squared gradient (g2_inline)
  const int row = blockIdx.x, lane = gq_lane(), wv = (int)threadIdx.x / GQ_WAVE;
  const int nw = (int)blockDim.x / GQ_WAVE;
  const bool g2_here = g.has_propose && g.pr.g2_inline;
  if (!g.has_sdf) {
    gq_fk_forward_row(g, row, lane, sW, sT, sC, sKey, sRad, nullptr, sPose);
    return;
  }
  // Two code paths so that the prefetched boxes are not live across the kinematics: wavefront 0 does the kinematics,
  // the others fetch mesh offsets and cluster boxes of their first query meanwhile; everybody meets at ONE barrier.
  if (wv == 0) {
    __builtin_amdgcn_s_setprio(3);  // the kinematics wavefront is the block's critical path until the queries start
    gq_fk_forward_row(g, row, lane, sW, sT, sC, sKey, sRad, sCP, sPose, g2_here ? sPart : nullptr);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    for (int c = 0; c < g.n; c += nw) {
      const GqSdfPre pre = gq_sdf_wave_prefetch(g.sdf, (int64_t)row * g.n + c, lane);
      gq_sdf_wave_query<GQ_FK_QUERY_TOPK>(g.sdf, (int64_t)row * g.n + c, gq_fk_contact_point(g, sCP, row, c), lane, pre);
    }
  } else {
    if (g2_here) {
      gq_colsq_partial(g.pr, wv, nw, lane, sPart);
      __syncthreads();  // barrier A (wavefront 0 waits in gq_fk_forward_row)
    }
    GqSdfPre pre;
    if (wv < g.n) pre = gq_sdf_wave_prefetch(g.sdf, (int64_t)row * g.n + wv, lane);
    // the z-score of the old energies (only the accept step needs it) is computed by wavefront 1 while it waits
    if (wv == 1 && g.has_propose) gq_zscore_row(g.pr, row, lane);
    __syncthreads();
    if (wv < g.n) gq_sdf_wave_query<GQ_FK_QUERY_TOPK>(g.sdf, (int64_t)row * g.n + wv, gq_fk_contact_point(g, sCP, row, wv), lane, pre);
    for (int c = wv + nw; c < g.n; c += nw) {
      const GqSdfPre p2 = gq_sdf_wave_prefetch(g.sdf, (int64_t)row * g.n + c, lane);
      gq_sdf_wave_query<GQ_FK_QUERY_TOPK>(g.sdf, (int64_t)row * g.n + c, gq_fk_contact_point(g, sCP, row, c), lane, p2);
    }
  }
}

struct GqFkBwdArgs {
  gqHand h;
  const float* hand_pose;
  const int64_t* idx;
  const float* Rg;
  const float* link_T;
  const float* node_W;
  const float* g_cpts;     // (B,n,3) or null
  const float* g_cnrm;     // (B,n,3) or null
  const float* g_spheres;  // (B,S,3) or null
  const float* g_wrench;   // (B,L,6) hand-frame link wrench (f, m about hand origin) or null
  const float* g_Rt;       // (B,12): [gsum(3), K(9)] from the penetration query or null
  const float* g_theta;    // (B,J) direct joint-angle gradient (E_joints) or null
  const float* g_R;        // (B,9) direct gradient on the global rotation matrix or null
  int B, n, D;
  float* node_F;     // (B, J, 6) workspace
  float* grad_pose;  // (B, D)
  gqRowEnergyDesc en;  // en.total != nullptr: E_dis, E_joints (+ its gradient) and the weighted total ride along
  int has_accept;      // MalaStar.accept_step runs last in the same wavefront (needs en.total)
  GqAcceptArgs ac;
};

__device__ __forceinline__ void gq_add6(float* a, gq3 f, gq3 m) {
  a[0] += f.x; a[1] += f.y; a[2] += f.z;
  a[3] += m.x; a[4] += m.y; a[5] += m.z;
}

// one wavefront per row.  Items (contacts, spheres) are processed one per lane into (f, m, node) records in LDS; lane j
// then folds, in item order, the records and link wrenches that ride on node j; children are folded into parents level
// by level; the twelve global-pose sums use the fixed DPP tree.  Every sum has a fixed order -> reproducible.
#define GQ_FK_MAX_ITEMS 320
__global__ __launch_bounds__(GQ_WAVE) void gq_fk_backward_kernel(GqFkBwdArgs g) {
  __shared__ float sI[GQ_FK_MAX_ITEMS * 6];
  __shared__ int sIn[GQ_FK_MAX_ITEMS];
  __shared__ float sNF[64 * 6];
  __shared__ float sLT[64 * 12];   // link transforms of this row
  __shared__ float sWr[64 * 6];    // link wrenches of this row
  __shared__ int sLN[64];          // link -> node
  __shared__ int sCh[64];          // child lists
  __shared__ float sG[128];        // the row's new gradient (accept step)
  const int row = blockIdx.x, lane = gq_lane();
  const gqHand& h = g.h;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  const float* LT = g.link_T + (size_t)row * h.L * 12;
  const float* W = g.node_W + (size_t)row * h.J * 12;
  const int n_c = (g.g_cpts || g.g_cnrm) ? g.n : 0;
  const int n_s = g.g_spheres ? h.S : 0;
  const int n_items = n_c + n_s;
  // ---- everything that does not depend on another load is requested first (a single wavefront hides no latency) ----
  for (int i = lane; i < h.L * 12; i += GQ_WAVE) sLT[i] = LT[i];
  if (g.g_wrench)
    for (int i = lane; i < h.L * 6; i += GQ_WAVE) sWr[i] = g.g_wrench[(size_t)row * h.L * 6 + i];
  if (lane < h.L) sLN[lane] = h.link_node[lane];
  int depth = -1, ch0 = 0, ch1 = 0, ntype = 0;
  gq3 axj = gq_mk(0, 0, 0);
  GqT Wj = gq_t_identity();
  if (lane < h.J) {
    depth = h.node_depth[lane];
    ch0 = h.child_off[lane];
    ch1 = h.child_off[lane + 1];
    ntype = h.node_type[lane];
    axj = gq_mk(h.node_axis[lane * 3], h.node_axis[lane * 3 + 1], h.node_axis[lane * 3 + 2]);
    Wj = gq_t_load(W + lane * 12);
  }
  if (lane < h.child_off[h.J]) sCh[lane] = h.child_idx[lane];  // J - (number of roots) entries
  // inputs of the energy tail and of the accept step (independent of everything computed here)
  float en_d2 = 0.0f, en_sg = 0.0f, en_jhi = 0.0f, en_jlo = 0.0f, en_th = 0.0f, en_efc = 0.0f, en_epen = 0.0f, en_espen = 0.0f;
  gq3 en_on = gq_mk(0, 0, 0), en_nh = gq_mk(0, 0, 0);
  if (g.en.total) {
    if (lane < g.en.n) {
      const size_t t = (size_t)row * g.en.n + lane;
      en_d2 = g.en.dist_sq[t];
      en_sg = (float)g.en.sign[t];
      en_on = gq_mk(g.en.obj_dir[t * 3], g.en.obj_dir[t * 3 + 1], g.en.obj_dir[t * 3 + 2]);
      en_nh = gq_mk(g.en.hand_normals[t * 3], g.en.hand_normals[t * 3 + 1], g.en.hand_normals[t * 3 + 2]);
    }
    if (lane < h.JA) {
      en_jhi = g.en.joints_upper[lane];
      en_jlo = g.en.joints_lower[lane];
      en_th = hp[9 + lane];
    }
    en_efc = g.en.e_fc[row];
    en_epen = g.en.e_pen[row];
    en_espen = g.en.e_spen[row];
  }
  // what lane 0 needs at the very end (global-pose part): fetched now, one value per lane, handed over by readlane
  float rt_l = 0.0f, pose_l = 0.0f;
  if (g.g_Rt && lane < 12) rt_l = g.g_Rt[(size_t)row * 12 + lane];
  if (lane < 9) pose_l = hp[lane];
  GqAcceptPre ap{};
  if (g.has_accept) ap = gq_accept_prefetch(g.ac, row, lane);
  __syncthreads();
  // ---- items: per-lane contribution to the global pose + (f, m) on the carrying node -----------------------------
  float acc[12];  // gt (3), gR (9) partial sums of this lane
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
  for (int it = lane; it < n_items; it += GQ_WAVE) {
    gq3 f = gq_mk(0, 0, 0), m = gq_mk(0, 0, 0);
    int l;
    if (it < n_c) {
      const int ci = (int)g.idx[(size_t)row * g.n + it];
      l = h.cand_link[ci];
      const GqT T = gq_t_load(sLT + l * 12);
      const gq3 ph = gq_t_apply(T, gq_mk(h.cand_pos[ci * 3], h.cand_pos[ci * 3 + 1], h.cand_pos[ci * 3 + 2]));
      const gq3 nh = gq_t_rot(T, gq_mk(h.cand_nrm[ci * 3], h.cand_nrm[ci * 3 + 1], h.cand_nrm[ci * 3 + 2]));
      if (g.g_cpts) {
        const float* q = g.g_cpts + ((size_t)row * g.n + it) * 3;
        const gq3 gp = gq_mk(q[0], q[1], q[2]);
        acc[0] += gp.x; acc[1] += gp.y; acc[2] += gp.z;
        acc[3] += gp.x * ph.x; acc[4] += gp.x * ph.y; acc[5] += gp.x * ph.z;
        acc[6] += gp.y * ph.x; acc[7] += gp.y * ph.y; acc[8] += gp.y * ph.z;
        acc[9] += gp.z * ph.x; acc[10] += gp.z * ph.y; acc[11] += gp.z * ph.z;
        const gq3 gph = gq_mtv(R, gp);
        f = f + gph;
        m = m + gq_cross(ph, gph);
      }
      if (g.g_cnrm) {
        const float* q = g.g_cnrm + ((size_t)row * g.n + it) * 3;
        const gq3 gn = gq_mk(q[0], q[1], q[2]);
        acc[3] += gn.x * nh.x; acc[4] += gn.x * nh.y; acc[5] += gn.x * nh.z;
        acc[6] += gn.y * nh.x; acc[7] += gn.y * nh.y; acc[8] += gn.y * nh.z;
        acc[9] += gn.z * nh.x; acc[10] += gn.z * nh.y; acc[11] += gn.z * nh.z;
        m = m + gq_cross(nh, gq_mtv(R, gn));
      }
    } else {
      const int sidx = it - n_c;
      l = h.sphere_link[sidx];
      const float* q = g.g_spheres + ((size_t)row * h.S + sidx) * 3;
      const gq3 gp = gq_mk(q[0], q[1], q[2]);
      const GqT T = gq_t_load(sLT + l * 12);
      const gq3 ph = gq_t_apply(T, gq_mk(h.sphere[sidx * 4], h.sphere[sidx * 4 + 1], h.sphere[sidx * 4 + 2]));
      acc[0] += gp.x; acc[1] += gp.y; acc[2] += gp.z;
      acc[3] += gp.x * ph.x; acc[4] += gp.x * ph.y; acc[5] += gp.x * ph.z;
      acc[6] += gp.y * ph.x; acc[7] += gp.y * ph.y; acc[8] += gp.y * ph.z;
      acc[9] += gp.z * ph.x; acc[10] += gp.z * ph.y; acc[11] += gp.z * ph.z;
      const gq3 gph = gq_mtv(R, gp);
      f = gph;
      m = gq_cross(ph, gph);
    }
    if (it < GQ_FK_MAX_ITEMS) {
      sI[it * 6 + 0] = f.x; sI[it * 6 + 1] = f.y; sI[it * 6 + 2] = f.z;
      sI[it * 6 + 3] = m.x; sI[it * 6 + 4] = m.y; sI[it * 6 + 5] = m.z;
      sIn[it] = sLN[l];
    }
  }
  __syncthreads();
  // ---- node accumulators: link wrenches (E_pen) then items, each in index order ---------------------------------------
  float nf[6] = {0, 0, 0, 0, 0, 0};
  if (lane < h.J) {
    if (g.g_wrench) {
#pragma unroll 4
      for (int l = 0; l < h.L; ++l) {
        const float mine = (sLN[l] == lane) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) nf[k] = fmaf(mine, sWr[l * 6 + k], nf[k]);
      }
    }
#pragma unroll 4
    for (int it = 0; it < n_items; ++it) {  // branch-free: every lane reads the (broadcast) record, the owner adds it
      // (a multiply, not a select: the compiler turns a select back into a branch around the LDS reads.  A non-finite
      // record then reaches every node of the row -- harmless, a row with any NaN is zeroed by the proposal anyway)
      const float mine = (sIn[it] == lane) ? 1.0f : 0.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k) nf[k] = fmaf(mine, sI[it * 6 + k], nf[k]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) sNF[lane * 6 + k] = nf[k];
  }
  __syncthreads();
  // ---- fold children into parents, deepest level first ----------------------------------------------------------------
  for (int d = h.max_depth - 1; d >= 0; --d) {
    if (depth == d) {
      for (int k = ch0; k < ch1; ++k) {
        const int c = sCh[k];
#pragma unroll
        for (int q = 0; q < 6; ++q) nf[q] += sNF[c * 6 + q];
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) sNF[lane * 6 + q] = nf[q];
    }
    __syncthreads();
  }
  float* go = g.grad_pose + (size_t)row * g.D;
  float ej = 0.0f;
  float gth = 0.0f;
  if (lane < h.J) {  // d E / d theta_j = axis_w . (m - o x f)  (revolute) | axis_w . f (prismatic)
    const gq3 f = gq_mk(nf[0], nf[1], nf[2]), m = gq_mk(nf[3], nf[4], nf[5]);
    const gq3 aw = gq_t_rot(Wj, axj);
    const gq3 o = gq_t_pos(Wj);
    gth = (ntype == 1) ? gq_dot(aw, m - gq_cross(o, f)) : gq_dot(aw, f);
  }
  if (h.coup) {  // coupled hand: d / d theta_act = C' (d / d theta_tree)  (the jacobian_fnc of the reference's hand files)
    __syncthreads();
    if (lane < h.J) sNF[lane] = gth;
    __syncthreads();
    gth = 0.0f;
    if (lane < h.JA)
      for (int j = 0; j < h.J; ++j) gth = fmaf(h.coup[j * h.JA + lane], sNF[j], gth);
  }
  if (lane < h.JA) {
    if (g.g_theta) gth += g.g_theta[(size_t)row * h.JA + lane];
    if (g.en.total) {  // E_joints = sum relu(theta - hi) + relu(lo - theta)  (energy.py:47-54)
      const float th = en_th, hi = en_jhi, lo = en_jlo;
      if (th > hi) {
        ej += th - hi;
        gth += g.en.w_joints;
      }
      if (th < lo) {
        ej += lo - th;
        gth -= g.en.w_joints;
      }
    }
    go[9 + lane] = gth;
    sG[9 + lane] = gth;
  }
  float e_dis = 0.0f, e_joints = 0.0f, total = 0.0f;
  if (g.en.total) {
    float ed = 0.0f;  // E_dis = sum_i exp(1 + vC_i . nH_i) |d_i|  (energy.py:25-28)
    if (lane < g.en.n) {
      const float root = sqrtf(en_d2 + 1e-8f);
      const float dt = en_sg * (en_on.x * en_nh.x + en_on.y * en_nh.y + en_on.z * en_nh.z);
      ed += expf(1.0f + dt) * root;
    }
    for (int c = lane + GQ_WAVE; c < g.en.n; c += GQ_WAVE) {
      const size_t t = (size_t)row * g.en.n + c;
      const float root = sqrtf(g.en.dist_sq[t] + 1e-8f);
      const float sg = (float)g.en.sign[t];
      const float dt = sg * (g.en.obj_dir[t * 3] * g.en.hand_normals[t * 3] + g.en.obj_dir[t * 3 + 1] * g.en.hand_normals[t * 3 + 1] +
                             g.en.obj_dir[t * 3 + 2] * g.en.hand_normals[t * 3 + 2]);
      ed += expf(1.0f + dt) * root;
    }
    e_dis = gq_dpp_sum(ed);
    e_joints = gq_dpp_sum(ej);
    total = g.en.w_dis * e_dis + g.en.w_fc * en_efc + g.en.w_pen * en_epen + g.en.w_spen * en_espen + g.en.w_joints * e_joints;
    if (lane == 0) {
      g.en.e_dis[row] = e_dis;
      g.en.e_joints[row] = e_joints;
      g.en.total[row] = total;
    }
  }
  // ---- global pose: fixed-tree sums over the lanes ---------------------------------------------------------------------
  float tot[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) tot[i] = acc[i];
  gq_wave_sums_f<12>(tot);  // all twelve lane sums at once (pairwise folding, fixed order)
  if (lane == 0) {
    gq3 gt = gq_mk(tot[0], tot[1], tot[2]);
    float gR[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gR[i] = tot[3 + i] + (g.g_R ? g.g_R[(size_t)row * 9 + i] : 0.0f);
    if (g.g_Rt) {  // grad_t = -R gsum ; grad_R = R K
      float e[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) e[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rt_l), i));
      gt = gt - gq_mv(R, gq_mk(e[0], e[1], e[2]));
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          gR[i * 3 + j] += R[i * 3 + 0] * e[3 + 0 * 3 + j] + R[i * 3 + 1] * e[3 + 1 * 3 + j] + R[i * 3 + 2] * e[3 + 2 * 3 + j];
    }
    go[0] = gt.x; go[1] = gt.y; go[2] = gt.z;
    sG[0] = gt.x; sG[1] = gt.y; sG[2] = gt.z;
    // Gram-Schmidt backward: columns of gR are the gradients of x, y, z
    float pz[9];
#pragma unroll
    for (int i = 3; i < 9; ++i) pz[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pose_l), i));
    const gq3 a = gq_mk(pz[3], pz[4], pz[5]), b = gq_mk(pz[6], pz[7], pz[8]);
    const float na = sqrtf(gq_dot(a, a));
    const gq3 x = (1.0f / na) * a;
    const gq3 yp = b - gq_dot(x, b) * x;
    const float ny = sqrtf(gq_dot(yp, yp));
    const gq3 y = (1.0f / ny) * yp;
    gq3 gx = gq_mk(gR[0], gR[3], gR[6]), gy = gq_mk(gR[1], gR[4], gR[7]);
    const gq3 gz = gq_mk(gR[2], gR[5], gR[8]);
    gx = gx + gq_cross(y, gz);
    gy = gy + gq_cross(gz, x);
    const gq3 gyp = (1.0f / ny) * (gy - gq_dot(gy, y) * y);
    const gq3 gb = gyp - gq_dot(gyp, x) * x;
    gx = gx - gq_dot(x, b) * gyp - gq_dot(gyp, x) * b;
    const gq3 ga = (1.0f / na) * (gx - gq_dot(gx, x) * x);
    go[3] = ga.x; go[4] = ga.y; go[5] = ga.z;
    go[6] = gb.x; go[7] = gb.y; go[8] = gb.z;
    sG[3] = ga.x; sG[4] = ga.y; sG[5] = ga.z;
    sG[6] = gb.x; sG[7] = gb.y; sG[8] = gb.z;
  }
  if (g.has_accept) {  // Metropolis test on the new total, state merge of this row -- from registers / LDS
    __syncthreads();
    const float term = lane == 0 ? e_dis : lane == 1 ? en_efc : lane == 2 ? en_epen : lane == 3 ? en_espen : e_joints;
    gq_accept_finish(g.ac, ap, row, lane, total, sG, term);
  }
}

// ---- self penetration (hand_model.py:989-1040) ----------------------------------------------------------------------
// block = 64 threads = 4 rows x 16 lanes; lane (row, q) scans sphere groups q, q+16, ... (a group = the spheres of
// one link), finds each group's most penetrating pair against all LATER groups, then lane q == 0 folds the groups
// in order (bitwise reproducible) into the energy and the centre gradients.
__global__ __launch_bounds__(64) void gq_self_pen_kernel(gqHand h, const float* __restrict__ centers, int B,
                                                         float gscale, float* __restrict__ e_spen,
                                                         float* __restrict__ g_centers) {
  __shared__ float s_pen[4][64];
  __shared__ int s_a[4][64], s_b[4][64];
  __shared__ float s_c[4][256 * 3];
  const int r4 = threadIdx.x >> 4, q = threadIdx.x & 15;
  const int row = blockIdx.x * 4 + r4;
  const bool ok = row < B;
  for (int i = q; i < h.S * 3; i += 16) s_c[r4][i] = centers[(size_t)(ok ? row : 0) * h.S * 3 + i];
  __syncthreads();
  const float* c = s_c[r4];
  const int ng = h.NG - 1;  // the last group has nothing after it
  for (int gi = q; gi < ng && gi < 64; gi += 16) {
    const int a0 = h.group_off[gi], a1 = h.group_off[gi + 1];
    float best = GQ_INF_F;
    int ba = -1, bb = -1;
    for (int a = a0; a < a1; ++a) {
      const gq3 pa = gq_mk(c[a * 3], c[a * 3 + 1], c[a * 3 + 2]);
      const float ra = h.sphere[a * 4 + 3];
      for (int b = a1; b < h.S; ++b) {
        const gq3 d = gq_mk(pa.x - c[b * 3] + 1e-13f, pa.y - c[b * 3 + 1] + 1e-13f, pa.z - c[b * 3 + 2] + 1e-13f);
        const float pen = sqrtf(gq_dot(d, d)) - (ra + h.sphere[b * 4 + 3]);
        if (pen < best) {
          best = pen;
          ba = a;
          bb = b;
        }
      }
    }
    s_pen[r4][gi] = best;
    s_a[r4][gi] = ba;
    s_b[r4][gi] = bb;
  }
  __syncthreads();
  if (!ok) return;
  float* gc = g_centers + (size_t)row * h.S * 3;
  for (int i = q; i < h.S * 3; i += 16) gc[i] = 0.0f;
  __syncthreads();
  if (q != 0) return;
  float e = 0.0f;
  for (int gi = 0; gi < ng && gi < 64; ++gi) {
    const float best = s_pen[r4][gi];
    const int ba = s_a[r4][gi], bb = s_b[r4][gi];
    if (best < 0.0f && ba >= 0) {
      e -= best;
      const gq3 d = gq_mk(c[ba * 3] - c[bb * 3] + 1e-13f, c[ba * 3 + 1] - c[bb * 3 + 1] + 1e-13f,
                          c[ba * 3 + 2] - c[bb * 3 + 2] + 1e-13f);
      const float inv = 1.0f / sqrtf(gq_dot(d, d));
      // E += -|a-b| + ... : dE/da = -(a-b)/|a-b|, dE/db = +(a-b)/|a-b|
      const float sc = gscale * inv;
      gc[ba * 3] -= d.x * sc; gc[ba * 3 + 1] -= d.y * sc; gc[ba * 3 + 2] -= d.z * sc;
      gc[bb * 3] += d.x * sc; gc[bb * 3 + 1] += d.y * sc; gc[bb * 3 + 2] += d.z * sc;
    }
  }
  e_spen[row] = e;
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <typename T>
static int gq_upload(T** dst, const T* src, size_t n) {
  *dst = nullptr;
  if (n == 0) return GQ_OK;
  GQ_CHECK_HIP(hipMalloc((void**)dst, n * sizeof(T)));
  GQ_CHECK_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return GQ_OK;
}

int gq_colsq_launch_(const float* grad, int B, int D, int clip, float* g2, void* stream);  // loop.hip
int gq_sdf_wave_args_(const gqMeshSet* ms, int64_t n_points, int64_t queries_per_mesh, float* dist_sq, int32_t* sign,
                      float* normal, float* closest, GqWaveArgs* out);  // sdf.hip

extern "C" {

int gq_hand_create(const gqHandDesc* d, gqHand** out) {
  GQ_REQUIRE(d && out, "hand_create: null");
  GQ_REQUIRE(d->n_dofs > 0 && d->n_dofs <= 64 && d->n_links > 0 && d->n_links <= 64 && d->n_cand >= 0 &&
                 d->n_spheres >= 0 && d->n_spheres <= 256,
             "hand_create: bad sizes J=%d L=%d C=%d S=%d", d->n_dofs, d->n_links, d->n_cand, d->n_spheres);
  for (int j = 0; j < d->n_dofs; ++j)
    GQ_REQUIRE(d->node_parent[j] < j && d->node_parent[j] >= -1, "hand_create: node_parent must be topologically sorted");
  for (int l = 0; l < d->n_links; ++l)
    GQ_REQUIRE(d->link_node[l] >= -1 && d->link_node[l] < d->n_dofs, "hand_create: link_node out of range");
  for (int c = 0; c < d->n_cand; ++c)
    GQ_REQUIRE(d->cand_link[c] >= 0 && d->cand_link[c] < d->n_links, "hand_create: cand_link out of range");
  for (int s = 0; s < d->n_spheres; ++s) {
    GQ_REQUIRE(d->sphere_link[s] >= 0 && d->sphere_link[s] < d->n_links, "hand_create: sphere_link out of range");
    GQ_REQUIRE(s == 0 || d->sphere_link[s] >= d->sphere_link[s - 1], "hand_create: sphere_link must be sorted");
  }
  const int JA = (d->n_actuated > 0 && d->coupling) ? d->n_actuated : d->n_dofs;
  GQ_REQUIRE(JA > 0 && JA <= 64 && (d->n_actuated <= 0 || d->coupling), "hand_create: n_actuated without a coupling matrix");
  gqHand* h = new gqHand();
  h->J = d->n_dofs;
  h->JA = JA;
  h->coup = nullptr;
  h->coup0 = nullptr;
  h->L = d->n_links;
  h->C = d->n_cand;
  h->S = d->n_spheres;
  int32_t groups[258];
  int ng = 0;
  for (int s = 0; s < d->n_spheres && ng < 64; ++s)
    if (s == 0 || d->sphere_link[s] != d->sphere_link[s - 1]) groups[ng++] = s;
  groups[ng] = d->n_spheres;
  h->NG = ng;
  int rc = 0;
  rc |= gq_upload(&h->node_parent, d->node_parent, h->J);
  rc |= gq_upload(&h->node_type, d->node_type, h->J);
  rc |= gq_upload(&h->node_pre, d->node_pre, (size_t)h->J * 12);
  rc |= gq_upload(&h->node_axis, d->node_axis, (size_t)h->J * 3);
  rc |= gq_upload(&h->link_node, d->link_node, h->L);
  rc |= gq_upload(&h->link_offset, d->link_offset, (size_t)h->L * 12);
  rc |= gq_upload(&h->cand_pos, d->cand_pos, (size_t)h->C * 3);
  rc |= gq_upload(&h->cand_nrm, d->cand_nrm, (size_t)h->C * 3);
  rc |= gq_upload(&h->cand_link, d->cand_link, h->C);
  rc |= gq_upload(&h->sphere, d->sphere, (size_t)h->S * 4);
  rc |= gq_upload(&h->sphere_link, d->sphere_link, h->S);
  rc |= gq_upload(&h->jlo, d->joints_lower, h->JA);
  rc |= gq_upload(&h->jhi, d->joints_upper, h->JA);
  if (d->n_actuated > 0 && d->coupling) {
    float zero[64] = {0};
    rc |= gq_upload(&h->coup, d->coupling, (size_t)h->J * h->JA);
    rc |= gq_upload(&h->coup0, d->coupling_offset ? d->coupling_offset : (const float*)zero, (size_t)h->J);
  }
  rc |= gq_upload(&h->group_off, (const int32_t*)groups, (size_t)ng + 1);
  {
    int32_t grp[256];
    int gcur = 0;
    for (int s = 0; s < d->n_spheres; ++s) {
      while (gcur + 1 < ng && s >= groups[gcur + 1]) ++gcur;
      grp[s] = gcur;
    }
    rc |= gq_upload(&h->sphere_grp, (const int32_t*)grp, (size_t)h->S);
  }
  {
    int32_t depth[64], coff[65], cidx[64];
    int md = 0;
    for (int j = 0; j < h->J; ++j) {
      depth[j] = d->node_parent[j] < 0 ? 0 : depth[d->node_parent[j]] + 1;
      md = depth[j] > md ? depth[j] : md;
    }
    int k = 0;
    for (int p = 0; p < h->J; ++p) {
      coff[p] = k;
      for (int j = 0; j < h->J; ++j)
        if (d->node_parent[j] == p) cidx[k++] = j;
    }
    coff[h->J] = k;
    h->max_depth = md;
    rc |= gq_upload(&h->node_depth, (const int32_t*)depth, h->J);
    rc |= gq_upload(&h->child_off, (const int32_t*)coff, (size_t)h->J + 1);
    rc |= gq_upload(&h->child_idx, (const int32_t*)cidx, (size_t)(k > 0 ? k : 1));
  }
  if (rc) return GQ_ERR_HIP;
  *out = h;
  return GQ_OK;
}

int gq_hand_destroy(gqHand* h) {
  if (!h) return GQ_OK;
  void* p[] = {h->node_parent, h->node_type, h->node_pre, h->node_axis, h->link_node, h->link_offset, h->cand_pos,
               h->cand_nrm, h->cand_link, h->sphere, h->sphere_link, h->jlo, h->jhi, h->group_off, h->node_depth,
               h->child_off, h->child_idx, h->sphere_grp, h->coup, h->coup0};
  for (void* q : p)
    if (q) (void)hipFree(q);
  delete h;
  return GQ_OK;
}

// workspace: node_W (B,J,12) floats [kept for backward] + node_F (B,J,6) floats
int gq_fk_workspace_bytes(const gqHand* h, int64_t batch, size_t* bytes) {
  GQ_REQUIRE(h && bytes && batch >= 0, "fk_workspace_bytes: bad arguments");
  *bytes = (size_t)batch * h->J * 18 * sizeof(float) + 256;
  return GQ_OK;
}

int gq_fk_forward(const gqHand* h, const float* hand_pose, const int64_t* contact_idx, int64_t batch, int n_contact,
                  float* Rg, float* link_T, float* contact_points, float* contact_normals, float* sphere_centers,
                  float spen_scale, float* e_spen, float* g_sphere_centers, const gqProposeDesc* propose,
                  const gqSdfDesc* sdf, void* workspace, size_t workspace_bytes, void* stream) {
  GQ_REQUIRE(h && hand_pose && Rg && link_T && workspace, "fk_forward: null pointer");
  GQ_REQUIRE(!e_spen || (sphere_centers && g_sphere_centers && h->S > 0 && h->S <= 256),
             "fk_forward: the fused self-penetration term needs sphere_centers and g_sphere_centers");
  GQ_REQUIRE(batch > 0 && n_contact >= 0, "fk_forward: bad sizes");
  GQ_REQUIRE(n_contact == 0 || (contact_idx && contact_points && contact_normals), "fk_forward: null contact buffers");
  GQ_REQUIRE(workspace_bytes >= (size_t)batch * h->J * 18 * sizeof(float), "fk_forward: workspace too small");
  GqFkArgs a{};
  a.h = *h;
  a.hand_pose = hand_pose;
  a.idx = contact_idx;
  a.B = (int)batch;
  a.n = n_contact;
  a.D = 9 + h->JA;
  a.Rg = Rg;
  a.link_T = link_T;
  a.node_W = (float*)workspace;
  a.cpts = contact_points;
  a.cnrm = contact_normals;
  a.spheres = sphere_centers;
  a.e_spen = e_spen;
  a.g_spheres = g_sphere_centers;
  a.spen_scale = spen_scale;
  if (propose) {
    const gqProposeDesc& p = *propose;
    GQ_REQUIRE(p.hand_pose && p.grad && p.contact_idx && p.u_switch && p.new_idx && p.ema && p.step && p.slot_ctr &&
                   p.slots > 0 && p.stepsize_period > 0 && contact_idx && a.D <= 128,
               "fk_forward: incomplete gqProposeDesc");
    GQ_REQUIRE(p.g2_scratch, "fk_forward: gqProposeDesc.g2_scratch (D floats) is missing");
    GQ_REQUIRE(!p.energy || (p.z_out && p.batch_each > 1), "fk_forward: z-score needs z_out and batch_each > 1");
    a.has_propose = 1;
    a.pr.hand_pose = p.hand_pose;
    a.pr.grad = p.grad;
    a.pr.g2 = p.g2_scratch;
    a.pr.idx = p.contact_idx;
    a.pr.u_switch = p.u_switch;
    a.pr.new_idx = p.new_idx;
    a.pr.B = (int)batch;
    a.pr.D = a.D;
    a.pr.n = n_contact;
    a.pr.clip = p.clip_grad;
    a.pr.step_size = p.step_size;
    a.pr.decay = p.decay;
    a.pr.mu = p.mu;
    a.pr.switch_p = p.switch_possibility;
    a.pr.stepsize_period = p.stepsize_period;
    a.pr.ema = p.ema;
    a.pr.step = p.step;
    a.pr.pose_out = const_cast<float*>(hand_pose);
    a.pr.idx_out = const_cast<int64_t*>(contact_idx);
    a.pr.s_out = p.step_size_out;
    a.pr.energy = p.energy;
    a.pr.batch_each = (int)p.batch_each;
    a.pr.z_out = p.z_out;
    a.pr.slot_ctr = p.slot_ctr;
    a.pr.slots = p.slots;
  }
  int nw = 1;
  if (sdf) {  // the contact queries of a row are answered by the row's block: up to 12 query wavefronts
    GQ_REQUIRE(n_contact > 0 && contact_points, "fk_forward: gqSdfDesc needs contact points");
    int rc = gq_sdf_wave_args_(sdf->meshes, batch * n_contact, sdf->queries_per_mesh, sdf->dist_sq, sdf->sign,
                               sdf->obj_dir, sdf->closest, &a.sdf);
    if (rc) return rc;
    a.has_sdf = 1;
    const int rounds = (n_contact + 11) / 12;
    nw = (n_contact + rounds - 1) / rounds;
  }
  if (propose) {
    // the RMS mean couples all rows (optimizer.py:231).  Small batches with query wavefronts in the block: every block
    // reduces it redundantly while its kinematics wavefront fetches the hand tables (no launch); otherwise a small
    // launch of its own, then everything per row
    a.pr.g2_inline = (nw >= 3 && a.D <= GQ_WAVE && batch <= 512) ? 1 : 0;
    if (!a.pr.g2_inline) {
      int rc = gq_colsq_launch_(propose->grad, (int)batch, a.D, propose->clip_grad, propose->g2_scratch, stream);
      if (rc) return rc;
    }
  }
  if (!a.has_sdf)
    hipLaunchKernelGGL(gq_fk_forward_row_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(gq_fk_forward_kernel, dim3((unsigned)batch), dim3(GQ_WAVE * nw), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_fk_backward(const gqHand* h, const float* hand_pose, const int64_t* contact_idx, int64_t batch, int n_contact,
                   const float* Rg, const float* link_T, const float* g_contact_points, const float* g_contact_normals,
                   const float* g_sphere_centers, const float* g_link_wrench, const float* g_Rt, const float* g_theta,
                   const float* g_R, float* grad_pose, const gqRowEnergyDesc* energy, const gqAcceptDesc* accept,
                   void* workspace, size_t workspace_bytes, void* stream) {
  GQ_REQUIRE(h && hand_pose && Rg && link_T && grad_pose && workspace, "fk_backward: null pointer");
  if (energy) {
    const gqRowEnergyDesc& e = *energy;
    GQ_REQUIRE(e.dist_sq && e.sign && e.obj_dir && e.hand_normals && e.joints_lower && e.joints_upper && e.e_fc &&
                   e.e_pen && e.e_spen && e.e_dis && e.e_joints && e.total && e.n >= 0,
               "fk_backward: incomplete gqRowEnergyDesc");
  }
  GQ_REQUIRE(batch > 0 && n_contact >= 0, "fk_backward: bad sizes");
  GQ_REQUIRE(workspace_bytes >= (size_t)batch * h->J * 18 * sizeof(float), "fk_backward: workspace too small");
  GqFkBwdArgs a{};
  a.h = *h;
  a.hand_pose = hand_pose;
  a.idx = contact_idx;
  a.Rg = Rg;
  a.link_T = link_T;
  a.node_W = (const float*)workspace;
  a.g_cpts = g_contact_points;
  a.g_cnrm = g_contact_normals;
  a.g_spheres = g_sphere_centers;
  a.g_wrench = g_link_wrench;
  a.g_Rt = g_Rt;
  a.g_theta = g_theta;
  a.g_R = g_R;
  a.B = (int)batch;
  a.n = (g_contact_points || g_contact_normals) ? n_contact : 0;
  a.D = 9 + h->JA;
  a.node_F = (float*)workspace + (size_t)batch * h->J * 12;
  a.grad_pose = grad_pose;
  if (energy) a.en = *energy;
  if (accept) {
    const gqAcceptDesc& c = *accept;
    GQ_REQUIRE(energy && contact_idx, "fk_backward: the fused accept step needs the gqRowEnergyDesc and contact_idx");
    GQ_REQUIRE(c.u_accept && c.step && c.energy && c.pose && c.idx && c.grad && c.accept && c.slot_ctr && c.slots > 0 &&
                   c.annealing_period > 0 && (c.n_terms == 0 || (c.terms_new && c.terms)),
               "fk_backward: incomplete gqAcceptDesc");
    a.has_accept = 1;
    a.ac.new_energy = energy->total;
    a.ac.u_accept = c.u_accept;
    a.ac.z = c.z;
    a.ac.reset_mask = c.reset_mask;
    a.ac.step = c.step;
    a.ac.pose_new = hand_pose;
    a.ac.idx_new = contact_idx;
    a.ac.grad_new = grad_pose;
    a.ac.B = (int)batch;
    a.ac.D = 9 + h->JA;
    a.ac.n = n_contact;
    a.ac.T0 = c.starting_temperature;
    a.ac.decay = c.decay;
    a.ac.annealing_period = c.annealing_period;
    a.ac.energy = c.energy;
    a.ac.pose = c.pose;
    a.ac.idx = c.idx;
    a.ac.grad = c.grad;
    a.ac.accept = c.accept;
    a.ac.temperature = c.temperature;
    a.ac.n_terms = c.n_terms;
    a.ac.terms_new = c.terms_new;
    a.ac.terms = c.terms;
    a.ac.slot_ctr = c.slot_ctr;
    a.ac.slots = c.slots;
  }
  GQ_REQUIRE(a.n + h->S <= GQ_FK_MAX_ITEMS, "fk_backward: n_contact + n_spheres = %d exceeds %d", a.n + h->S,
             GQ_FK_MAX_ITEMS);
  hipLaunchKernelGGL(gq_fk_backward_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_self_pen_forward(const gqHand* h, const float* sphere_centers, int64_t batch, float grad_scale, float* e_spen,
                        float* g_centers, void* stream) {
  GQ_REQUIRE(h && sphere_centers && e_spen && g_centers && batch > 0, "self_pen_forward: bad arguments");
  hipLaunchKernelGGL(gq_self_pen_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(64), 0, (hipStream_t)stream, *h,
                     sphere_centers, (int)batch, grad_scale, e_spen, g_centers);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
