// Two "stage" kernels that run the force-closure branch and the hand-penetration branch of one MALA* iteration side by
// side in the SAME launch (the branches are independent until the FK backward, scripts/fit.py:434-438):
//
//   stage A   blocks [0, B/4)        fc head   contact terms + cone matrix + all QP iterations; 4 rows per block,
//                                              one wavefront (SIMD) each
//             blocks [B/4, .. + gx B) pen query penetration-only hand query of 256 surface points of one row
//   stage B   blocks [0, B)          fc tail   stop rule + E_fc + QP backward + contact gradient (wave 0 only)
//             blocks [B, 2B)         pen bwd   link wrenches + E_pen of one row
//             blocks [2B, 2B + B/4)  spheres   world sphere centres + self penetration (optional; 4 rows per block)
//
// With B = 256 rows neither branch fills 256 CUs on its own (the QP is one wavefront per row), and separate streams
// cost 20..100 us of cross-queue dependency latency per iteration on this platform; putting both roles in one grid gives
// the overlap without any inter-queue hand-shake.  The bodies are the ones of fcstep.hip / sdf.hip (same arithmetic),
// the long-running fc blocks come first so that they are resident from the start.
#include "fcstep_dev.h"
#include "kin_dev.h"
#include "metric_dev.h"
#include "pen_dev.h"

int gq_qp_stop_launch_(const float* resid, const float* mu, int B, int max_iter, float eps, int not_improved_lim,
                       float* runmin, int* kstar, int32_t* n_iter, void* stream);
int gq_pen_points_per_thread_();  // sdf.hip: surface points per thread of the penetration query (gq_debug_set_pen_ppt)

// STOP: the large-batch stop-rule epilogue is compiled in (its 70 extra registers would cost the small-batch
// instantiation one wavefront per SIMD: 166 instead of 127 VGPRs)
template <int NC, bool STOP, int PPT>
__global__ __launch_bounds__(256, NC == 1 ? 4 : 2) void gq_stage_a_kernel(GqFcStepArgs f, GqPenArgs p, int gx, int nfc) {
  extern __shared__ char gq_lds[];
  const int b = (int)blockIdx.x;
  if (b < nfc) {  // four rows per block, one wavefront (= one SIMD) each: the fc role occupies B/4 CUs only
    const int wv = (int)threadIdx.x / GQ_WAVE, row = b * GQ_HEAD_ROWS + wv;
    if (wv >= GQ_HEAD_ROWS || row >= f.B) return;
    // the fc rows are the critical path of this launch (one long dependent instruction stream per wavefront); the query
    // wavefronts that share their SIMDs mostly wait for memory -- let the arbiter prefer the fc wavefront when both are ready
    __builtin_amdgcn_s_setprio(3);
#ifdef GQ_BLOCK_TIMES  // where the fc rows run: one record per fc block behind those of the query blocks
    if (threadIdx.x == 0 && p.span) {
      uint64_t* rec = p.span + 128 + 8 * (size_t)(gridDim.x - nfc + b);
      rec[0] = __builtin_amdgcn_s_memrealtime();
      rec[6] = gq_hw_id();
    }
#endif
    float hr, hm;
    gq_fc_head_body<NC>(f, row, reinterpret_cast<float*>(gq_lds) + wv * f.n * 6, &hr, &hm);
    if (STOP) {  // large batches: the stop rule as epilogue of the last head block (no stop launch)
      const int nrow = f.B - b * GQ_HEAD_ROWS < GQ_HEAD_ROWS ? f.B - b * GQ_HEAD_ROWS : GQ_HEAD_ROWS;
      gq_fc_head_epilogue(f, b, wv, nrow, hr, hm, reinterpret_cast<unsigned*>(gq_lds) + GQ_HEAD_ROWS * f.n * 6);
    }
#ifdef GQ_BLOCK_TIMES
    if (threadIdx.x == 0 && p.span) p.span[128 + 8 * (size_t)(gridDim.x - nfc + b) + 1] = __builtin_amdgcn_s_memrealtime();
#endif
  } else if (gx == 0) {  // link-driven query: one block per row (gqPenStepDesc.grid)
    gq_pen_cells_body(p, b - nfc, gq_lds);
  } else {
    // slice-major: the blocks that are dispatched last -- the ones that have to wait for a free slot (four blocks per CU
    // are resident, B/4 of the slots hold the fc rows) and so decide when the role ends -- are those of the last, partly
    // filled slice of every row (2500 = 9 x 256 + 196 points) instead of all slices of the last rows
    const int q = b - nfc;
    gq_pen_grid_body<true, GQ_PG_ECAP, GQ_PG_ICAP, PPT>(p, q / p.B, q % p.B, gq_lds);
  }
}

struct GqSpenRole {  // third role of stage B: sphere centres + self penetration of 4 rows per block (0 blocks = absent)
  gqHand h;
  GqSpenArgs sa;
  const float* Rg;
  const float* hand_pose;
  const float* link_T;
  int D;
};

// block = 4 rows, one wavefront each; LDS per wavefront: 512 B of keys + S x 16 B
__device__ __forceinline__ void gq_spen_role_body(const GqSpenRole& sp, int blk, int B, char* lds) {
  const int wv = (int)threadIdx.x / GQ_WAVE, row = blk * 4 + wv;
  if (row >= B) return;
  const int S = sp.h.S;
  char* base = lds + (size_t)wv * ((size_t)S * 16 + 512);
  unsigned long long* sKey = reinterpret_cast<unsigned long long*>(base);
  float* sC = reinterpret_cast<float*>(base + 512);
  float* sRad = sC + 3 * S;
  const float* hp = sp.hand_pose + (size_t)row * sp.D;
  gq_spheres_row(sp.h, sp.sa, sp.link_T + (size_t)row * sp.h.L * 12, sp.Rg + (size_t)row * 9, gq_mk(hp[0], hp[1], hp[2]),
                 row, gq_lane(), sC, sKey, sRad);
}
// the same role as a launch of its own (large batches: it rides on the penetration branch of the graph instead of
// lengthening the FK forward launch that both branches wait for)
__global__ __launch_bounds__(256) void gq_spheres_kernel(GqSpenRole sp, int B) {
  extern __shared__ char gq_lds[];
  gq_spen_role_body(sp, (int)blockIdx.x, B, gq_lds);
}

template <int NC, int RPL>
__global__ __launch_bounds__(256) void gq_stage_b_kernel(GqFcStepArgs f, GqPenBwdArgs p, GqSpenRole sp) {
  extern __shared__ char gq_lds[];
  const int b = (int)blockIdx.x;
  if (b >= 2 * f.B) {
    gq_spen_role_body(sp, b - 2 * f.B, f.B, gq_lds);
    return;
  }
  if (b < f.B) {  // one row per block here: 2B blocks = two per CU at B = 256, every tail wavefront has a CU's L1 to itself
    if (threadIdx.x >= GQ_WAVE) return;
    __builtin_amdgcn_s_setprio(3);  // the longest role of this launch: one dependent instruction stream per row
    gq_fc_tail_body<NC, RPL>(f, b, reinterpret_cast<float*>(gq_lds));
  } else {
    gq_pen_bwd_body(p, b - f.B, gq_lds);
  }
}

// ---- the same two launches for the reference's other force-closure energies (scripts/fit.py:343-347) ---------------------
//   stage A'  blocks [0, nalt)        contact terms of E_dis + dexgrasp (4 rows per block, one wavefront each) or TDG (one row
//                                     per block, 256 threads) energy with its analytic contact-point gradient
//             the other gx B blocks   penetration query, as in stage A
//   stage B'  blocks [0, B)           penetration backward;  [B, B + B/4) sphere centres + self penetration
struct GqAltArgs {
  const float* dist_sq; const int32_t* sign; const float* onrm; const float* closest;
  const float* cpts; const float* cnrm;  // contact points, HAND normals
  float w_dis;
  float* obj_normal; float* g_cpts; float* g_cnrm;
  int B, n;
  GqDexArgs dex;
  GqTdgArgs tdg;
};

// contact terms of one row by threads t0 .. : global outputs + LDS copies (contact points, outward object normals) for the
// energy body that follows in the same block
__device__ __forceinline__ void gq_alt_contact_terms(const GqAltArgs& a, int row, int t0, int stride, float* s_cp, float* s_on) {
  for (int c = t0; c < a.n; c += stride) {
    const size_t t = (size_t)row * a.n + c;
    const gq3 on = gq_mk(a.onrm[t * 3], a.onrm[t * 3 + 1], a.onrm[t * 3 + 2]);
    const gq3 nH = gq_mk(a.cnrm[t * 3], a.cnrm[t * 3 + 1], a.cnrm[t * 3 + 2]);
    const gq3 p = gq_mk(a.cpts[t * 3], a.cpts[t * 3 + 1], a.cpts[t * 3 + 2]);
    const gq3 cl = gq_mk(a.closest[t * 3], a.closest[t * 3 + 1], a.closest[t * 3 + 2]);
    const GqContactTerm ct = gq_contact_term(a.dist_sq[t], (float)a.sign[t], on, nH, p, cl, a.w_dis);
    a.obj_normal[t * 3] = ct.vC.x; a.obj_normal[t * 3 + 1] = ct.vC.y; a.obj_normal[t * 3 + 2] = ct.vC.z;
    a.g_cpts[t * 3] = ct.g_p.x; a.g_cpts[t * 3 + 1] = ct.g_p.y; a.g_cpts[t * 3 + 2] = ct.g_p.z;
    a.g_cnrm[t * 3] = ct.g_n.x; a.g_cnrm[t * 3 + 1] = ct.g_n.y; a.g_cnrm[t * 3 + 2] = ct.g_n.z;
    s_cp[c * 3] = p.x; s_cp[c * 3 + 1] = p.y; s_cp[c * 3 + 2] = p.z;
    s_on[c * 3] = ct.vC.x; s_on[c * 3 + 1] = ct.vC.y; s_on[c * 3 + 2] = ct.vC.z;
  }
}

template <int ENERGY, int PPT>  // 1 dexgrasp, 2 tdg
__global__ __launch_bounds__(256, 4) void gq_stage_alt_kernel(GqAltArgs a, GqPenArgs p, int gx, int nalt) {
  extern __shared__ char gq_lds[];
  const int b = (int)blockIdx.x;
  if (b < nalt) {
    float* sh = reinterpret_cast<float*>(gq_lds);
    if (ENERGY == 1) {
      const int wv = (int)threadIdx.x / GQ_WAVE, row = b * 4 + wv;
      if (row >= a.B) return;
      float* s_cp = sh + (size_t)wv * a.n * 6;
      float* s_on = s_cp + a.n * 3;
      gq_alt_contact_terms(a, row, gq_lane(), GQ_WAVE, s_cp, s_on);
      gq_wave_sync();
      __threadfence_block();  // the body accumulates into the g_cpts this wavefront has just stored
      gq_dexgrasp_body(a.dex, row, gq_lane(), s_cp, s_on);
    } else {
      const int row = b;
      float* s_cp = sh;
      float* s_on = sh + a.n * 3;
      gq_alt_contact_terms(a, row, (int)threadIdx.x, 256, s_cp, s_on);
      __syncthreads();
      gq_tdg_body(a.tdg, row, sh + (size_t)a.n * 6, s_cp, s_on);
    }
  } else if (gx == 0) {
    gq_pen_cells_body(p, b - nalt, gq_lds);
  } else {
    const int q = b - nalt;
    gq_pen_grid_body<true, GQ_PG_ECAP, GQ_PG_ICAP, PPT>(p, q / p.B, q % p.B, gq_lds);
  }
}

__global__ __launch_bounds__(256) void gq_stage_b_alt_kernel(GqPenBwdArgs p, GqSpenRole sp, int B) {
  extern __shared__ char gq_lds[];
  const int b = (int)blockIdx.x;
  if (b >= B) gq_spen_role_body(sp, b - B, B, gq_lds);
  else gq_pen_bwd_body(p, b, gq_lds);
}

static int gq_spen_role_fill(const gqPenStepDesc* pen, GqSpenRole* sp, int B, int* n_sp) {
  *n_sp = 0;
  if (!pen->hand) return GQ_OK;
  GQ_REQUIRE(pen->e_spen && pen->g_sphere_centers && pen->hand->S > 0 && pen->hand->S <= 256,
             "pen step: the self-penetration role needs e_spen, g_sphere_centers and 1..256 spheres");
  sp->h = *pen->hand;
  sp->sa.spheres = pen->sphere_centers;
  sp->sa.e_spen = pen->e_spen;
  sp->sa.g_spheres = pen->g_sphere_centers;
  sp->sa.spen_scale = pen->w_spen;
  sp->Rg = pen->Rg;
  sp->hand_pose = pen->hand_pose;
  sp->link_T = pen->link_T;
  sp->D = pen->pose_dim;
  *n_sp = (B + 3) / 4;
  return GQ_OK;
}

extern "C" {

int gq_alt_pen_step(const gqAltFcDesc* alt, const gqPenStepDesc* pen, void* stream) {
  GQ_REQUIRE(alt && pen, "alt_pen_step: null descriptor");
  hipStream_t st = (hipStream_t)stream;
  GQ_REQUIRE(alt->dist_sq && alt->sign && alt->obj_dir && alt->closest && alt->contact_pts && alt->hand_normals && alt->cog &&
                 alt->obj_normal && alt->g_contact_pts && alt->g_hand_normals && alt->e_fc && alt->batch > 0 &&
                 alt->n_contact > 0 && alt->n_contact <= 256, "alt_pen_step: bad arguments");
  GQ_REQUIRE(alt->energy == 1 || (alt->energy == 2 && alt->directions && alt->n_directions > 0 && alt->friction > 0.0f &&
                                  alt->friction <= 1.0f && alt->obb_length > 0.0f),
             "alt_pen_step: energy must be 1 (dexgrasp) or 2 (tdg with directions)");
  GqAltArgs a{};
  a.dist_sq = alt->dist_sq; a.sign = alt->sign; a.onrm = alt->obj_dir; a.closest = alt->closest;
  a.cpts = alt->contact_pts; a.cnrm = alt->hand_normals; a.w_dis = alt->w_dis;
  a.obj_normal = alt->obj_normal; a.g_cpts = alt->g_contact_pts; a.g_cnrm = alt->g_hand_normals;
  a.B = (int)alt->batch; a.n = alt->n_contact;
  a.dex.cpts = alt->contact_pts; a.dex.cnrm = alt->obj_normal; a.dex.cog = alt->cog;
  a.dex.B = a.B; a.dex.n = a.n; a.dex.tw = alt->torque_weight; a.dex.grad_e = nullptr; a.dex.w = alt->w_fc;
  a.dex.accumulate = 1; a.dex.e = alt->e_fc; a.dex.g_cpts = alt->g_contact_pts;
  a.tdg.cpts = alt->contact_pts; a.tdg.cnrm = alt->obj_normal; a.tdg.cog = alt->cog; a.tdg.dirs = alt->directions;
  a.tdg.B = a.B; a.tdg.n = a.n; a.tdg.P = alt->n_directions; a.tdg.miu = alt->friction;
  a.tdg.inv_obb = alt->obb_length > 0.0f ? 1.0f / alt->obb_length : 0.0f; a.tdg.scale = alt->scale;
  a.tdg.density = alt->enable_density; a.tdg.grad_e = nullptr; a.tdg.w = alt->w_fc; a.tdg.accumulate = 1;
  a.tdg.e = alt->e_fc; a.tdg.g_cpts = alt->g_contact_pts;
  GqPenArgs p{};
  int rc = gq_pen_fill(pen->links, pen->surface_points, pen->n_obj, pen->n_surface, pen->batch_each, pen->hand_pose,
                       pen->pose_dim, pen->Rg, pen->link_T, pen->dis, pen->link, pen->gvec, pen->span, &p, pen->grid);
  if (rc) return rc;
  p.patch = pen->patch_spheres;
  GQ_REQUIRE(p.occ && p.cand_off, "alt_pen_step: the link mesh set has no voxel candidate lists (gq_meshset_build_occupancy)");
  GQ_REQUIRE(p.B == a.B, "alt_pen_step: the two descriptors disagree on the batch (%d vs %d)", p.B, a.B);
  GqPenBwdArgs pb{};
  rc = gq_pen_bwd_fill(p.L, pen->surface_points, pen->n_obj, pen->n_surface, pen->batch_each, pen->hand_pose, pen->pose_dim,
                       pen->Rg, nullptr, pen->link, pen->gvec, pen->link_wrench, pen->gRt, pen->dis, pen->w_pen, pen->e_pen,
                       pen->span, pen->span_acc, &pb);
  if (rc) return rc;
  const int ppt = gq_pen_points_per_thread_();
  const int gx = pen->grid ? 0 : (p.P + 256 * ppt - 1) / (256 * ppt);
  GqSpenRole sp{};
  int n_sp = 0;
  rc = gq_spen_role_fill(pen, &sp, a.B, &n_sp);
  if (rc) return rc;
  const int nalt = alt->energy == 1 ? (a.B + 3) / 4 : a.B;
  const size_t lds_alt = alt->energy == 1 ? (size_t)4 * a.n * 6 * sizeof(float)
                                          : ((size_t)a.n * 6 + gq_tdg_lds_floats(a.n)) * sizeof(float);
  const size_t lds_a = std::max(pen->grid ? gq_pen_cells_lds_bytes(p.L, p.P) : gq_pen_grid_lds_bytes(p.L, GQ_PG_ECAP, GQ_PG_ICAP, ppt),
                                lds_alt);
  const size_t lds_b = std::max(gq_pen_bwd_lds_bytes(), n_sp ? (size_t)4 * ((size_t)sp.h.S * 16 + 512) : (size_t)0);
  const dim3 grid_a((unsigned)(nalt + (pen->grid ? 1 : gx) * p.B)), grid_b((unsigned)(a.B + n_sp)), block(256);
  if (alt->energy == 1) {
    if (ppt == 2) hipLaunchKernelGGL((gq_stage_alt_kernel<1, 2>), grid_a, block, lds_a, st, a, p, gx, nalt);
    else hipLaunchKernelGGL((gq_stage_alt_kernel<1, 1>), grid_a, block, lds_a, st, a, p, gx, nalt);
  } else {
    if (ppt == 2) hipLaunchKernelGGL((gq_stage_alt_kernel<2, 2>), grid_a, block, lds_a, st, a, p, gx, nalt);
    else hipLaunchKernelGGL((gq_stage_alt_kernel<2, 1>), grid_a, block, lds_a, st, a, p, gx, nalt);
  }
  GQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(gq_stage_b_alt_kernel, grid_b, block, lds_b, st, pb, sp, a.B);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_spheres_self_pen(const gqHand* h, const float* hand_pose, int pose_dim, const float* Rg, const float* link_T,
                        int64_t batch, float w_spen, float* sphere_centers, float* e_spen, float* g_sphere_centers,
                        void* stream) {
  GQ_REQUIRE(h && hand_pose && Rg && link_T && e_spen && g_sphere_centers && batch > 0 && h->S > 0 && h->S <= 256,
             "spheres_self_pen: bad arguments");
  GqSpenRole sp{};
  sp.h = *h;
  sp.sa.spheres = sphere_centers;
  sp.sa.e_spen = e_spen;
  sp.sa.g_spheres = g_sphere_centers;
  sp.sa.spen_scale = w_spen;
  sp.Rg = Rg;
  sp.hand_pose = hand_pose;
  sp.link_T = link_T;
  sp.D = pose_dim;
  hipLaunchKernelGGL(gq_spheres_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), (size_t)4 * ((size_t)h->S * 16 + 512),
                     (hipStream_t)stream, sp, (int)batch);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_fc_pen_step(const gqFcStepDesc* fc, const gqPenStepDesc* pen, void* stream) {
  GQ_REQUIRE(fc && pen, "fc_pen_step: null descriptor");
  hipStream_t st = (hipStream_t)stream;
  GqFcStepArgs f{};
  float* runmin = nullptr;
  int rc = gq_fc_step_fill(fc->dist_sq, fc->sign, fc->obj_dir, fc->closest, fc->contact_pts, fc->hand_normals, fc->cog,
                           fc->batch, fc->n_contact, fc->n_cone, fc->friction, fc->torque_weight, fc->max_limit,
                           fc->svd_gain, fc->values_gain, fc->eps, fc->max_iter, fc->w_dis, fc->w_fc, fc->obj_normal,
                           fc->g_contact_pts, fc->g_hand_normals, fc->e_fc, fc->x_sum, fc->n_iter, fc->workspace,
                           fc->workspace_bytes, &f, &runmin);
  if (rc) return rc;
  GqPenArgs p{};
  rc = gq_pen_fill(pen->links, pen->surface_points, pen->n_obj, pen->n_surface, pen->batch_each, pen->hand_pose,
                   pen->pose_dim, pen->Rg, pen->link_T, pen->dis, pen->link, pen->gvec, pen->span, &p, pen->grid);
  if (rc) return rc;
  p.patch = pen->patch_spheres;
  GQ_REQUIRE(p.occ && p.cand_off, "fc_pen_step: the link mesh set has no voxel candidate lists (gq_meshset_build_occupancy)");
  GQ_REQUIRE(p.B == f.B, "fc_pen_step: the two descriptors disagree on the batch (%d vs %d)", p.B, f.B);
  GqPenBwdArgs pb{};
  rc = gq_pen_bwd_fill(p.L, pen->surface_points, pen->n_obj, pen->n_surface, pen->batch_each, pen->hand_pose, pen->pose_dim,
                       pen->Rg, nullptr, pen->link, pen->gvec, pen->link_wrench, pen->gRt, pen->dis, pen->w_pen, pen->e_pen,
                       pen->span, pen->span_acc, &pb);
  if (rc) return rc;
  const int ppt = gq_pen_points_per_thread_();
  const int gx = pen->grid ? 0 : (p.P + 256 * ppt - 1) / (256 * ppt);  // 0: the link-driven query, one block per row
  const bool two = f.nz > GQ_WAVE;
  GqSpenRole sp{};
  int n_sp = 0;
  rc = gq_spen_role_fill(pen, &sp, f.B, &n_sp);
  if (rc) return rc;
  const int nfc = (f.B + GQ_HEAD_ROWS - 1) / GQ_HEAD_ROWS;
  const size_t lds_a = std::max(pen->grid ? gq_pen_cells_lds_bytes(p.L, p.P) : gq_pen_grid_lds_bytes(p.L, GQ_PG_ECAP, GQ_PG_ICAP, ppt),
                                (size_t)GQ_HEAD_ROWS * f.n * 6 * sizeof(float) + GQ_HEAD_LDS_WORDS * sizeof(unsigned));
  const size_t lds_b = std::max(std::max(gq_pen_bwd_lds_bytes(), (size_t)f.nz * 3 * sizeof(float)),
                                n_sp ? (size_t)4 * ((size_t)sp.h.S * 16 + 512) : (size_t)0);
  const dim3 grid_a((unsigned)(nfc + (pen->grid ? 1 : gx) * p.B)), grid_b((unsigned)(2 * f.B + n_sp)), block(256);
#define GQ_STAGE_A(NCV, STOPV)                                                                                        \
  do {                                                                                                                \
    if (ppt == 2) hipLaunchKernelGGL((gq_stage_a_kernel<NCV, STOPV, 2>), grid_a, block, lds_a, st, f, p, gx, nfc);    \
    else hipLaunchKernelGGL((gq_stage_a_kernel<NCV, STOPV, 1>), grid_a, block, lds_a, st, f, p, gx, nfc);             \
  } while (0)
  if (f.agg) {
    if (two) GQ_STAGE_A(2, true);
    else GQ_STAGE_A(1, true);
  } else {
    if (two) GQ_STAGE_A(2, false);
    else GQ_STAGE_A(1, false);
  }
#undef GQ_STAGE_A
  GQ_LAUNCH_CHECK();
  const bool fused_stop = f.B <= 4 * GQ_WAVE && f.max_iter <= 16;
  if (!fused_stop && !f.agg) {
    rc = gq_qp_stop_launch_(f.resid, f.mu_tab, f.B, f.max_iter, f.eps, f.not_improved_lim, runmin, f.kstar, f.n_iter, stream);
    if (rc) return rc;
  }
  if (two) {
    if (fused_stop) hipLaunchKernelGGL((gq_stage_b_kernel<2, 4>), grid_b, block, lds_b, st, f, pb, sp);
    else hipLaunchKernelGGL((gq_stage_b_kernel<2, 0>), grid_b, block, lds_b, st, f, pb, sp);
  } else {
    if (fused_stop) hipLaunchKernelGGL((gq_stage_b_kernel<1, 4>), grid_b, block, lds_b, st, f, pb, sp);
    else hipLaunchKernelGGL((gq_stage_b_kernel<1, 0>), grid_b, block, lds_b, st, f, pb, sp);
  }
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
