// Force-closure term of one MALA* iteration in two per-row kernels (one wavefront = one grasp):
//
//   gq_fc_head_kernel   contact terms of E_dis (energy.py:25-28) -> outward object normals (object_model.py:246) ->
//                       friction-cone grasp matrix F (span.py:263-295,341-346) -> all PDIPM iterations of the box QP
//                       (qp_solver.py:60-134 -> qpth), F never leaves the registers in between
//   gq_fc_tail_kernel   qpth's batch-global stop rule (every wave replays it on the (B, max_iter) residual table: no
//                       grid-wide dependency, no extra launch) -> the row's best iterate -> E_fc (span.py:402,
//                       registry.py:82-87) -> KKT-implicit backward of the QP -> d E_fc / d contact points
//
// Same arithmetic as the building blocks gq_contact_terms / gq_fc_forward / gq_fc_backward (fc.hip, loop.hip), which
// remain the C-ABI surface of the autograd route; this file is what GraspStepper runs.
#include "fcstep_dev.h"

int gq_qp_stop_launch_(const float* resid, const float* mu, int B, int max_iter, float eps, int not_improved_lim,
                       float* runmin, int* kstar, int32_t* n_iter, void* stream);

template <int NC>
__global__ __launch_bounds__(GQ_WAVE) void gq_fc_head_kernel(GqFcStepArgs g) {
  extern __shared__ float gq_sh[];
  gq_fc_head_body<NC>(g, (int)blockIdx.x, gq_sh);
}
// large batches: GQ_HEAD_ROWS rows per block (one wavefront each) + the stop rule as epilogue of the last block
template <int NC>
__global__ __launch_bounds__(GQ_HEAD_ROWS* GQ_WAVE, NC == 1 ? 4 : 2) void gq_fc_head_stop_kernel(GqFcStepArgs g) {
  extern __shared__ float gq_sh[];
  const int blk = (int)blockIdx.x, wv = (int)threadIdx.x / GQ_WAVE, row = blk * GQ_HEAD_ROWS + wv;
  if (row >= g.B) return;
  float hr, hm;
  gq_fc_head_body<NC>(g, row, gq_sh + wv * g.n * 6, &hr, &hm);
  const int nrow = g.B - blk * GQ_HEAD_ROWS < GQ_HEAD_ROWS ? g.B - blk * GQ_HEAD_ROWS : GQ_HEAD_ROWS;
  gq_fc_head_epilogue(g, blk, wv, nrow, hr, hm, reinterpret_cast<unsigned*>(gq_sh + GQ_HEAD_ROWS * g.n * 6));
}
template <int NC, int RPL>
__global__ __launch_bounds__(GQ_WAVE, (NC == 1 && RPL == 0) ? 4 : 1) void gq_fc_tail_kernel(GqFcStepArgs g) {  // NC = 1: 4 wavefronts per SIMD (<= 128 VGPRs)
  extern __shared__ float gq_sh[];
  gq_fc_tail_body<NC, RPL>(g, (int)blockIdx.x, gq_sh);
}

extern "C" {

// Fused force-closure step of the MALA* iteration: contact terms + E_fc forward + backward with constant upstream
// weights (w_dis on E_dis, w_fc on E_fc).  Same workspace as gq_fc_forward (gq_fc_workspace_bytes); gq_fc_peek works
// on it afterwards (F, x, val, svd).  The workspace must be zero-filled ONCE after it is allocated (a block counter of
// the head launch lives in it and returns to zero by itself).
int gq_fc_step(const float* dist_sq, const int32_t* sign, const float* obj_dir, const float* closest,
               const float* contact_pts, const float* hand_normals, const float* cog, int64_t batch, int n_contact,
               int n_cone, float friction, float torque_weight, float max_limit, float svd_gain, float values_gain,
               float eps, int max_iter, float w_dis, float w_fc, float* obj_normal, float* g_contact_pts,
               float* g_hand_normals, float* e_fc, float* x_sum, int32_t* n_iter, void* workspace,
               size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GqFcStepArgs a{};
  float* runmin = nullptr;
  int rc = gq_fc_step_fill(dist_sq, sign, obj_dir, closest, contact_pts, hand_normals, cog, batch, n_contact, n_cone,
                           friction, torque_weight, max_limit, svd_gain, values_gain, eps, max_iter, w_dis, w_fc, obj_normal,
                           g_contact_pts, g_hand_normals, e_fc, x_sum, n_iter, workspace, workspace_bytes, &a, &runmin);
  if (rc) return rc;
  const int nz = a.nz;
  const dim3 grid((unsigned)batch), block(GQ_WAVE);
  const size_t lds_head = (size_t)n_contact * 6 * sizeof(float), lds_tail = (size_t)nz * 3 * sizeof(float);
  const bool two = nz > GQ_WAVE;
  if (a.agg) {  // the stop rule rides in the head launch (see gq_fc_head_epilogue)
    const dim3 hgrid((unsigned)a.head_blocks), hblock(GQ_HEAD_ROWS * GQ_WAVE);
    const size_t lds = GQ_HEAD_ROWS * lds_head + GQ_HEAD_LDS_WORDS * sizeof(unsigned);
    if (two) hipLaunchKernelGGL((gq_fc_head_stop_kernel<2>), hgrid, hblock, lds, st, a);
    else hipLaunchKernelGGL((gq_fc_head_stop_kernel<1>), hgrid, hblock, lds, st, a);
  } else if (two) {
    hipLaunchKernelGGL((gq_fc_head_kernel<2>), grid, block, lds_head, st, a);
  } else {
    hipLaunchKernelGGL((gq_fc_head_kernel<1>), grid, block, lds_head, st, a);
  }
  GQ_LAUNCH_CHECK();
  const bool fused_stop = batch <= 4 * GQ_WAVE && max_iter <= 16;
  if (!fused_stop && !a.agg) {
    rc = gq_qp_stop_launch_(a.resid, a.mu_tab, a.B, max_iter, eps, a.not_improved_lim, runmin, a.kstar, n_iter, stream);
    if (rc) return rc;
  }
  if (two) {
    if (fused_stop) hipLaunchKernelGGL((gq_fc_tail_kernel<2, 4>), grid, block, lds_tail, st, a);
    else hipLaunchKernelGGL((gq_fc_tail_kernel<2, 0>), grid, block, lds_tail, st, a);
  } else {
    if (fused_stop) hipLaunchKernelGGL((gq_fc_tail_kernel<1, 4>), grid, block, lds_tail, st, a);
    else hipLaunchKernelGGL((gq_fc_tail_kernel<1, 0>), grid, block, lds_tail, st, a);
  }
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
