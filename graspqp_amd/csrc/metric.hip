// The two other force-closure energies the reference can be run with (scripts/fit.py:343-347, --energy_type):
//
//   dexgrasp   metrics/ops/dexgrasp.py:4-34   E = | sum_i [n_i ; tw (n_i x r_i)] |^2,  r_i = p_i - cog
//              (DexGraspNet's ||G'n||^2; tw = 0 at the reference's call site, core/energy.py:35-42)
//   tdg        metrics/ops/tdg.py:147-239     task-oriented grasp-wrench-space energy: for each of P sampled unit
//              directions u the friction-cone force of every contact that reaches farthest along G'u, summed (weighted by
//              the contact density) into a wrench W(u); E = 100 * mean_u (1 - cos(W(u), [u; 0]))
//
// Both take the contact points, the OBJECT normals at the contacts (constants: the reference gets them from the SDF
// without gradient) and the centre of gravity, and return the energy and its gradient w.r.t. the contact points in one
// launch (forward + analytic backward fused: the gradient is a by-product of the same sums).
#include "metric_dev.h"

__global__ __launch_bounds__(GQ_WAVE) void gq_dexgrasp_kernel(GqDexArgs g) {
  const int row = blockIdx.x;
  gq_dexgrasp_body(g, row, gq_lane(), g.cpts + (size_t)row * g.n * 3, g.cnrm + (size_t)row * g.n * 3);
}
__global__ __launch_bounds__(256) void gq_tdg_kernel(GqTdgArgs g) {
  extern __shared__ float gq_tdg_lds[];
  const int row = blockIdx.x;
  gq_tdg_body(g, row, gq_tdg_lds, g.cpts + (size_t)row * g.n * 3, g.cnrm + (size_t)row * g.n * 3);
}

extern "C" {

int gq_dexgrasp_energy(const float* contact_pts, const float* contact_normals, const float* cog, int64_t batch,
                       int n_contact, float torque_weight, const float* grad_e, float w, int accumulate, float* e_fc,
                       float* g_contact_pts, void* stream) {
  GQ_REQUIRE(contact_pts && contact_normals && cog && (e_fc || g_contact_pts), "dexgrasp_energy: null pointer");
  GQ_REQUIRE(batch > 0 && n_contact > 0, "dexgrasp_energy: bad sizes");
  GqDexArgs a{};
  a.cpts = contact_pts; a.cnrm = contact_normals; a.cog = cog;
  a.B = (int)batch; a.n = n_contact; a.tw = torque_weight;
  a.grad_e = grad_e; a.w = w; a.accumulate = accumulate;
  a.e = e_fc; a.g_cpts = g_contact_pts;
  hipLaunchKernelGGL(gq_dexgrasp_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_tdg_energy(const float* contact_pts, const float* contact_normals, const float* cog, const float* directions,
                  int n_directions, int64_t batch, int n_contact, float friction, float obb_length, int enable_density,
                  float scale, const float* grad_e, float w, int accumulate, float* e_fc, float* g_contact_pts,
                  void* stream) {
  GQ_REQUIRE(contact_pts && contact_normals && cog && directions && (e_fc || g_contact_pts), "tdg_energy: null pointer");
  GQ_REQUIRE(batch > 0 && n_contact > 0 && n_contact <= 1024 && n_directions > 0 && friction > 0.0f && friction <= 1.0f &&
                 obb_length > 0.0f, "tdg_energy: bad arguments");
  GqTdgArgs a{};
  a.cpts = contact_pts; a.cnrm = contact_normals; a.cog = cog; a.dirs = directions;
  a.B = (int)batch; a.n = n_contact; a.P = n_directions;
  a.miu = friction; a.inv_obb = 1.0f / obb_length; a.scale = scale; a.density = enable_density;
  a.grad_e = grad_e; a.w = w; a.accumulate = accumulate;
  a.e = e_fc; a.g_cpts = g_contact_pts;
  const size_t lds = gq_tdg_lds_floats(n_contact) * sizeof(float);
  hipLaunchKernelGGL(gq_tdg_kernel, dim3((unsigned)batch), dim3(256), lds, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
