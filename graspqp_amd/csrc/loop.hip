// Elementwise / small-reduction pieces of one MALA* iteration:
//   gq_energy_combine   E_dis, E_joints, weighted total and the gradients they feed into FK backward
//                       (reference core/energy.py:25-28,47-54; scripts/fit.py:434-438)
//   gq_mala_propose     RMS-normalised gradient step + contact re-sampling  (core/optimizer.py:199-273)
//   gq_zscore           per-object z-score of the accepted energies         (scripts/fit.py:403-406)
//   gq_mala_accept      Metropolis accept with z-score-scaled temperature and state merge (optimizer.py:289-340,
//                       fit.py:454-458)
// Random numbers are inputs (drawn by the host-side generator), exactly like the oracle.
#include "fc_dev.h"
#include "loop_dev.h"

struct GqCombineArgs {
  const float* dist_sq;   // (B,n) object SDF squared distance of the contact points
  const int32_t* sign;    // (B,n)
  const float* onrm;      // (B,n,3) unit (p - closest)/|.|
  const float* closest;   // (B,n,3)
  const float* cpts;      // (B,n,3)
  const float* cnrm;      // (B,n,3) hand contact normals (world)
  const float* hand_pose; // (B,D)
  const float* jlo;
  const float* jhi;
  const float* e_fc;      // (B)
  const float* pen_dis;   // (B,P) max-over-links signed distance of the object surface points
  const float* e_spen;    // (B)
  int B, n, D, J, P;
  float w_dis, w_fc, w_pen, w_spen, w_joints;
  float* e_dis;      // (B)
  float* e_joints;   // (B)
  float* e_pen;      // (B)
  float* total;      // (B)
  float* obj_normal; // (B,n,3) outward object normal = onrm * sign  (contact normals fed to E_fc)
  float* g_cpts;     // (B,n,3)  w_dis * dE_dis/dp   (E_fc part is added by the caller's fc backward)
  float* g_cnrm;     // (B,n,3)  w_dis * dE_dis/dnH
  float* g_theta;    // (B,J)    w_joints * dE_joints/dtheta
  float* g_pen;      // (B,P)    w_pen * [dis > 0]
};

// stage 1 (before E_fc is known): per-contact quantities; one thread per (row, contact)
__global__ void gq_contact_terms_kernel(GqCombineArgs g) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)g.B * g.n) return;
  const gq3 on = gq_mk(g.onrm[t * 3], g.onrm[t * 3 + 1], g.onrm[t * 3 + 2]);
  const gq3 nH = gq_mk(g.cnrm[t * 3], g.cnrm[t * 3 + 1], g.cnrm[t * 3 + 2]);
  const gq3 p = gq_mk(g.cpts[t * 3], g.cpts[t * 3 + 1], g.cpts[t * 3 + 2]);
  const gq3 cl = gq_mk(g.closest[t * 3], g.closest[t * 3 + 1], g.closest[t * 3 + 2]);
  const GqContactTerm c = gq_contact_term(g.dist_sq[t], (float)g.sign[t], on, nH, p, cl, g.w_dis);
  g.obj_normal[t * 3] = c.vC.x;
  g.obj_normal[t * 3 + 1] = c.vC.y;
  g.obj_normal[t * 3 + 2] = c.vC.z;
  g.g_cpts[t * 3] = c.g_p.x;
  g.g_cpts[t * 3 + 1] = c.g_p.y;
  g.g_cpts[t * 3 + 2] = c.g_p.z;
  g.g_cnrm[t * 3] = c.g_n.x;
  g.g_cnrm[t * 3 + 1] = c.g_n.y;
  g.g_cnrm[t * 3 + 2] = c.g_n.z;
}

// stage 2: per-row reductions; one block of 256 threads per row (deterministic tree reduction)
__global__ __launch_bounds__(256) void gq_row_energy_kernel(GqCombineArgs g) {
  __shared__ float red[256];
  const int row = blockIdx.x, tid = threadIdx.x;
  // E_pen = sum relu(dis) and its mask gradient
  float acc = 0.0f;
  for (int p = tid; p < g.P; p += 256) {
    const size_t o = (size_t)row * g.P + p;
    const float d = g.pen_dis[o];
    const bool in = d > 0.0f;
    acc += in ? d : 0.0f;
    g.g_pen[o] = in ? g.w_pen : 0.0f;
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float e_pen = red[0];
  __syncthreads();
  // E_dis
  float ed = 0.0f;
  for (int c = tid; c < g.n; c += 256) {
    const size_t t = (size_t)row * g.n + c;
    const float root = sqrtf(g.dist_sq[t] + 1e-8f);
    const float sg = (float)g.sign[t];
    const float dt = sg * (g.onrm[t * 3] * g.cnrm[t * 3] + g.onrm[t * 3 + 1] * g.cnrm[t * 3 + 1] +
                           g.onrm[t * 3 + 2] * g.cnrm[t * 3 + 2]);
    ed += expf(1.0f + dt) * root;
  }
  red[tid] = ed;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float e_dis = red[0];
  __syncthreads();
  // E_joints
  float ej = 0.0f;
  for (int j = tid; j < g.J; j += 256) {
    const float th = g.hand_pose[(size_t)row * g.D + 9 + j];
    const float hi = g.jhi[j], lo = g.jlo[j];
    float gt = 0.0f;
    if (th > hi) {
      ej += th - hi;
      gt += 1.0f;
    }
    if (th < lo) {
      ej += lo - th;
      gt -= 1.0f;
    }
    g.g_theta[(size_t)row * g.J + j] = g.w_joints * gt;
  }
  red[tid] = ej;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const float e_joints = red[0];
    g.e_dis[row] = e_dis;
    g.e_pen[row] = e_pen;
    g.e_joints[row] = e_joints;
    g.total[row] = g.w_dis * e_dis + g.w_fc * g.e_fc[row] + g.w_pen * e_pen + g.w_spen * g.e_spen[row] +
                   g.w_joints * e_joints;
  }
}

__global__ void gq_axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fmaf(a, x[i], y[i]);
}

__global__ void gq_scale_kernel(float* __restrict__ y, const float* __restrict__ x, float a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i];
}

__global__ void gq_fill_kernel(float* __restrict__ y, float a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a;
}

// ---- MALA* --------------------------------------------------------------------------------------------------------
// column mean of grad^2 over ALL rows (optimizer.py:231); one block per column, fixed-order tree reduction
__global__ __launch_bounds__(256) void gq_colsq_mean_kernel(const float* __restrict__ grad, int B, int D, int clip,
                                                            float* __restrict__ g2) {
  __shared__ float red[256];
  const int col = blockIdx.x, tid = threadIdx.x;
  float acc = 0.0f;
  for (int r = tid; r < B; r += 256) {
    float v = grad[(size_t)r * D + col];
    if (clip) {
      v = (v != v) ? 0.0f : fminf(fmaxf(v, -100.0f), 100.0f);  // NaN -> 0 first: fmaxf(NaN, -100) would be -100
    }
    acc = fmaf(v, v, acc);
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) g2[col] = red[0] / (float)B;
}

// same means for batches <= 512 rows and D <= 64 in the canonical order of loop_dev.h: wavefront u = unit u
__global__ __launch_bounds__(GQ_COLSQ_UNITS * GQ_WAVE) void gq_colsq_units_kernel(const float* __restrict__ grad, int B, int D,
                                                                              int clip, float* __restrict__ g2) {
  __shared__ float sPart[GQ_COLSQ_UNITS * GQ_WAVE];
  const int lane = gq_lane(), unit = (int)threadIdx.x / GQ_WAVE;
  if (lane < D) sPart[unit * D + lane] = gq_colsq_unit(grad, B, D, clip, unit, lane);
  __syncthreads();
  if (unit == 0 && lane < D) g2[lane] = gq_colsq_finish(sPart, B, D, lane);
}

// z = (E - mean_obj) / std_obj (unbiased), one block per object
__global__ __launch_bounds__(256) void gq_zscore_kernel(const float* __restrict__ energy, int batch_each,
                                                        float* __restrict__ z) {
  __shared__ float red[256];
  __shared__ float s_mean, s_std;
  const int obj = blockIdx.x, tid = threadIdx.x;
  const float* e = energy + (size_t)obj * batch_each;
  float acc = 0.0f;
  for (int i = tid; i < batch_each; i += 256) acc += e[i];
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) s_mean = red[0] / (float)batch_each;
  __syncthreads();
  const float mean = s_mean;
  acc = 0.0f;
  for (int i = tid; i < batch_each; i += 256) {
    const float d = e[i] - mean;
    acc = fmaf(d, d, acc);
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) s_std = sqrtf(red[0] / (float)(batch_each - 1));
  __syncthreads();
  for (int i = tid; i < batch_each; i += 256) z[(size_t)obj * batch_each + i] = (e[i] - mean) / s_std;
}

__global__ __launch_bounds__(GQ_WAVE) void gq_mala_propose_kernel(GqProposeArgs g) {
  gq_propose_body(g, (int)blockIdx.x, gq_lane());
}

__global__ __launch_bounds__(GQ_WAVE) void gq_mala_accept_kernel(GqAcceptArgs g) {
  gq_accept_body(g, (int)blockIdx.x, gq_lane());
}

int gq_colsq_launch_(const float* grad, int B, int D, int clip, float* g2, void* stream) {
  if (B <= 512 && D <= GQ_WAVE)
    hipLaunchKernelGGL(gq_colsq_units_kernel, dim3(1), dim3(GQ_COLSQ_UNITS * GQ_WAVE), 0, (hipStream_t)stream, grad, B, D, clip, g2);
  else
    hipLaunchKernelGGL(gq_colsq_mean_kernel, dim3((unsigned)D), dim3(256), 0, (hipStream_t)stream, grad, B, D, clip, g2);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

extern "C" {

// Everything of core/energy.py that is not a kernel of its own.  Outputs see GqCombineArgs.
int gq_contact_terms(const float* dist_sq, const int32_t* sign, const float* onrm, const float* closest,
                     const float* contact_pts, const float* contact_normals, int64_t batch, int n_contact, float w_dis,
                     float* obj_normal, float* g_contact_pts, float* g_contact_normals, void* stream) {
  GQ_REQUIRE(dist_sq && sign && onrm && closest && contact_pts && contact_normals && obj_normal && g_contact_pts &&
                 g_contact_normals && batch > 0 && n_contact > 0,
             "contact_terms: bad arguments");
  GqCombineArgs a{};
  a.dist_sq = dist_sq;
  a.sign = sign;
  a.onrm = onrm;
  a.closest = closest;
  a.cpts = contact_pts;
  a.cnrm = contact_normals;
  a.B = (int)batch;
  a.n = n_contact;
  a.w_dis = w_dis;
  a.obj_normal = obj_normal;
  a.g_cpts = g_contact_pts;
  a.g_cnrm = g_contact_normals;
  const int64_t tot = batch * n_contact;
  hipLaunchKernelGGL(gq_contact_terms_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_row_energy(const float* dist_sq, const int32_t* sign, const float* onrm, const float* contact_normals,
                  const float* hand_pose, const float* joints_lower, const float* joints_upper, const float* e_fc,
                  const float* pen_dis, const float* e_spen, int64_t batch, int n_contact, int n_dofs,
                  int64_t n_surface, float w_dis, float w_fc, float w_pen, float w_spen, float w_joints, float* e_dis,
                  float* e_joints, float* e_pen, float* total, float* g_theta, float* g_pen, void* stream) {
  GQ_REQUIRE(dist_sq && sign && onrm && contact_normals && hand_pose && joints_lower && joints_upper && e_fc &&
                 pen_dis && e_spen && e_dis && e_joints && e_pen && total && g_theta && g_pen && batch > 0,
             "row_energy: bad arguments");
  GqCombineArgs a{};
  a.dist_sq = dist_sq;
  a.sign = sign;
  a.onrm = onrm;
  a.cnrm = contact_normals;
  a.hand_pose = hand_pose;
  a.jlo = joints_lower;
  a.jhi = joints_upper;
  a.e_fc = e_fc;
  a.pen_dis = pen_dis;
  a.e_spen = e_spen;
  a.B = (int)batch;
  a.n = n_contact;
  a.D = 9 + n_dofs;
  a.J = n_dofs;
  a.P = (int)n_surface;
  a.w_dis = w_dis;
  a.w_fc = w_fc;
  a.w_pen = w_pen;
  a.w_spen = w_spen;
  a.w_joints = w_joints;
  a.e_dis = e_dis;
  a.e_joints = e_joints;
  a.e_pen = e_pen;
  a.total = total;
  a.g_theta = g_theta;
  a.g_pen = g_pen;
  hipLaunchKernelGGL(gq_row_energy_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_axpy(float* y, const float* x, float a, int64_t n, void* stream) {
  if (n == 0) return GQ_OK;
  GQ_REQUIRE(y && x && n > 0, "axpy: bad arguments");
  hipLaunchKernelGGL(gq_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_scale(float* y, const float* x, float a, int64_t n, void* stream) {
  if (n == 0) return GQ_OK;
  GQ_REQUIRE(y && x && n > 0, "scale: bad arguments");
  hipLaunchKernelGGL(gq_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_fill(float* y, float a, int64_t n, void* stream) {
  if (n == 0) return GQ_OK;
  GQ_REQUIRE(y && n > 0, "fill: bad arguments");
  hipLaunchKernelGGL(gq_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, a, n);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// MalaStar.try_step (optimizer.py:199-273) with the draws injected; g2_scratch: (D) floats
int gq_mala_propose(const float* hand_pose, const float* grad, const int64_t* contact_idx, const float* u_switch,
                    const int64_t* new_idx, int64_t batch, int pose_dim, int n_contact, float step_size,
                    int stepsize_period, float decay, float mu, float switch_possibility, int clip_grad, float* ema,
                    int64_t* step, float* pose_out, int64_t* idx_out, float* step_size_out, float* g2_scratch,
                    const float* energy, int64_t batch_each, float* z_out, void* stream) {
  GQ_REQUIRE(hand_pose && grad && contact_idx && u_switch && new_idx && ema && step && pose_out && idx_out &&
                 g2_scratch && batch > 0 && pose_dim > 9 && pose_dim <= 128 && n_contact > 0 && stepsize_period > 0,
             "mala_propose: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  {
    const int rc = gq_colsq_launch_(grad, (int)batch, pose_dim, clip_grad, g2_scratch, stream);
    if (rc) return rc;
  }
  GqProposeArgs a{};
  a.hand_pose = hand_pose;
  a.grad = grad;
  a.g2 = g2_scratch;
  a.idx = contact_idx;
  a.u_switch = u_switch;
  a.new_idx = new_idx;
  a.B = (int)batch;
  a.D = pose_dim;
  a.n = n_contact;
  a.clip = clip_grad;
  a.step_size = step_size;
  a.decay = decay;
  a.mu = mu;
  a.switch_p = switch_possibility;
  a.stepsize_period = stepsize_period;
  a.ema = ema;
  a.step = step;
  a.pose_out = pose_out;
  a.idx_out = idx_out;
  a.s_out = step_size_out;
  GQ_REQUIRE(energy == nullptr || (z_out != nullptr && batch_each > 0 && batch % batch_each == 0),
             "mala_propose: z-score needs z_out and batch_each dividing batch");
  a.energy = energy;
  a.batch_each = (int)batch_each;
  a.z_out = z_out;
  hipLaunchKernelGGL(gq_mala_propose_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), 0, st, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_zscore(const float* energy, int64_t n_obj, int64_t batch_each, float* z, void* stream) {
  GQ_REQUIRE(energy && z && n_obj > 0 && batch_each > 0, "zscore: bad arguments");
  hipLaunchKernelGGL(gq_zscore_kernel, dim3((unsigned)n_obj), dim3(256), 0, (hipStream_t)stream, energy,
                     (int)batch_each, z);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// MalaStar.accept_step + the masked state update of fit.py:454-458
int gq_mala_accept(const float* new_energy, const float* u_accept, const float* z, const uint8_t* reset_mask,
                   const int64_t* step, const float* pose_new, const int64_t* idx_new, const float* grad_new,
                   int64_t batch, int pose_dim, int n_contact, float starting_temperature, float decay,
                   int annealing_period, float* energy, float* pose, int64_t* idx, float* grad, uint8_t* accept,
                   float* temperature, int n_terms, const float* terms_new, float* terms, void* stream) {
  GQ_REQUIRE(new_energy && u_accept && step && pose_new && idx_new && grad_new && energy && pose && idx && grad &&
                 accept && batch > 0 && annealing_period > 0,
             "mala_accept: bad arguments");
  GQ_REQUIRE(n_terms == 0 || (terms_new && terms), "mala_accept: null term buffers");
  GqAcceptArgs a{};
  a.new_energy = new_energy;
  a.u_accept = u_accept;
  a.z = z;
  a.reset_mask = reset_mask;
  a.step = step;
  a.pose_new = pose_new;
  a.idx_new = idx_new;
  a.grad_new = grad_new;
  a.B = (int)batch;
  a.D = pose_dim;
  a.n = n_contact;
  a.T0 = starting_temperature;
  a.decay = decay;
  a.annealing_period = annealing_period;
  a.energy = energy;
  a.pose = pose;
  a.idx = idx;
  a.grad = grad;
  a.accept = accept;
  a.temperature = temperature;
  a.n_terms = n_terms;
  a.terms_new = terms_new;
  a.terms = terms;
  hipLaunchKernelGGL(gq_mala_accept_kernel, dim3((unsigned)batch), dim3(GQ_WAVE), 0, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
