// (Re-)initialisation of grasps on the device: reference core/initializations.py:15-193 (initialize_convex_hull), run at
// t = 0 and for the rows selected by the z-score rule every reset_epochs iterations (scripts/fit.py:315,408-422).
//
//   gq_init_sample_kernel   M points per object on its convex hull, area-weighted (face by CDF search, then
//                           origin + l1 e1 + l2 e2 with (l1, l2) folded into the triangle -- trimesh.sample.sample_surface),
//                           pushed out by `inflate` along the face normal (initializations.py:57-59)
//   gq_init_fps_kernel      farthest-point sampling of batch_each of them per object, starting at sample 0
//                           (pytorch3d.ops.sample_farthest_points, random_start_point=False; initializations.py:64-66):
//                           one 1024-thread block per object
//   gq_init_pose_kernel     per row: n = direction from the sample to the hull (= minus the face normal: on a convex hull
//                           the closest surface point of x + eps n_f is x), look-at rotation (initializations.py:83-117),
//                           random stand-off distance and roll / pitch / tilt (:119-146), translation and rot6d (:148-186),
//                           truncated-normal joint angles around the default state (:164-176)
//
// The random numbers are inputs (uniform draws made by the host's device generator), exactly like the draws of the MALA*
// step.  The hull itself (scipy / qhull at object set-up, like the reference's trimesh call) and its area CDF are set-up
// data.  trimesh's "even" rejection of close sample pairs is not applied: the farthest-point sampling makes it moot.
#include "common.h"
#include "wave.h"

struct GqInitArgs {
  const float* hull_fv;    // (sumF,3,3)
  const float* hull_cdf;   // (sumF) normalised cumulative area per object
  const int32_t* hull_off; // (n_obj+1)
  int n_obj, M, K;         // M samples per object, K = batch_each
  float inflate;
  const float* u_face;     // (n_obj,M)
  const float* u_len;      // (n_obj,M,2)
  float* pts;              // (n_obj,M,3) inflated samples
  float* nrm;              // (n_obj,M,3) face normals
  int32_t* sel;            // (n_obj,K)
  float* mind;             // (n_obj,M) running minimum squared distances of the farthest-point sampling
};

__global__ __launch_bounds__(256) void gq_init_sample_kernel(GqInitArgs g) {
  const int obj = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.M) return;
  const int f0 = g.hull_off[obj], f1 = g.hull_off[obj + 1];
  const float u = g.u_face[(size_t)obj * g.M + i];
  // searchsorted(cdf, u): first face whose cumulative share is >= u
  int lo = f0, hi = f1 - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (g.hull_cdf[mid] < u) lo = mid + 1;
    else hi = mid;
  }
  const float* v = g.hull_fv + (size_t)lo * 9;
  const gq3 a = gq_mk(v[0], v[1], v[2]);
  const gq3 e1 = gq_mk(v[3], v[4], v[5]) - a, e2 = gq_mk(v[6], v[7], v[8]) - a;
  float l1 = g.u_len[((size_t)obj * g.M + i) * 2], l2 = g.u_len[((size_t)obj * g.M + i) * 2 + 1];
  if (l1 + l2 > 1.0f) {
    l1 = fabsf(l1 - 1.0f);
    l2 = fabsf(l2 - 1.0f);
  }
  gq3 n = gq_cross(e1, e2);
  n = (1.0f / sqrtf(gq_dot(n, n))) * n;
  const gq3 p = a + l1 * e1 + l2 * e2 + g.inflate * n;
  float* o = g.pts + ((size_t)obj * g.M + i) * 3;
  o[0] = p.x; o[1] = p.y; o[2] = p.z;
  o = g.nrm + ((size_t)obj * g.M + i) * 3;
  o[0] = n.x; o[1] = n.y; o[2] = n.z;
}

// one block of 1024 threads per object; thread t owns samples t, t + 1024, ...; the running minimum distances live in
// the workspace (M floats per object: L2-resident), the samples are re-read from there as well
__global__ __launch_bounds__(1024) void gq_init_fps_kernel(GqInitArgs g) {
  __shared__ unsigned long long s_key[16];
  __shared__ float s_cur[3];
  __shared__ int s_pick;
  const int obj = blockIdx.x, tid = threadIdx.x, lane = gq_lane(), wv = tid / GQ_WAVE;
  const float* P = g.pts + (size_t)obj * g.M * 3;
  float* dist = g.mind + (size_t)obj * g.M;
  for (int i = tid; i < g.M; i += 1024) dist[i] = GQ_INF_F;
  int cur = 0;
  for (int it = 0; it < g.K; ++it) {
    if (tid == 0) {
      g.sel[(size_t)obj * g.K + it] = cur;
      s_cur[0] = P[cur * 3]; s_cur[1] = P[cur * 3 + 1]; s_cur[2] = P[cur * 3 + 2];
    }
    __syncthreads();
    const float cx = s_cur[0], cy = s_cur[1], cz = s_cur[2];
    // key = (distance bits, ~index): the maximum picks the largest distance and, among equals, the smallest index
    unsigned long long best = 0ull;
    for (int i = tid; i < g.M; i += 1024) {
      const float dx = P[i * 3] - cx, dy = P[i * 3 + 1] - cy, dz = P[i * 3 + 2] - cz;
      const float d = fminf(dist[i], dx * dx + dy * dy + dz * dz);
      dist[i] = d;
      const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
      best = key > best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o, GQ_WAVE);
      best = other > best ? other : best;
    }
    if (lane == 0) s_key[wv] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned long long b = s_key[0];
      for (int w = 1; w < 16; ++w) b = s_key[w] > b ? s_key[w] : b;
      s_pick = (int)(0xffffffffu - (unsigned)(b & 0xffffffffull));
    }
    __syncthreads();
    cur = s_pick;
  }
}

struct GqInitPoseArgs {
  const float* pts; const float* nrm; const int32_t* sel;
  int n_obj, M, K, J, D;
  float fwd[3], up[3];
  const float* default_state; const float* jlo; const float* jhi;
  float jitter, d_lo, d_hi, rot_lo, rot_hi, pitch_lo, pitch_hi, tilt_lo, tilt_hi;
  const float* u_pose;   // (B,4): distance, rotate, pitch, tilt
  const float* u_joint;  // (B,J)
  float* hand_pose;      // (B,D)
  float* p_out;          // (B,3) or null: the shell point of the row
  float* n_out;          // (B,3) or null: the direction towards the hull
};

__device__ __forceinline__ void gq_mat3_mul(const float* A, const float* Bm, float* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * Bm[j] + A[i * 3 + 1] * Bm[3 + j] + A[i * 3 + 2] * Bm[6 + j];
}

__global__ __launch_bounds__(256) void gq_init_pose_kernel(GqInitPoseArgs g) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  const int B = g.n_obj * g.K;
  if (row >= B) return;
  const int obj = row / g.K, j = row % g.K;
  const int s = g.sel[(size_t)obj * g.K + j];
  const float* pp = g.pts + ((size_t)obj * g.M + s) * 3;
  const float* nn = g.nrm + ((size_t)obj * g.M + s) * 3;
  const gq3 p = gq_mk(pp[0], pp[1], pp[2]);
  const gq3 n = gq_mk(-nn[0], -nn[1], -nn[2]);  // towards the hull
  // look_at(p, p + n): forward = camera - target = -n (initializations.py:96-97)
  gq3 fw = gq_mk(-n.x, -n.y, -n.z);
  fw = (1.0f / sqrtf(gq_dot(fw, fw))) * fw;
  gq3 up = gq_mk(g.up[0], g.up[1], g.up[2]);
  if (!(fabsf(gq_dot(up, fw)) < 0.95f)) up = gq_mk(0.0f, 1.0f, 0.0f);
  gq3 right = gq_cross(up, fw);
  right = (1.0f / sqrtf(gq_dot(right, right))) * right;
  const gq3 upv = gq_cross(fw, right);
  const float O[9] = {fw.x, upv.x, right.x, fw.y, upv.y, right.y, fw.z, upv.z, right.z};  // columns forward, up, right
  const gq3 f0 = gq_mk(g.fwd[0], g.fwd[1], g.fwd[2]), u0 = gq_mk(g.up[0], g.up[1], g.up[2]);
  const gq3 c1 = gq_cross(f0, u0);
  const float Bs[9] = {f0.x, -c1.x, u0.x, f0.y, -c1.y, u0.y, f0.z, -c1.z, u0.z};  // columns forward, -(forward x up), up
  float Rg[9];
  gq_mat3_mul(O, Bs, Rg);
  const float* u = g.u_pose + (size_t)row * 4;
  const float dist = g.d_lo + (g.d_hi - g.d_lo) * u[0];
  const float rot = g.rot_lo + (g.rot_hi - g.rot_lo) * u[1];
  const float pitch = g.pitch_lo + (g.pitch_hi - g.pitch_lo) * u[2];
  const float tilt = g.tilt_lo + (g.tilt_hi - g.tilt_lo) * u[3];
  // euler2mat(tilt, pitch, rotate, 'rxyz') = Rx(tilt) Ry(pitch) Rz(rotate)
  float si, ci, sj, cj, sk, ck;
  sincosf(tilt, &si, &ci);
  sincosf(pitch, &sj, &cj);
  sincosf(rot, &sk, &ck);
  const float Rl[9] = {cj * ck, -cj * sk, sj,
                       ci * sk + si * sj * ck, ci * ck - si * sj * sk, -si * cj,
                       si * sk - ci * sj * ck, si * ck + ci * sj * sk, ci * cj};
  float R[9];
  gq_mat3_mul(Rg, Rl, R);
  float* hp = g.hand_pose + (size_t)row * g.D;
  hp[0] = p.x - dist * n.x; hp[1] = p.y - dist * n.y; hp[2] = p.z - dist * n.z;
  hp[3] = R[0]; hp[4] = R[3]; hp[5] = R[6];  // first column of R
  hp[6] = R[1]; hp[7] = R[4]; hp[8] = R[7];  // second column
  for (int d = 0; d < g.J; ++d) {
    const float lo = g.jlo[d], hi = g.jhi[d];
    const float mu = fminf(fmaxf(g.default_state[d], lo), hi);
    const float sg = g.jitter * (hi - lo);
    const float a = lo - 1e-6f, b = hi + 1e-6f;
    // torch.nn.init.trunc_normal_: uniform in [2 Phi(alpha) - 1, 2 Phi(beta) - 1], erfinv, scale, shift, clamp
    const float l = erff((a - mu) / sg * 0.70710678118654752f), h2 = erff((b - mu) / sg * 0.70710678118654752f);
    const float v = l + g.u_joint[(size_t)row * g.J + d] * (h2 - l);
    float x = erfinvf(v) * sg * 1.4142135623730951f + mu;
    x = fminf(fmaxf(x, a), b);
    hp[9 + d] = x;
  }
  if (g.p_out) { g.p_out[row * 3] = p.x; g.p_out[row * 3 + 1] = p.y; g.p_out[row * 3 + 2] = p.z; }
  if (g.n_out) { g.n_out[row * 3] = n.x; g.n_out[row * 3 + 1] = n.y; g.n_out[row * 3 + 2] = n.z; }
}

extern "C" {

int gq_init_workspace_bytes(int64_t n_obj, int64_t samples_per_object, int64_t batch_each, size_t* bytes) {
  GQ_REQUIRE(bytes && n_obj > 0 && samples_per_object > 0 && batch_each > 0, "init_workspace_bytes: bad arguments");
  *bytes = (size_t)n_obj * samples_per_object * 7 * sizeof(float) + (size_t)n_obj * batch_each * sizeof(int32_t) + 512;
  return GQ_OK;
}

int gq_init_convex_hull(const gqInitDesc* d, void* stream) {
  GQ_REQUIRE(d, "init_convex_hull: null descriptor");
  GQ_REQUIRE(d->hull_face_verts && d->hull_cdf && d->hull_offsets && d->u_face && d->u_len && d->u_pose && d->u_joint &&
                 d->default_state && d->joints_lower && d->joints_upper && d->hand_pose && d->workspace,
             "init_convex_hull: null pointer in the descriptor");
  GQ_REQUIRE(d->n_obj > 0 && d->batch_each > 0 && d->n_dofs > 0 && d->samples_per_object >= d->batch_each,
             "init_convex_hull: bad sizes (n_obj=%lld, batch_each=%lld, samples=%lld)", (long long)d->n_obj,
             (long long)d->batch_each, (long long)d->samples_per_object);
  GQ_REQUIRE(d->samples_per_object < (1ll << 30), "init_convex_hull: too many samples per object");
  size_t need = 0;
  gq_init_workspace_bytes(d->n_obj, d->samples_per_object, d->batch_each, &need);
  GQ_REQUIRE(d->workspace_bytes >= need, "init_convex_hull: workspace too small (%zu < %zu)", d->workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  GqInitArgs a{};
  a.hull_fv = d->hull_face_verts;
  a.hull_cdf = d->hull_cdf;
  a.hull_off = d->hull_offsets;
  a.n_obj = (int)d->n_obj;
  a.M = (int)d->samples_per_object;
  a.K = (int)d->batch_each;
  a.inflate = d->inflate;
  a.u_face = d->u_face;
  a.u_len = d->u_len;
  char* w = (char*)d->workspace;
  a.pts = (float*)w;
  a.nrm = a.pts + (size_t)a.n_obj * a.M * 3;
  a.mind = a.nrm + (size_t)a.n_obj * a.M * 3;
  a.sel = (int32_t*)(a.mind + (size_t)a.n_obj * a.M);
  hipLaunchKernelGGL(gq_init_sample_kernel, dim3((unsigned)((a.M + 255) / 256), (unsigned)a.n_obj), dim3(256), 0, st, a);
  GQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(gq_init_fps_kernel, dim3((unsigned)a.n_obj), dim3(1024), 0, st, a);
  GQ_LAUNCH_CHECK();
  GqInitPoseArgs p{};
  p.pts = a.pts;
  p.nrm = a.nrm;
  p.sel = a.sel;
  p.n_obj = a.n_obj;
  p.M = a.M;
  p.K = a.K;
  p.J = d->n_dofs;
  p.D = 9 + d->n_dofs;
  for (int i = 0; i < 3; ++i) {
    p.fwd[i] = d->forward_axis[i];
    p.up[i] = d->up_axis[i];
  }
  p.default_state = d->default_state;
  p.jlo = d->joints_lower;
  p.jhi = d->joints_upper;
  p.jitter = d->jitter_strength;
  p.d_lo = d->distance_lower; p.d_hi = d->distance_upper;
  p.rot_lo = d->rotate_lower; p.rot_hi = d->rotate_upper;
  p.pitch_lo = d->pitch_lower; p.pitch_hi = d->pitch_upper;
  p.tilt_lo = d->tilt_lower; p.tilt_hi = d->tilt_upper;
  p.u_pose = d->u_pose;
  p.u_joint = d->u_joint;
  p.hand_pose = d->hand_pose;
  p.p_out = d->shell_points;
  p.n_out = d->shell_dirs;
  const int B = a.n_obj * a.K;
  hipLaunchKernelGGL(gq_init_pose_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, p);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Surface samples of the object meshes (reference core/object_model.py:163-178: pytorch3d sample_points_from_meshes with
// 100 x num_samples points, then sample_farthest_points down to num_samples, first point = sample 0): the two kernels of
// the hull initialisation with inflate = 0 on the object's own faces, then a gather of the picked samples.
__global__ void gq_gather_points_kernel(const float* __restrict__ pts, const int32_t* __restrict__ sel, int M, int K,
                                        float* __restrict__ out) {
  const int obj = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  const float* p = pts + ((size_t)obj * M + sel[(size_t)obj * K + i]) * 3;
  float* o = out + ((size_t)obj * K + i) * 3;
  o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

int gq_surface_fps(const float* face_verts, const float* area_cdf, const int32_t* face_offsets, int64_t n_obj,
                   int64_t samples_per_object, int64_t n_keep, const float* u_face, const float* u_len, float* points_out,
                   void* workspace, size_t workspace_bytes, void* stream) {
  GQ_REQUIRE(face_verts && area_cdf && face_offsets && u_face && u_len && points_out && workspace,
             "surface_fps: null pointer");
  GQ_REQUIRE(n_obj > 0 && n_keep > 0 && samples_per_object >= n_keep && samples_per_object < (1ll << 30),
             "surface_fps: bad sizes (n_obj=%lld, samples=%lld, keep=%lld)", (long long)n_obj, (long long)samples_per_object,
             (long long)n_keep);
  size_t need = 0;
  gq_init_workspace_bytes(n_obj, samples_per_object, n_keep, &need);
  GQ_REQUIRE(workspace_bytes >= need, "surface_fps: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  GqInitArgs a{};
  a.hull_fv = face_verts;
  a.hull_cdf = area_cdf;
  a.hull_off = face_offsets;
  a.n_obj = (int)n_obj;
  a.M = (int)samples_per_object;
  a.K = (int)n_keep;
  a.inflate = 0.0f;
  a.u_face = u_face;
  a.u_len = u_len;
  a.pts = (float*)workspace;
  a.nrm = a.pts + (size_t)a.n_obj * a.M * 3;
  a.mind = a.nrm + (size_t)a.n_obj * a.M * 3;
  a.sel = (int32_t*)(a.mind + (size_t)a.n_obj * a.M);
  hipLaunchKernelGGL(gq_init_sample_kernel, dim3((unsigned)((a.M + 255) / 256), (unsigned)a.n_obj), dim3(256), 0, st, a);
  GQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(gq_init_fps_kernel, dim3((unsigned)a.n_obj), dim3(1024), 0, st, a);
  GQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(gq_gather_points_kernel, dim3((unsigned)((a.K + 255) / 256), (unsigned)a.n_obj), dim3(256), 0, st, a.pts,
                     a.sel, a.M, a.K, points_out);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
