// Mesh signed-distance queries with the TorchSDF output contract (reference call sites object_model.py:220,
// hand_model.py:953) and the fused hand-penetration query (hand_model.py:875-987).
//
// Two kernel shapes cover the two regimes of the grasp loop:
//   * "wave per query"  (contacts vs object mesh: few queries, 1e3-2e4 faces): the 64 lanes of a wavefront stride
//     over the face records (one 64-byte record per lane -> four coalesced 16-byte loads), keep a running
//     (dist^2, face) minimum and reduce it with a 64-bit key so the lowest face index wins ties.
//   * "point per lane"  (object surface points vs hand-link meshes: ~1e6 queries, 2e2-1e3 faces per link): every
//     lane owns one point; the face records are wave-uniform so they travel through the scalar cache / SGPRs
//     and the per-lane work is pure FP32 VALU.
#include "tri.h"

__global__ void gq_face_prep_kernel(const float* __restrict__ fv, GqFace* __restrict__ rec, int64_t F) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  const float* v = fv + i * 9;
  rec[i] = gq_make_face(gq_mk(v[0], v[1], v[2]), gq_mk(v[3], v[4], v[5]), gq_mk(v[6], v[7], v[8]));
}

// ---- wave per query -------------------------------------------------------------------------------------------
// queries are grouped: query q uses mesh (q / queries_per_mesh); mesh m's records are rec[off[m] .. off[m+1])
__global__ __launch_bounds__(256) void gq_sdf_wave_kernel(const float* __restrict__ points, int64_t N,
                                                          const GqFace* __restrict__ rec,
                                                          const int32_t* __restrict__ off, int single_F,
                                                          int64_t queries_per_mesh,
                                                          float* __restrict__ dist_sq, int32_t* __restrict__ sign,
                                                          float* __restrict__ normal, float* __restrict__ closest) {
  const int64_t q = (int64_t)blockIdx.x * (blockDim.x / GQ_WAVE) + (threadIdx.x / GQ_WAVE);
  if (q >= N) return;
  const int lane = gq_lane();
  const int mesh = (int)(q / queries_per_mesh);
  const int f0 = off ? off[mesh] : 0, f1 = off ? off[mesh + 1] : single_F;
  const gq3 p = gq_mk(points[q * 3 + 0], points[q * 3 + 1], points[q * 3 + 2]);
  float best = GQ_INF_F;
  int bi = 0x7fffffff;
  for (int f = f0 + lane; f < f1; f += GQ_WAVE) {
    const GqFace fc = rec[f];
    const gq3 d = p - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z);
    const float d2 = gq_tri_dist2(fc, d);
    if (d2 < best) {
      best = d2;
      bi = f;
    }
  }
  unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned int)bi;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(key, o, GQ_WAVE);
    key = other < key ? other : key;
  }
  const int face = (int)(key & 0xffffffffu);
  if (lane == 0) {
    GqSdfOut o;
    if (face >= f0 && face < f1) {
      o = gq_tri_finish(rec[face], p);
    } else {  // empty mesh or all-NaN distances
      o.dist2 = GQ_INF_F;
      o.sign = 1;
      o.normal = gq_mk(0, 0, 0);
      o.closest = p;
    }
    dist_sq[q] = o.dist2;
    sign[q] = o.sign;
    if (normal) {
      normal[q * 3 + 0] = o.normal.x;
      normal[q * 3 + 1] = o.normal.y;
      normal[q * 3 + 2] = o.normal.z;
    }
    closest[q * 3 + 0] = o.closest.x;
    closest[q * 3 + 1] = o.closest.y;
    closest[q * 3 + 2] = o.closest.z;
  }
}

// ---- point per lane -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gq_sdf_points_kernel(const float* __restrict__ points, int64_t N,
                                                            const GqFace* __restrict__ rec, int F,
                                                            float* __restrict__ dist_sq, int32_t* __restrict__ sign,
                                                            float* __restrict__ normal, float* __restrict__ closest) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = q < N;
  const int64_t qq = ok ? q : 0;
  const gq3 p = gq_mk(points[qq * 3 + 0], points[qq * 3 + 1], points[qq * 3 + 2]);
  float best = GQ_INF_F;
  int bi = 0;
  for (int f = 0; f < F; ++f) {  // f is wave-uniform: the record is fetched once per wave (scalar loads)
    const GqFace fc = rec[f];
    const gq3 d = p - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z);
    const float d2 = gq_tri_dist2(fc, d);
    if (d2 < best) {
      best = d2;
      bi = f;
    }
  }
  if (!ok) return;
  const GqSdfOut o = gq_tri_finish(rec[bi], p);
  dist_sq[q] = o.dist2;
  sign[q] = o.sign;
  if (normal) {
    normal[q * 3 + 0] = o.normal.x;
    normal[q * 3 + 1] = o.normal.y;
    normal[q * 3 + 2] = o.normal.z;
  }
  closest[q * 3 + 0] = o.closest.x;
  closest[q * 3 + 1] = o.closest.y;
  closest[q * 3 + 2] = o.closest.z;
}

// d(dist_sq)/d(points) = 2 (p - closest) * g
__global__ void gq_sdf_bwd_kernel(const float* __restrict__ g, const float* __restrict__ points,
                                  const float* __restrict__ closest, int64_t N, float* __restrict__ grad_points) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 3) return;
  grad_points[i] = 2.0f * (points[i] - closest[i]) * g[i / 3];
}

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
  int B, P, L, D, batch_each;
  float* dis;     // (B, P)
  int32_t* link;  // (B, P)
  float* gvec;    // (B, P, 3)
};

__global__ __launch_bounds__(256) void gq_hand_pen_kernel(GqPenArgs g) {
  const int row = blockIdx.y;
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = pt < g.P;
  const int obj = row / g.batch_each;
  const float* sp = g.surf + ((size_t)obj * g.P + (ok ? pt : 0)) * 3;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  const gq3 xw = gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]);
  const gq3 xh = gq_mtv(R, xw);  // R^T (x - t)
  float best_dis = -GQ_INF_F;
  int best_link = 0;
  gq3 best_g = gq_mk(0, 0, 0);
  for (int l = 0; l < g.L; ++l) {
    const int f0 = g.off[l], f1 = g.off[l + 1];
    if (f1 <= f0) continue;
    const float* T = g.link_T + ((size_t)row * g.L + l) * 12;  // wave-uniform
    const float Rl[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const gq3 tl = gq_mk(T[3], T[7], T[11]);
    const gq3 xl = gq_mtv(Rl, xh - tl);
    float bd = GQ_INF_F;
    int bi = f0;
    for (int f = f0; f < f1; ++f) {
      const GqFace fc = g.rec[f];
      const gq3 d = xl - gq_mk(fc.r0.x, fc.r0.y, fc.r0.z);
      const float d2 = gq_tri_dist2(fc, d);
      if (d2 < bd) {
        bd = d2;
        bi = f;
      }
    }
    const GqSdfOut o = gq_tri_finish(g.rec[bi], xl);
    const float root = sqrtf(o.dist2 + 1e-8f);
    const float dis = root * (float)(-o.sign);
    if (dis > best_dis) {
      best_dis = dis;
      best_link = l;
      // d dis / d x_l = -sign (x_l - c) / sqrt(d^2 + 1e-8); rotate into the hand frame
      const gq3 gl = ((float)(-o.sign) / root) * (xl - o.closest);
      best_g = gq_mv(Rl, gl);
    }
  }
  if (!ok) return;
  const size_t o = (size_t)row * g.P + pt;
  g.dis[o] = best_dis;
  g.link[o] = best_link;
  g.gvec[o * 3 + 0] = best_g.x;
  g.gvec[o * 3 + 1] = best_g.y;
  g.gvec[o * 3 + 2] = best_g.z;
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
};

__global__ __launch_bounds__(256) void gq_hand_pen_bwd_kernel(GqPenBwdArgs g) {
  extern __shared__ float sm[];  // [4 waves][L*6 + 12]
  const int row = blockIdx.x;
  const int tid = threadIdx.x, lane = gq_lane(), wv = tid / GQ_WAVE;
  const int stride = g.L * 6 + 12;
  for (int i = tid; i < 4 * stride; i += 256) sm[i] = 0.0f;
  __syncthreads();
  float* acc = sm + wv * stride;
  const int obj = row / g.batch_each;
  const float* hp = g.hand_pose + (size_t)row * g.D;
  const float* R = g.Rg + (size_t)row * 9;
  for (int base = 0; base < g.P; base += 256) {
    const int pt = base + tid;
    float w = 0.0f;
    int lk = 0;
    gq3 G = gq_mk(0, 0, 0), xh = gq_mk(0, 0, 0);
    if (pt < g.P) {
      const size_t o = (size_t)row * g.P + pt;
      w = g.w[o];
      if (w != 0.0f) {
        lk = g.link[o];
        G = w * gq_mk(g.gvec[o * 3], g.gvec[o * 3 + 1], g.gvec[o * 3 + 2]);
        const float* sp = g.surf + ((size_t)obj * g.P + pt) * 3;
        xh = gq_mtv(R, gq_mk(sp[0] - hp[0], sp[1] - hp[1], sp[2] - hp[2]));
      }
    }
    unsigned long long mask = __ballot(w != 0.0f);
    while (mask) {  // wave-uniform loop over contributing lanes, in lane order
      const int s = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int l = gq_readlane_i(lk, s);
      const gq3 Gs = gq_mk(gq_readlane(G.x, s), gq_readlane(G.y, s), gq_readlane(G.z, s));
      const gq3 xs = gq_mk(gq_readlane(xh.x, s), gq_readlane(xh.y, s), gq_readlane(xh.z, s));
      const gq3 ms = gq_cross(xs, Gs);
      if (lane == 0) {
        float* a = acc + l * 6;
        a[0] -= Gs.x;
        a[1] -= Gs.y;
        a[2] -= Gs.z;
        a[3] -= ms.x;
        a[4] -= ms.y;
        a[5] -= ms.z;
        float* k = acc + g.L * 6;
        k[0] += Gs.x;
        k[1] += Gs.y;
        k[2] += Gs.z;
        k[3] += xs.x * Gs.x;
        k[4] += xs.x * Gs.y;
        k[5] += xs.x * Gs.z;
        k[6] += xs.y * Gs.x;
        k[7] += xs.y * Gs.y;
        k[8] += xs.y * Gs.z;
        k[9] += xs.z * Gs.x;
        k[10] += xs.z * Gs.y;
        k[11] += xs.z * Gs.z;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < stride; i += 256) {
    const float v = ((sm[i] + sm[stride + i]) + sm[2 * stride + i]) + sm[3 * stride + i];
    if (i < g.L * 6) g.wrench[(size_t)row * g.L * 6 + i] = v;
    else g.gRt[(size_t)row * 12 + (i - g.L * 6)] = v;
  }
}

// ---- mesh-set handle: concatenated face records of n_mesh meshes on the device -----------------------------------
struct gqMeshSet {
  GqFace* rec;
  int32_t* off_dev;
  int32_t* off_host;
  int n_mesh;
  int64_t n_faces;
};

extern "C" {

int gq_meshset_create(const float* face_verts_host, const int32_t* face_offset_host, int n_mesh, gqMeshSet** out) {
  GQ_REQUIRE(face_verts_host && face_offset_host && out && n_mesh > 0, "meshset_create: bad arguments");
  const int64_t F = face_offset_host[n_mesh];
  GQ_REQUIRE(F > 0 && face_offset_host[0] == 0, "meshset_create: empty face list");
  for (int i = 0; i < n_mesh; ++i)
    GQ_REQUIRE(face_offset_host[i + 1] >= face_offset_host[i], "meshset_create: offsets must be non-decreasing");
  gqMeshSet* ms = new gqMeshSet();
  ms->n_mesh = n_mesh;
  ms->n_faces = F;
  ms->off_host = new int32_t[n_mesh + 1];
  memcpy(ms->off_host, face_offset_host, sizeof(int32_t) * (n_mesh + 1));
  float* tmp = nullptr;
  GQ_CHECK_HIP(hipMalloc(&tmp, (size_t)F * 9 * 4));
  GQ_CHECK_HIP(hipMalloc(&ms->rec, (size_t)F * sizeof(GqFace)));
  GQ_CHECK_HIP(hipMalloc(&ms->off_dev, sizeof(int32_t) * (n_mesh + 1)));
  GQ_CHECK_HIP(hipMemcpy(tmp, face_verts_host, (size_t)F * 9 * 4, hipMemcpyHostToDevice));
  GQ_CHECK_HIP(hipMemcpy(ms->off_dev, face_offset_host, sizeof(int32_t) * (n_mesh + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gq_face_prep_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, 0, tmp, ms->rec, F);
  GQ_LAUNCH_CHECK();
  GQ_CHECK_HIP(hipDeviceSynchronize());
  GQ_CHECK_HIP(hipFree(tmp));
  *out = ms;
  return GQ_OK;
}

int gq_meshset_destroy(gqMeshSet* ms) {
  if (!ms) return GQ_OK;
  hipFree(ms->rec);
  hipFree(ms->off_dev);
  delete[] ms->off_host;
  delete ms;
  return GQ_OK;
}

int gq_meshset_num_faces(const gqMeshSet* ms, int mesh, int64_t* n) {
  GQ_REQUIRE(ms && n && mesh >= -1 && mesh < ms->n_mesh, "meshset_num_faces: bad arguments");
  *n = mesh < 0 ? ms->n_faces : (ms->off_host[mesh + 1] - ms->off_host[mesh]);
  return GQ_OK;
}

int gq_sdf_workspace_bytes(int64_t n_faces, size_t* bytes) {
  GQ_REQUIRE(bytes && n_faces >= 0, "sdf_workspace_bytes: bad arguments");
  *bytes = (size_t)n_faces * sizeof(GqFace) + 256;
  return GQ_OK;
}

// TorchSDF-contract query against a raw (F,3,3) device triangle soup.
int gq_sdf_forward(const float* points, int64_t n_points, const float* face_verts, int64_t n_faces, float* dist_sq,
                   int32_t* sign, float* normal, float* closest, void* workspace, size_t workspace_bytes,
                   void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_points == 0) return GQ_OK;
  GQ_REQUIRE(points && face_verts && dist_sq && sign && closest && workspace, "sdf_forward: null pointer");
  GQ_REQUIRE(n_points > 0 && n_faces > 0 && n_faces < (1ll << 31), "sdf_forward: bad sizes N=%lld F=%lld",
             (long long)n_points, (long long)n_faces);
  GQ_REQUIRE(workspace_bytes >= (size_t)n_faces * sizeof(GqFace) + 256, "sdf_forward: workspace too small");
  GqFace* rec = (GqFace*)workspace;
  hipLaunchKernelGGL(gq_face_prep_kernel, dim3((unsigned)((n_faces + 255) / 256)), dim3(256), 0, st, face_verts, rec,
                     n_faces);
  GQ_LAUNCH_CHECK();
  if (n_points >= 131072) {
    hipLaunchKernelGGL(gq_sdf_points_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, points,
                       n_points, rec, (int)n_faces, dist_sq, sign, normal, closest);
  } else {
    hipLaunchKernelGGL(gq_sdf_wave_kernel, dim3((unsigned)((n_points + 3) / 4)), dim3(256), 0, st, points, n_points,
                       rec, (const int32_t*)nullptr, (int)n_faces, n_points, dist_sq, sign, normal, closest);
  }
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Query against a mesh set: points (n_mesh * queries_per_mesh, 3); query q uses mesh q / queries_per_mesh
// (object_model.py:217-220: one mesh per object, batch_size_each * n_contact queries each).
int gq_sdf_forward_meshset(const gqMeshSet* ms, const float* points, int64_t n_points, int64_t queries_per_mesh,
                           float* dist_sq, int32_t* sign, float* normal, float* closest, void* stream) {
  if (n_points == 0) return GQ_OK;
  GQ_REQUIRE(ms && points && dist_sq && sign && closest, "sdf_forward_meshset: null pointer");
  GQ_REQUIRE(queries_per_mesh > 0 && n_points == queries_per_mesh * ms->n_mesh,
             "sdf_forward_meshset: n_points=%lld != queries_per_mesh=%lld * n_mesh=%d", (long long)n_points,
             (long long)queries_per_mesh, ms->n_mesh);
  hipLaunchKernelGGL(gq_sdf_wave_kernel, dim3((unsigned)((n_points + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     points, n_points, ms->rec, ms->off_dev, 0, queries_per_mesh, dist_sq, sign, normal, closest);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_sdf_backward(const float* grad_dist_sq, const float* points, const float* closest, int64_t n_points,
                    float* grad_points, void* stream) {
  if (n_points == 0) return GQ_OK;
  GQ_REQUIRE(grad_dist_sq && points && closest && grad_points && n_points > 0, "sdf_backward: bad arguments");
  hipLaunchKernelGGL(gq_sdf_bwd_kernel, dim3((unsigned)((n_points * 3 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, grad_dist_sq, points, closest, n_points, grad_points);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

// Fused hand-penetration query (HandModel.cal_distance, hand_model.py:875-987).
int gq_hand_pen_forward(const gqMeshSet* links, const float* surface_points, int64_t n_obj, int64_t n_surface,
                        int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg, const float* link_T,
                        float* dis, int32_t* link, float* gvec, void* stream) {
  GQ_REQUIRE(links && surface_points && hand_pose && Rg && link_T && dis && link && gvec, "hand_pen_forward: null");
  GQ_REQUIRE(n_obj > 0 && n_surface > 0 && batch_each > 0 && pose_dim >= 9, "hand_pen_forward: bad sizes");
  GqPenArgs a{};
  a.surf = surface_points;
  a.hand_pose = hand_pose;
  a.Rg = Rg;
  a.link_T = link_T;
  a.rec = links->rec;
  a.off = links->off_dev;
  a.B = (int)(n_obj * batch_each);
  a.P = (int)n_surface;
  a.L = links->n_mesh;
  a.D = pose_dim;
  a.batch_each = (int)batch_each;
  a.dis = dis;
  a.link = link;
  a.gvec = gvec;
  GQ_REQUIRE(a.B <= 65535, "hand_pen_forward: B=%d exceeds grid.y limit", a.B);
  hipLaunchKernelGGL(gq_hand_pen_kernel, dim3((unsigned)((a.P + 255) / 256), (unsigned)a.B), dim3(256), 0,
                     (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

int gq_hand_pen_backward(int n_links, const float* surface_points, int64_t n_obj, int64_t n_surface,
                         int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                         const float* grad_dis, const int32_t* link, const float* gvec, float* link_wrench, float* gRt,
                         void* stream) {
  GQ_REQUIRE(surface_points && hand_pose && Rg && grad_dis && link && gvec && link_wrench && gRt,
             "hand_pen_backward: null");
  GQ_REQUIRE(n_links > 0 && n_links <= 256 && n_obj > 0 && n_surface > 0 && batch_each > 0, "hand_pen_backward: sizes");
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
  const size_t shm = (size_t)4 * (n_links * 6 + 12) * sizeof(float);
  hipLaunchKernelGGL(gq_hand_pen_bwd_kernel, dim3((unsigned)a.B), dim3(256), shm, (hipStream_t)stream, a);
  GQ_LAUNCH_CHECK();
  return GQ_OK;
}

}  // extern "C"
