/* graspqp_hip.h -- C ABI of libgraspqp_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the per-MALA*-iteration hot path of leggedrobotics/graspqp.  Every entry point
 * replaces one interface of the (Python) reference; the reference file:line it stands in for is cited.
 *
 * Conventions
 *   - all array arguments are DEVICE pointers unless the name ends in _host; tensors are dense, row-major,
 *     float32 / int32 / int64 exactly as noted; the caller owns every buffer (no ownership transfer);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on it and
 *     graph-capturable (no allocation / synchronisation inside) except the *_create functions;
 *   - return value 0 = ok, otherwise an error code; gq_last_error() returns a thread-local message;
 *     nothing throws across the ABI;
 *   - workspaces are caller-provided; their sizes come from the matching *_workspace_bytes function.
 */
#ifndef GRASPQP_HIP_H
#define GRASPQP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------------- */
int gq_version(void);
const char* gq_last_error(void);
int gq_device_check(int device, char* arch_out, int arch_len);
/* gqTimer: event pair that a launch taking a `timer` argument fills with the kernel's own start/stop timestamps */
int gq_timer_create(void** out);
int gq_timer_elapsed_ms(void* timer, float* ms);
int gq_timer_destroy(void* timer);

/* ---- mesh signed distance: torchsdf.compute_sdf / index_vertices_by_faces ------------------------
 * reference call sites: core/object_model.py:147,220  core/hand_model.py:352,953
 * contract: dist_sq (N) squared distance; sign (N) int32 +1 outside / -1 inside; normal (N,3) unit
 * (p - closest)/|p - closest| (may be NULL); closest (N,3).  Only dist_sq is differentiable, w.r.t. points. */
typedef struct gqMeshSet gqMeshSet; /* n_mesh triangle soups resident on the device */
int gq_meshset_create(const float* face_verts_host /* (sumF,3,3) */, const int32_t* face_offset_host /* (n_mesh+1) */,
                      int n_mesh, gqMeshSet** out);
int gq_meshset_destroy(gqMeshSet* ms);
/* setup-time 32^3 occupancy grid per mesh; enables the penetration_only = 1 fast path of gq_hand_pen_forward */
int gq_meshset_build_occupancy(gqMeshSet* ms);
int gq_meshset_num_faces(const gqMeshSet* ms, int mesh /* -1 = all */, int64_t* n);
int gq_sdf_workspace_bytes(int64_t n_faces, size_t* bytes);
int gq_sdf_forward(const float* points /* (N,3) */, int64_t n_points, const float* face_verts /* (F,3,3) */,
                   int64_t n_faces, float* dist_sq, int32_t* sign, float* normal, float* closest, void* workspace,
                   size_t workspace_bytes, void* stream);
/* object_model.py:217-220: query q uses mesh q / queries_per_mesh (one mesh per object) */
int gq_sdf_forward_meshset(const gqMeshSet* ms, const float* points, int64_t n_points, int64_t queries_per_mesh,
                           float* dist_sq, int32_t* sign, float* normal, float* closest, void* stream);
/* The same query for MANY points against ONE mesh (the per-link calls of HandModel.cal_distance, core/hand_model.py:
 * 914-953: batch * 2500 surface points per call): one query per lane through an implicit 4-ary box hierarchy over the
 * Morton-sorted faces, staged in LDS when it fits (csrc/bvh.hip).  Exact (same winner rule as gq_sdf_forward: smallest
 * ranking distance, ties to the smallest face index; the winner is finished exactly).  gq_bvh_create takes HOST triangles
 * (n_faces,3,3), 1 <= n_faces <= 65536.                                                                                 */
typedef struct gqBvh gqBvh;
int gq_bvh_create(const float* face_verts_host, int64_t n_faces, gqBvh** out);
int gq_bvh_destroy(gqBvh* bvh);
int gq_debug_set_bvh_sorted(int on); /* 1: the queries of a 2048-point chunk are ordered by direction inside the block before
                                        the traversal (coherent wavefronts; measured slower, A/B runs); 0 (default): plain order */
int gq_sdf_forward_bvh(const gqBvh* bvh, const float* points, int64_t n_points, float* dist_sq, int32_t* sign,
                       float* normal /* or NULL */, float* closest, void* stream);
int gq_sdf_backward(const float* grad_dist_sq, const float* points, const float* closest, int64_t n_points,
                    float* grad_points, void* stream);

/* ---- box-constrained QP: qpth.qp.QPFunction as used by SQPLsqSolver.solve --------------------------
 * reference: metrics/solver/qp_solver.py:8,60-134 (QPFunction(maxIter=12, eps=5e-2), G = [I;-I], h = [u;-l]).
 * lam / slack are (B, 2 nz): upper-bound block then lower-bound block.  lower/upper may be NULL (scalars used).
 * gq_lsq_*: Q = A'A + ridge I, p = -A'b with A (B,m,nz), m <= 8, b (B,m) or NULL (= 0); nz <= 128 (dense Q: <= 64).         */
int gq_boxqp_workspace_bytes(int64_t batch, int nz, int max_iter, size_t* bytes);
int gq_boxqp_forward(const float* Q /* (B,nz,nz) */, const float* p /* (B,nz) or NULL */, const float* lower,
                     const float* upper, float lower_s, float upper_s, int64_t batch, int nz, float eps, int max_iter,
                     int not_improved_lim, float* x, float* lam, float* slack, int32_t* best_iter /* (B) or NULL */,
                     int32_t* n_iter /* (1) or NULL */, void* workspace, size_t workspace_bytes, void* stream);
int gq_boxqp_backward(const float* Q, const float* lam, const float* slack, const float* grad_x, int64_t batch, int nz,
                      float* dx /* (B,nz) = grad_p; grad_Q = (dx x' + x dx')/2 */,
                      float* dlam /* (B,2nz); grad_h = -dlam */, void* stream);
int gq_lsq_boxqp_forward(const float* A, const float* b, const float* lower, const float* upper, float lower_s,
                         float upper_s, int64_t batch, int m, int nz, float ridge, float eps, int max_iter,
                         int not_improved_lim, float* x, float* lam, float* slack, int32_t* best_iter, int32_t* n_iter,
                         void* workspace, size_t workspace_bytes, void* stream);
int gq_lsq_boxqp_backward(const float* A, const float* lam, const float* slack, const float* grad_x, int64_t batch,
                          int m, int nz, float ridge, float* dx, float* dlam, void* stream);
/* qpth's batch-global stopping rule (qpth/solvers/pdipm/batch.py forward loop) on the (B, max_iter)
 * tables of per-iteration residuals and mu that the forward kernels record: kstar[0] = index of the last iteration
 * whose record counts, kstar[1] = *n_iter = kstar[0] + 1.  Stop at the first iteration where no row improved its
 * running-best residual for `not_improved_lim` iterations in a row, or max_rows(best residual) < eps, or
 * min_rows(mu) > 1e32; NaN residuals never improve a row.  The forward entry points call exactly this.            */
int gq_boxqp_stop_rule(const float* resid /* (B,max_iter) */, const float* mu /* (B,max_iter) */, int64_t batch,
                       int max_iter, float eps, int not_improved_lim, float* runmin_scratch /* (B) */,
                       int32_t* kstar /* (2) */, int32_t* n_iter /* (1) or NULL */, void* stream);

/* ---- force-closure energy: energy_fnc for energy_type "graspqp" ------------------------------------
 * reference: metrics/ops/span.py:263-295,313-415  metrics/ops/registry.py:31-89
 * E_fc = values_gain (1/2 |F x|^2 + 0.01) exp(-svd_gain (prod sigma(F))^(1/6)), 1 <= x <= max_limit + 1.
 * cog is (B,3).  gq_fc_backward must follow gq_fc_forward on the same workspace.                           */
int gq_fc_workspace_bytes(int64_t batch, int n_contact, int n_cone, int max_iter, size_t* bytes);
int gq_fc_forward(const float* contact_pts, const float* contact_normals, const float* cog, int64_t batch,
                  int n_contact, int n_cone, float friction, float torque_weight, float max_limit, float svd_gain,
                  float values_gain, float eps, int max_iter, float* e_fc /* (B) */,
                  float* x_sum /* (B,n_contact) or NULL */, int32_t* n_iter, void* workspace, size_t workspace_bytes,
                  void* stream);
int gq_fc_backward(const float* contact_pts, const float* contact_normals, const float* cog, const float* grad_e,
                   int64_t batch, int n_contact, int n_cone, float friction, float torque_weight, float svd_gain,
                   float values_gain, int accumulate /* 1: grad_contact_pts += */, float* grad_contact_pts,
                   void* workspace, size_t workspace_bytes, void* stream);
int gq_fc_peek(void* workspace, size_t workspace_bytes, int64_t batch, int n_contact, int n_cone, const float** F,
               const float** x, const float** val, const float** svd);
/* Fused form of gq_contact_terms + gq_fc_forward + gq_fc_backward for constant upstream weights (the MALA* loop,
 * scripts/fit.py:434-438: w_dis on E_dis, w_fc on E_fc): two launches per iteration instead of nine, the grasp matrix
 * stays in registers between the cone construction and the QP iterations, and qpth's batch-global stop rule is
 * replayed inside the second kernel (batches <= 256 rows) or applied by the first kernel's last block to per-block
 * aggregates (larger batches) -- no launch of its own either way.  Inputs as gq_contact_terms; g_contact_pts receives
 * w_dis dE_dis/dp + w_fc dE_fc/dp, g_hand_normals w_dis dE_dis/dnH.  Workspace: gq_fc_workspace_bytes; gq_fc_peek
 * works afterwards.  ZERO-FILL THE WORKSPACE ONCE after allocating it (hipMemset): the block counter of the
 * large-batch stop rule lives in it and wraps back to zero at the end of every launch.                            */
int gq_fc_step(const float* dist_sq, const int32_t* sign, const float* obj_dir, const float* closest,
               const float* contact_pts, const float* hand_normals, const float* cog, int64_t batch, int n_contact,
               int n_cone, float friction, float torque_weight, float max_limit, float svd_gain, float values_gain,
               float eps, int max_iter, float w_dis, float w_fc, float* obj_normal, float* g_contact_pts,
               float* g_hand_normals, float* e_fc, float* x_sum, int32_t* n_iter, void* workspace,
               size_t workspace_bytes, void* stream);

/* ---- the reference's other force-closure energies (scripts/fit.py:343-347, --energy_type dexgrasp | tdg) -------------
 * Same inputs as the graspqp energy: contact points, OBJECT normals at the contacts (constants), cog (B,3).  One launch
 * gives the energy and its gradient w.r.t. the contact points: g_contact_pts (+)= upstream * dE/dp with upstream =
 * grad_e[row] if grad_e != NULL else w; accumulate = 1 adds to g_contact_pts.  e_fc or g_contact_pts may be NULL.
 * gq_dexgrasp_energy: metrics/ops/dexgrasp.py:4-34,  E = |sum_i [n_i ; torque_weight (n_i x (p_i - cog))]|^2.
 * gq_tdg_energy: metrics/ops/tdg.py:147-239 (TDGEnergy.forward behind TDGSpanMetric): directions (P,3) = the force part of
 *   target_direction_6D (unit vectors; the torque part is zero), friction = miu_coef[0], obb_length = obj_obb_length,
 *   scale = the factor 100 of TDGSpanMetric.forward.                                                                */
int gq_dexgrasp_energy(const float* contact_pts, const float* contact_normals, const float* cog, int64_t batch,
                       int n_contact, float torque_weight, const float* grad_e /* (B) or NULL */, float w, int accumulate,
                       float* e_fc /* (B) or NULL */, float* g_contact_pts /* (B,n,3) or NULL */, void* stream);
int gq_tdg_energy(const float* contact_pts, const float* contact_normals, const float* cog, const float* directions,
                  int n_directions, int64_t batch, int n_contact, float friction, float obb_length, int enable_density,
                  float scale, const float* grad_e, float w, int accumulate, float* e_fc, float* g_contact_pts,
                  void* stream);

/* The same step with the hand-penetration branch of the iteration (gq_hand_pen_forward with penetration_only = 1, then
 * gq_hand_pen_backward in its fused-E_pen form) running in the SAME two launches: the two branches are independent
 * until gq_fk_backward, neither fills the GPU on its own at batch 256, and one grid holding both roles overlaps them
 * without a cross-stream dependency.  Fields = the parameters of gq_fc_step / gq_hand_pen_forward /
 * gq_hand_pen_backward of the same names.                                                                      */
typedef struct gqHand gqHand; /* declared with gq_hand_create below */
typedef struct gqPointGrid gqPointGrid; /* gq_pointgrid_create, below */
typedef struct gqFcStepDesc {
  const float* dist_sq; const int32_t* sign; const float* obj_dir; const float* closest;
  const float* contact_pts; const float* hand_normals; const float* cog;
  int64_t batch; int32_t n_contact; int32_t n_cone;
  float friction, torque_weight, max_limit, svd_gain, values_gain, eps; int32_t max_iter; float w_dis, w_fc;
  float* obj_normal; float* g_contact_pts; float* g_hand_normals; float* e_fc; float* x_sum; int32_t* n_iter;
  void* workspace; size_t workspace_bytes;
} gqFcStepDesc;
typedef struct gqPenStepDesc {
  const gqMeshSet* links; const float* surface_points; int64_t n_obj; int64_t n_surface; int64_t batch_each;
  const float* hand_pose; int32_t pose_dim; const float* Rg; const float* link_T;
  float* dis; int32_t* link; float* gvec;       /* link / gvec zero-initialised by the caller, see gq_hand_pen_forward */
  float* link_wrench; float* gRt; float w_pen; float* e_pen;
  uint64_t* span; uint64_t* span_acc;           /* optional in-kernel timing of the query, see gq_hand_pen_backward */
  /* optional third role of the second launch: world sphere centres + self penetration (gq_self_pen_forward on the
   * centres of link_T), so that gq_fk_forward can be called without spheres; hand == NULL: absent               */
  const gqHand* hand; float w_spen; float* e_spen; float* g_sphere_centers; float* sphere_centers /* or NULL */;
  const gqPointGrid* grid;                      /* optional: the query role runs link-driven (gq_hand_pen_forward_cells) */
  const float* patch_spheres;                   /* optional: gq_surface_patches (see gq_hand_pen_forward) */
} gqPenStepDesc;
int gq_fc_pen_step(const gqFcStepDesc* fc, const gqPenStepDesc* pen, void* stream);
/* gq_fc_pen_step for the reference's other energy types (scripts/fit.py:343-347; metrics/ops/dexgrasp.py:4-34,
 * metrics/ops/tdg.py:147-239): the contact terms of E_dis (gq_contact_terms) and gq_dexgrasp_energy / gq_tdg_energy (with
 * upstream weight w_fc, accumulated onto w_dis dE_dis/dp) as the first role of the first launch, beside the penetration
 * query; penetration backward (+ self penetration) in the second.  energy: 1 = dexgrasp, 2 = tdg; the remaining fields are
 * the parameters of the same names of gq_contact_terms / gq_dexgrasp_energy / gq_tdg_energy.  Same bits as the separate
 * launches.                                                                                                          */
typedef struct gqAltFcDesc {
  const float* dist_sq; const int32_t* sign; const float* obj_dir; const float* closest;
  const float* contact_pts; const float* hand_normals; const float* cog;
  int64_t batch; int32_t n_contact; int32_t energy;
  float torque_weight;                                            /* dexgrasp (0 at the reference's call site) */
  const float* directions; int32_t n_directions; float friction, obb_length; int32_t enable_density; float scale; /* tdg */
  float w_dis, w_fc;
  float* obj_normal; float* g_contact_pts; float* g_hand_normals; float* e_fc;
} gqAltFcDesc;
int gq_alt_pen_step(const gqAltFcDesc* alt, const gqPenStepDesc* pen, void* stream);

/* ---- hand kinematics: HandModel.set_parameters / fk / _set_contact_idxs -----------------------------
 * reference: core/hand_model.py:762-766,787-873,1220-1267  utils/transforms.py:5-13
 * The hand is described by a reduced kinematic tree (fixed joints folded, see graspqp_amd/hands/spec.py).
 * hand_pose (B, 9 + JA) = [t(3), rot6d(6), theta_actuated]; transforms are 3x4 row-major [R|t].  JA = number of
 * actuated joints (= n_dofs unless the hand is coupled, see gqHandDesc.n_actuated).                              */
typedef struct gqHandDesc { /* all pointers HOST */
  int32_t n_dofs, n_links, n_cand, n_spheres;
  const int32_t* node_parent; /* (J) parent node, -1 = base; parents precede children */
  const int32_t* node_type;   /* (J) 1 revolute, 2 prismatic */
  const float* node_pre;      /* (J,12) fixed transform parent node frame -> joint frame */
  const float* node_axis;     /* (J,3) unit axis in the joint frame */
  const int32_t* link_node;   /* (L) node a mesh link rides on, -1 = base */
  const float* link_offset;   /* (L,12) node frame -> link frame */
  const float* cand_pos;      /* (C,3) contact candidates, link frame */
  const float* cand_nrm;      /* (C,3) */
  const int32_t* cand_link;   /* (C) */
  const float* sphere;        /* (S,4) penetration spheres x y z r, link frame */
  const int32_t* sphere_link; /* (S) non-decreasing */
  const float* joints_lower;  /* (JA) limits of the ACTUATED joints (JA = n_actuated, or J when n_actuated = 0) */
  const float* joints_upper;  /* (JA) */
  /* coupled hands (reference hands/{ability_hand,panda,schunk}.py: joint_filter + joint_calc_fnc / jacobian_fnc): the pose
   * carries n_actuated joint values, the J tree joints follow theta_tree = coupling theta_actuated + coupling_offset.
   * n_actuated = 0 (or coupling = NULL): every tree joint is actuated, pose dimension 9 + J.                        */
  int32_t n_actuated;
  const float* coupling;        /* (J, n_actuated) row-major, host, or NULL */
  const float* coupling_offset; /* (J) host, or NULL */
} gqHandDesc;
typedef struct gqHand gqHand;
int gq_hand_create(const gqHandDesc* desc, gqHand** out);
int gq_hand_destroy(gqHand* h);
/* Optional head of gq_fk_forward / tail of gq_fk_backward: MalaStar.try_step and MalaStar.accept_step
 * (core/optimizer.py:199-273, 289-340; parameters as gq_mala_propose / gq_mala_accept) run in the same wavefront as
 * the row's kinematics, so an iteration needs no launch of its own for them.  u_switch / new_idx / u_accept hold
 * `slots` iterations of random draws, (slots,B,n) / (slots,B,n) / (slots,B); slot_ctr (2 x int32, device, zeroed
 * once) selects the current slot and is advanced on the device, so that the whole iteration can be replayed from a
 * hipGraph: the host refills the buffers every `slots` iterations.                                               */
typedef struct gqProposeDesc {
  const float* hand_pose;      /* (B,D) accepted pose; the proposal goes to gq_fk_forward's hand_pose argument      */
  const float* grad;           /* (B,D) */
  const int64_t* contact_idx;  /* (B,n) accepted indices; the proposal goes to gq_fk_forward's contact_idx argument */
  const float* u_switch; const int64_t* new_idx;
  float* ema; int64_t* step; float* step_size_out /* (B) or NULL */; float* g2_scratch /* (D) */;
  const float* energy /* (B) or NULL */; int64_t batch_each; float* z_out;
  float step_size; int32_t stepsize_period; float decay, mu, switch_possibility; int32_t clip_grad;
  int32_t* slot_ctr; int32_t slots;
} gqProposeDesc;
/* Optional: the object SDF of the row's contact points (gq_sdf_forward_meshset on contact_points, queries_per_mesh =
 * batch_size * n_contact) answered in the same launch by extra wavefronts of the row's block.                   */
typedef struct gqSdfDesc {
  const gqMeshSet* meshes; int64_t queries_per_mesh;
  float* dist_sq; int32_t* sign; float* obj_dir; float* closest;
} gqSdfDesc;
typedef struct gqAcceptDesc {
  const float* u_accept; const float* z; const uint8_t* reset_mask; const int64_t* step;
  float starting_temperature, decay; int32_t annealing_period;
  float* energy; float* pose; int64_t* idx; float* grad; uint8_t* accept; float* temperature;
  int32_t n_terms; const float* terms_new; float* terms;
  int32_t* slot_ctr; int32_t slots;
} gqAcceptDesc;
int gq_fk_workspace_bytes(const gqHand* h, int64_t batch, size_t* bytes);
int gq_fk_forward(const gqHand* h, const float* hand_pose, const int64_t* contact_idx /* (B,n) */, int64_t batch,
                  int n_contact, float* Rg /* (B,9) */, float* link_T /* (B,L,12) */, float* contact_points /* (B,n,3) */,
                  float* contact_normals /* (B,n,3) */, float* sphere_centers /* (B,S,3) or NULL */,
                  float spen_scale, float* e_spen /* (B) or NULL: gq_self_pen_forward fused in */,
                  float* g_sphere_centers /* (B,S,3), with e_spen: spen_scale * dE_spen/dcentre */,
                  const gqProposeDesc* propose /* NULL, or: hand_pose / contact_idx are first WRITTEN by the proposal */,
                  const gqSdfDesc* sdf /* NULL, or the object SDF of the contact points in the same launch */,
                  void* workspace, size_t workspace_bytes, void* stream);
/* Optional tail of gq_fk_backward: E_dis, E_joints (with its gradient) and the weighted total of one row
 * (core/energy.py:25-28,47-54; scripts/fit.py:434-438), so the iteration needs no separate reduction launch.     */
typedef struct gqRowEnergyDesc {
  const float* dist_sq;      /* (B,n) object SDF of the contact points (gq_sdf_forward*)  */
  const int32_t* sign;       /* (B,n)                                                      */
  const float* obj_dir;      /* (B,n,3) unit (p - closest)/|.|                             */
  const float* hand_normals; /* (B,n,3) world contact normals of the hand (gq_fk_forward)  */
  const float* joints_lower; /* (J) */
  const float* joints_upper; /* (J) */
  const float* e_fc;         /* (B) */
  const float* e_pen;        /* (B) */
  const float* e_spen;       /* (B) */
  int32_t n;                 /* contacts per row */
  float w_dis, w_fc, w_pen, w_spen, w_joints;
  float* e_dis;              /* (B) out */
  float* e_joints;           /* (B) out */
  float* total;              /* (B) out: sum_k w_k E_k */
} gqRowEnergyDesc;
/* analytic backward (replaces autograd through pytorch_kinematics); the workspace must be the one written by
 * gq_fk_forward for the same hand_pose.  Any gradient input may be NULL.  g_link_wrench (B,L,6) = (f, m about the
 * hand origin) in the hand frame and g_Rt (B,12) come from gq_hand_pen_backward.  energy: NULL or see above.    */
int gq_fk_backward(const gqHand* h, const float* hand_pose, const int64_t* contact_idx, int64_t batch, int n_contact,
                   const float* Rg, const float* link_T, const float* g_contact_points, const float* g_contact_normals,
                   const float* g_sphere_centers, const float* g_link_wrench, const float* g_Rt, const float* g_theta,
                   const float* g_R, float* grad_pose /* (B,9+J) */, const gqRowEnergyDesc* energy,
                   const gqAcceptDesc* accept /* NULL, or the Metropolis test on energy->total + state merge */,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- export-time kinematics: scripts/fit.py:224-300 (export_poses) --------------------------------------------
 * Explicit geometric Jacobians in the hand frame.  `workspace` is the FK workspace written by gq_fk_forward for the
 * same poses (it holds the per-joint frames), link_T the link transforms of that call.
 * gq_link_jacobian: HandModel.jacobian (core/hand_model.py:772-777 -> the pytorch_kinematics fork's tree
 *   Chain.jacobian, or the hand's jacobian_fnc for coupled hands): (B,L,6,JA) = [J_v ; J_w] of every mesh link at its
 *   frame origin, columns = actuated joints.
 * gq_contact_jacobian: the linear contact Jacobian J_v + J_w x r of hand_model.py:1176-1196, (B,n,3,J).
 * gq_joint_velocities: HandModel.get_req_joint_velocities (hand_model.py:1155-1218, coupled form): theta =
 *   pinv(J) d with the damped pseudo-inverse of hand_model.py:46-54 (lambda = 1e-3 there), J (B,m,n_dofs) with m = 3 n,
 *   directions (B,m) in the WORLD frame when Rg is given (they are rotated into the hand frame, :1166) else in the hand
 *   frame; residual (B,m) = (J theta - d)^2, ee_vel (B,m) = J theta rotated back to the world frame.  n_dofs <= 64.
 * gq_root_pose_wxyz: (B,7) = [t, unit quaternion w x y z] of hand_pose[:, :9] (fit.py:260-263).                    */
int gq_link_jacobian(const gqHand* h, int64_t batch, const float* link_T /* (B,L,12) */, float* jac /* (B,L,6,JA) */,
                     const void* workspace, size_t workspace_bytes, void* stream);
int gq_contact_jacobian(const gqHand* h, const int64_t* contact_idx /* (B,n) */, int64_t batch, int n_contact,
                        const float* link_T, float* jac /* (B,n,3,JA) */, const void* workspace, size_t workspace_bytes,
                        void* stream);
/* d (sum_ij G_ij . J_ij) / d theta for the contact Jacobian above: grad_jac (B,n,3,JA) -> grad_theta (B,JA), the kinematic
 * Hessian of the tree in closed form (what autograd through the Jacobian gives the reference for E_manipulativity,
 * core/energy.py:80-87 via hand_model.py:1155-1218).  Same link_T / workspace as gq_contact_jacobian.                     */
int gq_contact_jacobian_backward(const gqHand* h, const int64_t* contact_idx /* (B,n) */, int64_t batch, int n_contact,
                                 const float* link_T, const float* grad_jac /* (B,n,3,JA) */, float* grad_theta /* (B,JA) */,
                                 const void* workspace, size_t workspace_bytes, void* stream);
int gq_joint_velocities(const float* jac, const float* directions, const float* Rg /* (B,9) or NULL */, int64_t batch,
                        int m, int n_dofs, float damping, float* theta /* (B,n_dofs) */, float* residual /* or NULL */,
                        float* ee_vel /* or NULL */, void* stream);
int gq_root_pose_wxyz(const float* hand_pose, int64_t batch, int pose_dim, float* root_pose /* (B,7) */, void* stream);

/* ---- hand penetration: HandModel.cal_distance (E_pen) --------------------------------------------------
 * reference: core/hand_model.py:875-987, core/energy.py:57-62.  links = mesh set of the L link meshes.
 * dis (B,P) = max over links of sqrt(d^2 + 1e-8) * (-sign); link (B,P) argmax; gvec (B,P,3) = d dis / d x_h.
 * penetration_only = 0: dis exact everywhere.
 * penetration_only = 1: what E_pen uses (energy.py:59-61 zeroes dis <= 0): dis is exact where it is > 0 and -1e30
 *   elsewhere; link / gvec are WRITTEN ONLY where dis > 0 (pass zero-initialised buffers).  Per link a point is looked
 *   up in the 32^3 voxel grid of the link mesh (gq_meshset_build_occupancy) and only the voxel's candidate faces are
 *   ranked.  = 2: AABB culling only, = 3: occupancy grid + load-balancing queues (needs the workspace) -- both kept
 *   for A/B tests; they write link / gvec everywhere.                                                          */
int gq_hand_pen_forward(const gqMeshSet* links, const float* surface_points /* (n_obj,P,3) */, int64_t n_obj,
                        int64_t n_surface, int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                        const float* link_T, int penetration_only, float* dis, int32_t* link, float* gvec,
                        void* workspace /* NULL, or gq_hand_pen_workspace_bytes (ZEROED before first use): load-balanced path */,
                        size_t workspace_bytes, void* timer /* gqTimer or NULL */,
                        uint64_t* span /* NULL, or {min start, max end} in 100 MHz device ticks, pre-set to {~0, 0} */,
                        const float* patch_spheres /* NULL, or gq_surface_patches of surface_points: lets every block of
                                                      the penetration_only = 1 query drop the links out of reach first */,
                        void* stream);
/* Bounding sphere (centre xyz, radius) of every 256-point slice of every object's surface points, (n_obj, ceil(P/256), 4);
 * set-up time.  Surface points in Morton order (graspqp_amd.utils.meshes.surface_points) give compact slices.        */
int gq_surface_patches(const float* surface_points, int64_t n_obj, int64_t n_surface, float* patch_spheres, void* stream);
int gq_hand_pen_workspace_bytes(int64_t batch, int64_t n_surface, int n_links, size_t* bytes);
/* The same penetration-only query (penetration_only = 1: dis exact where > 0, -1e30 elsewhere; link / gvec written only
 * where dis > 0) driven by the LINKS: a coarse uniform grid over every object's surface points (gqPointGrid, set-up
 * time; cells_per_axis 0 = default 8) lets a row's block test only the points filed under the cells a link's box can
 * touch, so the work follows the overlaps instead of points x links.  Same dis / link / gvec as gq_hand_pen_forward.  */
typedef struct gqPointGrid gqPointGrid;
int gq_pointgrid_create(const float* surface_points_host /* (n_obj,P,3) */, int64_t n_obj, int64_t n_surface,
                        int cells_per_axis, gqPointGrid** out);
int gq_pointgrid_destroy(gqPointGrid* grid);
int gq_hand_pen_forward_cells(const gqMeshSet* links, const gqPointGrid* grid, const float* surface_points, int64_t n_obj,
                              int64_t n_surface, int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                              const float* link_T, float* dis, int32_t* link, float* gvec, void* timer /* gqTimer or NULL */,
                              uint64_t* span /* as gq_hand_pen_forward */, void* stream);
/* diagnostics (NULL = off): 12 device words.  Stand-alone gq_hand_pen_forward (penetration_only = 1) adds [4] (point,
 * link) pairs that reach candidate evaluation, [5] executed point-triangle rankings, [6] pairs ranked inline because the
 * block's LDS lists were full, [7] blocks, [8] (wavefront, link) bounding-sphere tests executed, [9] of those with a point
 * inside the sphere, [10] (point, link) pairs inside the link box, [11] scanning wavefronts; gq_sdf_forward_meshset adds
 * [0] 64-face cluster visits, [1] queries, sets [2] = max visits of a query, adds [3] queries with > 16 visits.  Not
 * read by the fused launches.                                                                                      */
int gq_debug_set_pen_counters(uint64_t* counters /* device, 12 words, or NULL */);
/* A/B switch: 1 = plain block -> query mapping in gq_sdf_forward_meshset; 0 (default) = XCD-aware (with >= 8 meshes the
 * queries of mesh m run on the blocks b with b % 8 == m % 8, i.e. on one XCD, so each L2 holds only its own meshes). */
int gq_debug_set_sdf_mapping(int plain);
/* clusters taken up per round by the stand-alone mesh-distance kernel: 0 = default (4), 2 / 4 forced (A/B runs; results
 * do not depend on it).                                                                                             */
int gq_debug_set_sdf_topk(int topk);
/* LDS list capacities of the stand-alone hand-penetration query: 0 = by launch size (default), 1 / 2 / 3 = 512 / 256 /
 * 128 entries per block (A/B runs; results do not depend on it).                                                  */
int gq_debug_set_pen_caps(int mode);
/* surface points per thread of the hand-penetration query: 0 = defaults (2 as a role of gq_fc_pen_step, 1 in
 * gq_hand_pen_forward), 1 / 2 forced for both (A/B runs; results do not depend on it).                              */
int gq_debug_set_pen_ppt(int ppt);
/* grad_dis (B,P) = upstream d E / d dis.  grad_dis == NULL selects the fused E_pen form: the weights are
 * w_pen * [dis > 0] and e_pen (B) = sum_j relu(dis_j) is written as well (core/energy.py:59-61).
 * span / span_acc (optional): the 64 x {min start, max end} shards filled by gq_hand_pen_forward are folded into
 * span_acc = {sum of launch spans, launches} (100 MHz ticks) and re-armed, so a hipGraph replay can time the query. */
int gq_hand_pen_backward(int n_links, const float* surface_points, int64_t n_obj, int64_t n_surface,
                         int64_t batch_each, const float* hand_pose, int pose_dim, const float* Rg,
                         const float* grad_dis /* (B,P) or NULL */, const int32_t* link, const float* gvec,
                         float* link_wrench /* (B,L,6) */, float* gRt /* (B,12) */, const float* dis /* (B,P) */,
                         float w_pen, float* e_pen /* (B) */, uint64_t* span, uint64_t* span_acc, void* stream);

/* ---- self penetration: HandModel.self_penetration (E_spen), core/hand_model.py:989-1040 ------------------ */
int gq_self_pen_forward(const gqHand* h, const float* sphere_centers /* (B,S,3) world */, int64_t batch,
                        float grad_scale, float* e_spen /* (B) */,
                        float* g_centers /* (B,S,3) grad_scale * dE/dcentre */, void* stream);

/* The same term straight from the kinematics of gq_fk_forward (world sphere centres from link_T / Rg / the translation in
 * hand_pose, then the pair scan): the self-penetration role of gq_fc_pen_step as a launch of its own, bit-identical to
 * it and to the e_spen tail of gq_fk_forward.  For large batches, where it rides on the penetration branch of the
 * iteration instead of lengthening the FK forward launch.  sphere_centers (B,S,3) may be NULL.                     */
int gq_spheres_self_pen(const gqHand* h, const float* hand_pose, int pose_dim, const float* Rg, const float* link_T,
                        int64_t batch, float w_spen, float* sphere_centers, float* e_spen /* (B) */,
                        float* g_sphere_centers /* (B,S,3) w_spen * dE/dcentre */, void* stream);

/* ---- energy composition: core/energy.py:25-28,47-62 and scripts/fit.py:434-438 ---------------------------- */
int gq_contact_terms(const float* dist_sq, const int32_t* sign, const float* onrm, const float* closest,
                     const float* contact_pts, const float* contact_normals, int64_t batch, int n_contact, float w_dis,
                     float* obj_normal /* (B,n,3) = onrm*sign */, float* g_contact_pts, float* g_contact_normals,
                     void* stream);
int gq_row_energy(const float* dist_sq, const int32_t* sign, const float* onrm, const float* contact_normals,
                  const float* hand_pose, const float* joints_lower, const float* joints_upper, const float* e_fc,
                  const float* pen_dis, const float* e_spen, int64_t batch, int n_contact, int n_dofs,
                  int64_t n_surface, float w_dis, float w_fc, float w_pen, float w_spen, float w_joints, float* e_dis,
                  float* e_joints, float* e_pen, float* total, float* g_theta /* (B,J) */, float* g_pen /* (B,P) */,
                  void* stream);
/* The same terms one by one, each with its derivative, for the autograd route of the class surface (calculate_energy on
 * HandModel / ObjectModel): the derivative is written by the forward launch and the backward is a broadcast multiply with
 * the upstream row gradient -- one launch per term where the reference's torch expressions issue a dozen.
 *   gq_signed_distance: ObjectModel.cal_distance, core/object_model.py:222-227 -- distance = sqrt(dist_sq + 1e-8) * (-sign),
 *                       normal_out = normal_in * sign, g_dist_sq = d distance / d dist_sq; n = number of queries.
 *   gq_energy_dis:      core/energy.py:25-28 -- "gendexgrasp": e = sum_j exp(1 - (-obj_normal . hand_normal)) |distance|,
 *                       g_distance = d e / d distance, g_hand_normal = d e / d hand_normal; obj_normal == NULL selects the
 *                       "dexgraspnet" form e = sum_j |distance| (hand_normal / g_hand_normal unused).
 *   gq_energy_joints:   core/energy.py:47-52 -- e = sum relu(theta - upper) + relu(lower - theta) over the last n_dofs
 *                       columns of hand_pose; g_hand_pose (B, pose_dim) = d e / d hand_pose (zero in the root columns).
 *   gq_energy_pen:      core/energy.py:58-61 -- e = sum_p where(distances <= 0, 0, distances).                            */
int gq_signed_distance(const float* dist_sq, const int32_t* sign, const float* normal_in, int64_t n, float* distance,
                       float* normal_out, float* g_dist_sq, void* stream);
int gq_energy_dis(const float* distance /* (B,n) */, const float* obj_normal /* (B,n,3) or NULL */,
                  const float* hand_normal /* (B,n,3) */, int64_t batch, int n_contact, float* e_dis /* (B) */,
                  float* g_distance /* (B,n) */, float* g_hand_normal /* (B,n,3) */, void* stream);
int gq_energy_joints(const float* hand_pose, const float* joints_lower, const float* joints_upper, int64_t batch, int pose_dim,
                     int n_dofs, float* e_joints /* (B) */, float* g_hand_pose /* (B,pose_dim) */, void* stream);
int gq_energy_pen(const float* distances /* (B,P) */, int64_t batch, int64_t n_surface, float* e_pen /* (B) */, void* stream);
int gq_axpy(float* y, const float* x, float a, int64_t n, void* stream);
int gq_scale(float* y, const float* x, float a, int64_t n, void* stream);
int gq_fill(float* y, float a, int64_t n, void* stream);

/* ---- (re-)initialisation: initialize_convex_hull, core/initializations.py:15-193 (scripts/fit.py:315,408-422) --------
 * Per object: samples_per_object points on its convex hull (area-weighted), pushed out by `inflate` (0.01 in the
 * reference) along the face normal; farthest-point sampling of batch_each of them (start = sample 0); per row the look-at
 * rotation towards the hull composed with a random roll / pitch / tilt, a random stand-off distance, and joint angles
 * from a truncated normal around default_state (sigma = jitter_strength * joint range).  hand_pose (B, 9 + n_dofs),
 * B = n_obj * batch_each, is written for ALL rows; applying it to the rows of an env_mask is the caller's business
 * (HandModel.set_parameters(env_mask=...), GraspStepper.step_reset).  The hull (triangles oriented outward, set-up time)
 * and every random number are inputs: u_face (n_obj, samples) and u_len (n_obj, samples, 2) pick the samples, u_pose
 * (B,4) = distance / rotate / pitch / tilt and u_joint (B, n_dofs), all uniform in [0,1).  Contact indices are a plain
 * randint (initializations.py:190-192) and stay with the caller.                                                    */
typedef struct gqInitDesc {
  const float* hull_face_verts;   /* (sumF,3,3) device */
  const float* hull_cdf;          /* (sumF) device: cumulative face area / total area, per object */
  const int32_t* hull_offsets;    /* (n_obj+1) device */
  int64_t n_obj, batch_each, samples_per_object;
  int32_t n_dofs;
  float inflate;
  float forward_axis[3], up_axis[3];                 /* HandModel.forward_axis / up_axis */
  const float* default_state; const float* joints_lower; const float* joints_upper;  /* (n_dofs) device */
  float jitter_strength, distance_lower, distance_upper, rotate_lower, rotate_upper, pitch_lower, pitch_upper,
      tilt_lower, tilt_upper;                         /* scripts/fit.py:59-71 */
  const float* u_face; const float* u_len; const float* u_pose; const float* u_joint;
  float* hand_pose;               /* (B, 9 + n_dofs) out */
  float* shell_points;            /* (B,3) out or NULL: the inflated hull point each row looks at */
  float* shell_dirs;              /* (B,3) out or NULL: unit direction from that point towards the hull */
  void* workspace; size_t workspace_bytes;           /* gq_init_workspace_bytes */
} gqInitDesc;
int gq_init_workspace_bytes(int64_t n_obj, int64_t samples_per_object, int64_t batch_each, size_t* bytes);
int gq_init_convex_hull(const gqInitDesc* desc, void* stream);
/* Surface samples of object meshes on the device (reference core/object_model.py:163-178: pytorch3d
 * sample_points_from_meshes with 100 x num_samples points, then sample_farthest_points(K = num_samples) starting at sample
 * 0): face_verts (sumF,3,3) of n_obj meshes, area_cdf (sumF) normalised cumulative face area per mesh, face_offsets
 * (n_obj+1); u_face (n_obj,M), u_len (n_obj,M,2) uniform draws; points_out (n_obj,n_keep,3).  Workspace:
 * gq_init_workspace_bytes(n_obj, samples_per_object, n_keep).                                                          */
int gq_surface_fps(const float* face_verts, const float* area_cdf, const int32_t* face_offsets, int64_t n_obj,
                   int64_t samples_per_object, int64_t n_keep, const float* u_face, const float* u_len, float* points_out,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- MALA* optimiser: MalaStar.try_step / accept_step, core/optimizer.py:199-273,289-340; fit.py:403-406,454-458
 * random draws are inputs: u_switch (B,n) U[0,1), new_idx (B,n) in [0,C), u_accept (B) U[0,1).               */
int gq_mala_propose(const float* hand_pose, const float* grad, const int64_t* contact_idx, const float* u_switch,
                    const int64_t* new_idx, int64_t batch, int pose_dim, int n_contact, float step_size,
                    int stepsize_period, float decay, float mu, float switch_possibility, int clip_grad,
                    float* ema /* (B,D) in/out */, int64_t* step /* (B) in/out */, float* pose_out, int64_t* idx_out,
                    float* step_size_out /* (B) or NULL */, float* g2_scratch /* (D) */,
                    const float* energy /* (B) or NULL: also emit the per-object z-score of the accepted energies */,
                    int64_t batch_each, float* z_out /* (B) */, void* stream);
int gq_zscore(const float* energy, int64_t n_obj, int64_t batch_each, float* z, void* stream);
int gq_mala_accept(const float* new_energy, const float* u_accept, const float* z, const uint8_t* reset_mask,
                   const int64_t* step, const float* pose_new, const int64_t* idx_new, const float* grad_new,
                   int64_t batch, int pose_dim, int n_contact, float starting_temperature, float decay,
                   int annealing_period, float* energy, float* pose, int64_t* idx, float* grad, uint8_t* accept,
                   float* temperature, int n_terms, const float* terms_new, float* terms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRASPQP_HIP_H */
