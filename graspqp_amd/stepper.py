"""GraspStepper -- one MALA* iteration (reference scripts/fit.py:399-458) as a fixed sequence of HIP launches.

No autograd, no host synchronisation, no allocation inside ``step``: all buffers are created once, so the
whole iteration is capturable into a hipGraph (``capture()``).  It runs exactly the kernels that the autograd
route (``graspqp_amd.core``) runs, in the reference's order:

    propose (+z-score) -> FK+contacts -> { object SDF -> contact terms -> E_fc fwd -> E_fc bwd | hand penetration
    fwd -> bwd (+E_pen) | self penetration } -> FK bwd (+E_dis, E_joints, total) -> accept

State (device tensors): hand_pose (B,D), contact_idx (B,n) i64, grad (B,D), energy (B), ema (B,D), step (B) i64,
terms (5,B) = accepted [E_dis, E_fc, E_pen, E_spen, E_joints].
"""

from __future__ import annotations

import ctypes

import torch

from . import _C, ops

TERM_NAMES = ("E_dis", "E_fc", "E_pen", "E_spen", "E_joints")
DEFAULT_WEIGHTS = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}  # fit.py:51-55


class GraspStepper:
    def __init__(self, hand: ops.HandHandle, object_meshes: ops.MeshSet, surface_points: torch.Tensor, batch_each: int,
                 n_contact: int, weights=None, fc_cfg=None, mala_cfg=None, device="cuda", seed=1,
                 penetration_only: bool = True, energy_type: str = "graspqp", optimizer: str = "mala_star",
                 tdg_directions=None, point_grid: int = 0, split_self_pen: bool = True):
        """energy_type: "graspqp" (default) | "dexgrasp" | "tdg" (scripts/fit.py:343-347); every type has the fused four-
        launch form (the force-closure role of the first stage launch is the contact terms + that energy) and the per-role
        form on two graph branches.  optimizer: "mala_star" | "dexgraspnet"
        (AnnealingDexGraspNet, core/optimizer.py:11-149: no z-score in the temperature, no re-initialisation)."""
        if energy_type not in ("graspqp", "dexgrasp", "tdg") or optimizer not in ("mala_star", "dexgraspnet"):
            raise NotImplementedError(f"energy_type={energy_type!r} / optimizer={optimizer!r}")
        self.energy_type, self.optimizer = energy_type, optimizer
        self.split_self_pen = bool(split_self_pen)  # False: A/B switch, self penetration stays in the FK forward launch
        self.hand, self.objs = hand, object_meshes
        self.dev = torch.device(device)
        self.surf = surface_points.to(self.dev, torch.float32).contiguous()  # (n_obj,P,3)
        self.n_obj, self.P = self.surf.shape[0], self.surf.shape[1]
        self.be, self.n = int(batch_each), int(n_contact)
        self.B = self.n_obj * self.be
        self.J, self.L, self.S = hand.J, hand.L, hand.S
        self.D = 9 + self.J
        self.w = dict(DEFAULT_WEIGHTS)
        if weights:
            self.w.update(weights)
        self.fc = dict(ops.FC_DEFAULTS)
        if fc_cfg:
            self.fc.update(fc_cfg)
        self.mala = dict(switch_possibility=0.4, starting_temperature=18.0, temperature_decay=0.95, annealing_period=30,
                         step_size=0.005, stepsize_period=50, mu=0.98, clip_grad=False)  # fit.py:42-48
        if mala_cfg:
            self.mala.update(mala_cfg)
        self.cog = self.surf.mean(dim=1).repeat_interleave(self.be, dim=0).contiguous()  # object_model.py:64-68
        self.jlo = torch.tensor(hand.spec.joints_lower, device=self.dev)
        self.jhi = torch.tensor(hand.spec.joints_upper, device=self.dev)
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(seed)
        B, D, n, P, L, S = self.B, self.D, self.n, self.P, self.L, max(self.S, 1)
        f = lambda *s: torch.zeros(*s, device=self.dev)
        # state
        self.hand_pose, self.grad, self.ema = f(B, D), f(B, D), f(B, D)
        self.contact_idx = torch.zeros(B, n, dtype=torch.long, device=self.dev)
        self.energy = f(B)
        self.step_count = torch.zeros(B, dtype=torch.long, device=self.dev)
        self.terms = f(5, B)
        # proposal / scratch
        self.pose_new, self.grad_new = f(B, D), f(B, D)
        self.idx_new = torch.zeros(B, n, dtype=torch.long, device=self.dev)
        self.Rg, self.link_T = f(B, 9), f(B, L, 12)
        self.cpts, self.cnrm, self.spheres = f(B, n, 3), f(B, n, 3), f(B, S, 3)
        self.d2, self.onrm, self.closest = f(B, n), f(B, n, 3), f(B, n, 3)
        self.sgn = torch.zeros(B, n, dtype=torch.int32, device=self.dev)
        self.obj_normal, self.g_cpts, self.g_cnrm, self.g_cpts_fc = f(B, n, 3), f(B, n, 3), f(B, n, 3), f(B, n, 3)
        self.pen_dis, self.pen_gvec, self.g_pen = f(B, P), f(B, P, 3), f(B, P)
        self.pen_link = torch.zeros(B, P, dtype=torch.int32, device=self.dev)
        self.wrench, self.gRt = f(B, L, 6), f(B, 12)
        self.e_spen, self.g_sph, self.g_sph_w = f(B), f(B, S, 3), f(B, S, 3)
        self.terms_new, self.total_new = f(5, B), f(B)
        self.g_theta, self.w_fc_vec = f(B, self.J), torch.full((B,), float(self.w["E_fc"]), device=self.dev)
        self.z, self.temperature, self.s_out = f(B), f(B), f(B)
        self.accept = torch.zeros(B, dtype=torch.uint8, device=self.dev)
        self.g2 = f(D)
        self.u_switch, self.u_accept = f(B, n), f(B)
        self.new_idx = torch.zeros(B, n, dtype=torch.long, device=self.dev)
        self._u_sw, self._u_ac = f(64, B, n), f(64, B)
        self._n_ix = torch.zeros(64, B, n, dtype=torch.long, device=self.dev)
        self._draw_pos = 0
        self._cur = (self.u_switch, self.new_idx, self.u_accept)
        self.n_iter = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.x_sum = f(B, n)
        self.fk_ws, self.fk_nb = hand.fk_ws(B, self.dev)
        self.fc_nb = ops._size_call("gq_fc_workspace_bytes", ctypes.c_int64(B), n, int(self.fc["n_cone_vecs"]),
                                    int(self.fc["max_iter"]))
        self.fc_ws = ops._ws(self.fc_nb, self.dev).zero_()  # zero once: block counter of the large-batch stop rule
        self.pen_nb, self.pen_ws = 0, None
        if int(penetration_only) == 3:  # queue path of the penetration query (A/B tests); counters start at zero
            self.pen_nb = ops._size_call("gq_hand_pen_workspace_bytes", ctypes.c_int64(B), ctypes.c_int64(P), self.L)
            self.pen_ws = torch.zeros(self.pen_nb, dtype=torch.uint8, device=self.dev)
        self._graph, self._graph_iters, self._graph_pending = None, 1, 0
        self._after_reset = False
        self._reinit = None  # 0-dim device flag of the last step_reset: did its mask select any row (fit.py:412)
        self.kernel_events = None
        self._span = torch.zeros(64, 2, dtype=torch.int64, device=self.dev)
        self._span[:, 0] = -1  # {~0, 0}: armed
        self._span_acc = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self._side = None
        self.penetration_only = int(penetration_only)  # E_pen only needs dis > 0 (energy.py:59-61)
        self._can_fuse = self.penetration_only == 1  # every energy type has a fused form (gq_fc_pen_step / gq_alt_pen_step)
        # bounding spheres of the 256-point slices of the surface points: block-level link pre-cull of the penetration query
        self.patch = torch.empty(self.n_obj, (self.P + 255) // 256, 4, device=self.dev)
        _C.call("gq_surface_patches", _C.f32(self.surf), ctypes.c_int64(self.n_obj), ctypes.c_int64(self.P), _C.f32(self.patch),
                _C.stream_ptr())
        # link-driven penetration query through a coarse grid over the objects' surface points (0 = the point-driven one)
        self.grid = None
        if self.penetration_only == 1 and point_grid and self.P <= 4096:
            self.grid = ops.PointGrid(self.surf, int(point_grid))
        self.tdg_dirs = None
        if energy_type == "tdg":
            if tdg_directions is None:
                from .metrics.ops.tdg import random_sample_points_on_sphere

                tdg_directions = random_sample_points_on_sphere(3, 1000)
            self.tdg_dirs = torch.as_tensor(tdg_directions, dtype=torch.float32).to(self.dev).contiguous()
        e = _C.RowEnergyDesc()
        e.dist_sq, e.sign, e.obj_dir, e.hand_normals = (t.data_ptr() for t in (self.d2, self.sgn, self.onrm, self.cnrm))
        e.joints_lower, e.joints_upper = self.jlo.data_ptr(), self.jhi.data_ptr()
        e.e_fc, e.e_pen, e.e_spen = (self.terms_new[i].data_ptr() for i in (1, 2, 3))
        e.n = n
        e.w_dis, e.w_fc, e.w_pen, e.w_spen, e.w_joints = (float(self.w[k]) for k in TERM_NAMES)
        e.e_dis, e.e_joints, e.total = self.terms_new[0].data_ptr(), self.terms_new[4].data_ptr(), self.total_new.data_ptr()
        self._row_energy = e
        # descriptors of the fused force-closure + penetration launches (gq_fc_pen_step)
        fc, fd = self.fc, _C.FcStepDesc()
        fd.dist_sq, fd.sign, fd.obj_dir, fd.closest = (t.data_ptr() for t in (self.d2, self.sgn, self.onrm, self.closest))
        fd.contact_pts, fd.hand_normals, fd.cog = self.cpts.data_ptr(), self.cnrm.data_ptr(), self.cog.data_ptr()
        fd.batch, fd.n_contact, fd.n_cone = B, n, int(fc["n_cone_vecs"])
        fd.friction, fd.torque_weight, fd.max_limit = float(fc["friction"]), float(fc["torque_weight"]), float(fc["max_limit"])
        fd.svd_gain, fd.values_gain, fd.eps, fd.max_iter = (float(fc["svd_gain"]), float(fc["values_gain"]), float(fc["eps"]),
                                                            int(fc["max_iter"]))
        fd.w_dis, fd.w_fc = float(self.w["E_dis"]), float(self.w["E_fc"])
        fd.obj_normal, fd.g_contact_pts, fd.g_hand_normals = (t.data_ptr() for t in (self.obj_normal, self.g_cpts, self.g_cnrm))
        fd.e_fc, fd.x_sum, fd.n_iter = self.terms_new[1].data_ptr(), self.x_sum.data_ptr(), self.n_iter.data_ptr()
        fd.workspace, fd.workspace_bytes = self.fc_ws.data_ptr(), self.fc_nb
        pd = _C.PenStepDesc()
        pd.links, pd.surface_points = self.hand.links.handle, self.surf.data_ptr()
        pd.n_obj, pd.n_surface, pd.batch_each, pd.pose_dim = self.n_obj, P, self.be, D
        pd.Rg, pd.link_T = self.Rg.data_ptr(), self.link_T.data_ptr()
        pd.dis, pd.link, pd.gvec = self.pen_dis.data_ptr(), self.pen_link.data_ptr(), self.pen_gvec.data_ptr()
        pd.link_wrench, pd.gRt, pd.w_pen, pd.e_pen = (self.wrench.data_ptr(), self.gRt.data_ptr(), float(self.w["E_pen"]),
                                                      self.terms_new[2].data_ptr())
        pd.span, pd.span_acc = self._span.data_ptr(), self._span_acc.data_ptr()
        pd.grid = self.grid.handle if self.grid is not None else None
        pd.patch_spheres = self.patch.data_ptr()
        if self.S > 0:  # sphere centres + self penetration as a third role of the second fused launch
            pd.hand, pd.w_spen = self.hand.handle, float(self.w["E_spen"])
            pd.e_spen, pd.g_sphere_centers, pd.sphere_centers = (self.terms_new[3].data_ptr(), self.g_sph_w.data_ptr(),
                                                                  self.spheres.data_ptr())
        self._fc_desc, self._pen_desc = fd, pd
        self._alt_desc = None
        if energy_type != "graspqp":  # the fused launches of the other energy types (gq_alt_pen_step)
            ad = _C.AltFcDesc()
            ad.dist_sq, ad.sign, ad.obj_dir, ad.closest = fd.dist_sq, fd.sign, fd.obj_dir, fd.closest
            ad.contact_pts, ad.hand_normals, ad.cog = fd.contact_pts, fd.hand_normals, fd.cog
            ad.batch, ad.n_contact = B, n
            ad.energy = 1 if energy_type == "dexgrasp" else 2
            ad.torque_weight = 0.0  # the reference's call site (core/energy.py:35-42)
            if energy_type == "tdg":
                ad.directions, ad.n_directions = self.tdg_dirs.data_ptr(), int(self.tdg_dirs.shape[0])
                ad.friction, ad.obb_length, ad.enable_density, ad.scale = 0.2, 0.2, 1, 100.0  # as _eval_contacts
            ad.w_dis, ad.w_fc = float(self.w["E_dis"]), float(self.w["E_fc"])
            ad.obj_normal, ad.g_contact_pts, ad.g_hand_normals = fd.obj_normal, fd.g_contact_pts, fd.g_hand_normals
            ad.e_fc = self.terms_new[1].data_ptr()
            self._alt_desc = ad
        # MalaStar.try_step / accept_step as head / tail of the FK kernels
        self._fuse_loop = True
        self._slot_ctr = torch.zeros(2, dtype=torch.int32, device=self.dev)
        m, pr = self.mala, _C.ProposeDesc()
        pr.hand_pose, pr.grad, pr.contact_idx = self.hand_pose.data_ptr(), self.grad.data_ptr(), self.contact_idx.data_ptr()
        pr.u_switch, pr.new_idx = self._u_sw.data_ptr(), self._n_ix.data_ptr()
        pr.ema, pr.step, pr.step_size_out = self.ema.data_ptr(), self.step_count.data_ptr(), self.s_out.data_ptr()
        pr.g2_scratch = self.g2.data_ptr()
        pr.energy, pr.batch_each, pr.z_out = self.energy.data_ptr(), self.be, self.z.data_ptr()
        pr.step_size, pr.stepsize_period, pr.decay = float(m["step_size"]), int(m["stepsize_period"]), float(m["temperature_decay"])
        pr.mu, pr.switch_possibility, pr.clip_grad = float(m["mu"]), float(m["switch_possibility"]), int(bool(m["clip_grad"]))
        pr.slot_ctr, pr.slots = self._slot_ctr.data_ptr(), 64
        ac = _C.AcceptDesc()
        ac.u_accept, ac.reset_mask, ac.step = self._u_ac.data_ptr(), None, self.step_count.data_ptr()
        ac.z = self.z.data_ptr() if optimizer == "mala_star" else None  # AnnealingDexGraspNet: plain annealing
        ac.starting_temperature, ac.decay, ac.annealing_period = (float(m["starting_temperature"]), float(m["temperature_decay"]),
                                                                  int(m["annealing_period"]))
        ac.energy, ac.pose, ac.idx, ac.grad = (t.data_ptr() for t in (self.energy, self.hand_pose, self.contact_idx, self.grad))
        ac.accept, ac.temperature = self.accept.data_ptr(), self.temperature.data_ptr()
        ac.n_terms, ac.terms_new, ac.terms = 5, self.terms_new.data_ptr(), self.terms.data_ptr()
        ac.slot_ctr, ac.slots = self._slot_ctr.data_ptr(), 64
        self._propose_desc, self._accept_desc = pr, ac
        sd = _C.SdfDesc()
        sd.meshes, sd.queries_per_mesh = self.objs.handle, self.be * n
        sd.dist_sq, sd.sign, sd.obj_dir, sd.closest = (t.data_ptr() for t in (self.d2, self.sgn, self.onrm, self.closest))
        self._sdf_desc = sd

    # ---- energy + gradient of the pose in (pose, idx) -> terms_new (5,B), total_new (B), grad_new (B,D) ----------
    # Four pieces: FK (+ self penetration), then two independent branches (contacts -> object SDF -> E_fc fwd+bwd |
    # hand penetration fwd+bwd), then FK backward with the row energies.  ``_evaluate`` runs the branches on two
    # streams when ``fork`` is set (inside a hipGraph capture they become parallel graph branches).
    def _eval_fk(self, pose, idx, st, loop=False, sdf=False, spheres=True):
        B, n = self.B, self.n
        sph = self.S > 0 and spheres  # the fused path computes spheres + self penetration in gq_fc_pen_step instead
        _C.call("gq_fk_forward", self.hand.handle, _C.f32(pose), _C.i64(idx), B, n, _C.f32(self.Rg), _C.f32(self.link_T),
                _C.f32(self.cpts), _C.f32(self.cnrm), _C.f32(self.spheres) if sph else None,
                float(self.w["E_spen"]), _C.f32(self.terms_new[3]) if sph else None,
                _C.f32(self.g_sph_w) if sph else None, ctypes.byref(self._propose_desc) if loop else None,
                ctypes.byref(self._sdf_desc) if sdf else None, _C.ptr(self.fk_ws), self.fk_nb, st)
        if self.S == 0:
            _C.call("gq_fill", _C.f32(self.terms_new[3]), 0.0, self.B, st)

    def _eval_contacts(self, st):
        B, n, w, fc = self.B, self.n, self.w, self.fc
        C, f32, i32 = _C.call, _C.f32, _C.i32
        e_fc = self.terms_new[1]
        C("gq_sdf_forward_meshset", self.objs.handle, f32(self.cpts), B * n, self.be * n, f32(self.d2), i32(self.sgn),
          f32(self.onrm), f32(self.closest), st)
        if self.energy_type != "graspqp":
            # E_dis terms (+ outward object normals), then the other force-closure energy adds w_fc dE/dp in one launch
            C("gq_contact_terms", f32(self.d2), i32(self.sgn), f32(self.onrm), f32(self.closest), f32(self.cpts),
              f32(self.cnrm), ctypes.c_int64(B), n, float(w["E_dis"]), f32(self.obj_normal), f32(self.g_cpts), f32(self.g_cnrm), st)
            if self.energy_type == "dexgrasp":  # torque_weight = 0 at the reference's call site (core/energy.py:35-42)
                C("gq_dexgrasp_energy", f32(self.cpts), f32(self.obj_normal), f32(self.cog), ctypes.c_int64(B), n, 0.0, None,
                  float(w["E_fc"]), 1, f32(e_fc), f32(self.g_cpts), st)
            else:
                C("gq_tdg_energy", f32(self.cpts), f32(self.obj_normal), f32(self.cog), f32(self.tdg_dirs), self.tdg_dirs.shape[0],
                  ctypes.c_int64(B), n, 0.2, 0.2, 1, 100.0, None, float(w["E_fc"]), 1, f32(e_fc), f32(self.g_cpts), st)
            return
        C("gq_fc_step", f32(self.d2), i32(self.sgn), f32(self.onrm), f32(self.closest), f32(self.cpts), f32(self.cnrm),
          f32(self.cog), B, n, int(fc["n_cone_vecs"]), float(fc["friction"]), float(fc["torque_weight"]),
          float(fc["max_limit"]), float(fc["svd_gain"]), float(fc["values_gain"]), float(fc["eps"]), int(fc["max_iter"]),
          float(w["E_dis"]), float(w["E_fc"]), f32(self.obj_normal), f32(self.g_cpts), f32(self.g_cnrm), f32(e_fc),
          f32(self.x_sum), i32(self.n_iter), _C.ptr(self.fc_ws), self.fc_nb, st)

    def _eval_spen(self, pose, st):
        _C.call("gq_spheres_self_pen", self.hand.handle, _C.f32(pose), self.D, _C.f32(self.Rg), _C.f32(self.link_T),
                ctypes.c_int64(self.B), float(self.w["E_spen"]), _C.f32(self.spheres), _C.f32(self.terms_new[3]),
                _C.f32(self.g_sph_w), st)

    def _eval_pen(self, pose, st, timer=None):
        """Hand-penetration query (the roofline kernel of bench.py) + its backward.  ``timer`` = HIP event pair around
        the query (hipExtLaunchKernelGGL start/stop events).  The query also records its own execution span in
        100 MHz s_memrealtime ticks (64 shards of {min block start, max block end}); the backward launch folds them
        into ``_span_acc`` = {sum, launches}, which works inside a hipGraph replay too."""
        if self.grid is not None:
            _C.call("gq_hand_pen_forward_cells", self.hand.links.handle, self.grid.handle, _C.f32(self.surf), self.n_obj, self.P,
                    self.be, _C.f32(pose), self.D, _C.f32(self.Rg), _C.f32(self.link_T), _C.f32(self.pen_dis),
                    _C.i32(self.pen_link), _C.f32(self.pen_gvec), timer, _C.ptr(self._span), st)
        else:
            _C.call("gq_hand_pen_forward", self.hand.links.handle, _C.f32(self.surf), self.n_obj, self.P, self.be,
                    _C.f32(pose), self.D, _C.f32(self.Rg), _C.f32(self.link_T), int(self.penetration_only),
                    _C.f32(self.pen_dis), _C.i32(self.pen_link), _C.f32(self.pen_gvec),
                    _C.ptr(self.pen_ws), self.pen_nb, timer, _C.ptr(self._span), _C.f32(self.patch), st)
        _C.call("gq_hand_pen_backward", self.L, _C.f32(self.surf), self.n_obj, self.P, self.be, _C.f32(pose), self.D,
                _C.f32(self.Rg), None, _C.i32(self.pen_link), _C.f32(self.pen_gvec), _C.f32(self.wrench), _C.f32(self.gRt),
                _C.f32(self.pen_dis), float(self.w["E_pen"]), _C.f32(self.terms_new[2]), _C.ptr(self._span),
                _C.ptr(self._span_acc), st)

    def _eval_tail(self, pose, idx, st, loop=False):
        B, n = self.B, self.n
        f32 = _C.f32
        _C.call("gq_fk_backward", self.hand.handle, f32(pose), _C.i64(idx), B, n, f32(self.Rg), f32(self.link_T),
                f32(self.g_cpts), f32(self.g_cnrm), f32(self.g_sph_w) if self.S > 0 else None, f32(self.wrench),
                f32(self.gRt), None, None, f32(self.grad_new), ctypes.byref(self._row_energy),
                ctypes.byref(self._accept_desc) if loop else None, _C.ptr(self.fk_ws), self.fk_nb, st)

    def _evaluate(self, pose, idx, st, fork=False, timer=None, fused=False, loop=False):
        """loop=True: one whole MALA* iteration -- the proposal is the head of the FK forward kernel (pose / idx are its
        outputs), the accept step the tail of the FK backward kernel."""
        # small batches: the contact queries ride along with the kinematics (latency); large ones: their own launch
        # (throughput -- query wavefronts should not hold slots while wavefront 0 of their block does the kinematics)
        attach = fused and self.B <= 512
        # per-role launches, 384..1023 rows: sphere centres + self penetration leave the FK forward launch (which both
        # branches wait for) and ride on the penetration branch, as in the fused form (same device code, same bits):
        # +3.9 % at 512 rows.  From 2048 rows on that branch is the longer one (-1.1 %), and a third graph branch for the
        # role loses at every size (-3.6 % at 512, -1 % at 1024, 0 at 2048), so there it stays in the FK forward launch.
        split_spen = not fused and self.S > 0 and 384 <= self.B < 1024 and self.split_self_pen
        self._eval_fk(pose, idx, st, loop, sdf=attach, spheres=not fused and not split_spen)
        if fused:
            if not attach:
                _C.call("gq_sdf_forward_meshset", self.objs.handle, _C.f32(self.cpts), self.B * self.n, self.be * self.n,
                        _C.f32(self.d2), _C.i32(self.sgn), _C.f32(self.onrm), _C.f32(self.closest), st)
            # both branches side by side in two launches
            self._pen_desc.hand_pose = pose.data_ptr()
            if self._alt_desc is None:
                _C.call("gq_fc_pen_step", ctypes.byref(self._fc_desc), ctypes.byref(self._pen_desc), st)
            else:
                _C.call("gq_alt_pen_step", ctypes.byref(self._alt_desc), ctypes.byref(self._pen_desc), st)
        elif not fork:
            self._eval_contacts(st)
            if split_spen:
                self._eval_spen(pose, st)
            self._eval_pen(pose, st, timer)
        else:
            main = torch.cuda.current_stream()
            sb = self._side
            sb.wait_stream(main)
            self._eval_contacts(st)
            if split_spen:
                self._eval_spen(pose, ctypes.c_void_p(sb.cuda_stream))
            self._eval_pen(pose, ctypes.c_void_p(sb.cuda_stream), timer)
            main.wait_stream(sb)
        self._eval_tail(pose, idx, st, loop)

    def evaluate(self, pose, idx):
        """Energy terms, total and d total / d pose at an arbitrary (pose, idx); returns clones."""
        self.pose_new.copy_(pose)
        self.idx_new.copy_(idx)
        self._evaluate(self.pose_new, self.idx_new, _C.stream_ptr())
        return ({k: self.terms_new[i].clone() for i, k in enumerate(TERM_NAMES)}, self.total_new.clone(),
                self.grad_new.clone())

    def reset(self, hand_pose, contact_idx):
        """fit.py:381-396: first energy evaluation; the first gradient is zeroed (optimizer.zero_grad())."""
        self.hand_pose.copy_(hand_pose)
        self.contact_idx.copy_(contact_idx)
        self._evaluate(self.hand_pose, self.contact_idx, _C.stream_ptr())
        self.energy.copy_(self.total_new)
        self.terms.copy_(self.terms_new)
        self.grad.zero_()
        self.ema.zero_()
        self.step_count.zero_()

    def draw(self, draws=None):
        """Random draws of one iteration (optimizer.py:253-257,305): full-size draws + select (no host sync),
        generated 64 iterations at a time (3 generator launches per 64 iterations).  The kernels pick the current slot
        themselves (device counter ``_slot_ctr``, mirrored by ``_draw_pos``).  ``draws`` injects one iteration."""
        k = self._draw_pos
        if draws is not None:
            self._u_sw[k].copy_(draws[0])
            self._n_ix[k].copy_(draws[1])
            self._u_ac[k].copy_(draws[2])
        elif k == 0:
            self._u_sw.uniform_(generator=self.gen)
            self._n_ix.random_(0, self.hand.spec.n_contact_candidates, generator=self.gen)
            self._u_ac.uniform_(generator=self.gen)
        self._cur = (self._u_sw[k], self._n_ix[k], self._u_ac[k])
        self._draw_pos = (k + 1) % 64

    def _propose(self, st):
        """MalaStar.try_step + z-score of the old energies; launched eagerly so that it can read this iteration's slice
        of the pre-generated random draws directly (no copy)."""
        B, D, n, m = self.B, self.D, self.n, self.mala
        C, f32, i64 = _C.call, _C.f32, _C.i64
        C("gq_mala_propose", f32(self.hand_pose), f32(self.grad), i64(self.contact_idx), f32(self._cur[0]),
          i64(self._cur[1]), B, D, n, float(m["step_size"]), int(m["stepsize_period"]), float(m["temperature_decay"]),
          float(m["mu"]), float(m["switch_possibility"]), int(bool(m["clip_grad"])), f32(self.ema), i64(self.step_count),
          f32(self.pose_new), i64(self.idx_new), f32(self.s_out), f32(self.g2), f32(self.energy), self.be, f32(self.z),
          st)  # the z-score of the OLD energies (fit.py:403-406) rides in the same launch

    def _accept(self, st):
        B, D, n, m = self.B, self.D, self.n, self.mala
        C, f32, i64 = _C.call, _C.f32, _C.i64
        C("gq_mala_accept", f32(self.total_new), f32(self._cur[2]), f32(self.z) if self.optimizer == "mala_star" else None, None,
          i64(self.step_count), f32(self.pose_new), i64(self.idx_new), f32(self.grad_new), B, D, n,
          float(m["starting_temperature"]), float(m["temperature_decay"]), int(m["annealing_period"]), f32(self.energy),
          f32(self.hand_pose), i64(self.contact_idx), f32(self.grad), _C.u8(self.accept), f32(self.temperature), 5,
          f32(self.terms_new), f32(self.terms), st)

    def start_kernel_timing(self):
        """Time the hand-penetration query from now on (bench.py): clears the in-kernel span accumulator; outside a
        hipGraph every launch additionally gets a HIP event pair."""
        self._span_acc.zero_()
        self.kernel_events = []

    def kernel_times_ms(self):
        """-> (event_ms list, span_ms mean, launches): HIP-event durations of the eagerly launched queries and the mean
        in-kernel s_memrealtime span (100 MHz) of all queries since ``start_kernel_timing``."""
        ev_ms = []
        for t in self.kernel_events or []:
            ms = ctypes.c_float(0)
            _C.call("gq_timer_elapsed_ms", t, ctypes.byref(ms))
            _C.call("gq_timer_destroy", t)
            ev_ms.append(float(ms.value))
        acc = self._span_acc.cpu()
        n = int(acc[1])
        span_ms = (float(acc[0]) / max(n, 1)) / 1e5  # ticks of 10 ns -> ms
        self.kernel_events = None
        return ev_ms, span_ms, n

    def _iteration(self, st, timer=None, fork=False, fused=False):
        """One MALA* iteration as launches on ``st`` (eager, or under hipGraph capture)."""
        if self._fuse_loop:
            self._evaluate(self.pose_new, self.idx_new, st, fork=fork, timer=timer, fused=fused, loop=True)
        else:
            self._propose(st)
            self._evaluate(self.pose_new, self.idx_new, st, fork=fork, timer=timer, fused=fused)
            self._accept(st)

    def step(self, draws=None):
        """One MALA* iteration.  ``draws`` = (u_switch, new_idx, u_accept) to inject random numbers (tests)."""
        if self._after_reset:
            return self._step_after_reset(draws)
        if self._graph_pending and self._draw_pos == 0:
            self.flush()  # queued iterations still need the draws that the refill below would overwrite
        self.draw(draws)
        st = _C.stream_ptr()
        if self._graph is not None and self._fuse_loop:
            if self._graph_iters > 1:  # the captured graph holds several iterations: replay it once per group
                self._graph_pending += 1
                if self._graph_pending == self._graph_iters:
                    self._graph.replay()
                    self._graph_pending = 0
                return
            self._graph.replay()
        elif self._graph is not None:
            self._propose(st)
            self._graph.replay()
            self._accept(st)
        else:
            timer = None
            if self.kernel_events is not None and len(self.kernel_events) < 4096:
                timer = ctypes.c_void_p(0)
                _C.call("gq_timer_create", ctypes.byref(timer))
                self.kernel_events.append(timer)
            self._iteration(st, timer=timer)

    def _step_after_reset(self, draws=None):
        """The iteration that follows a re-initialisation.  Reference quirk: HandModel.set_parameters(env_mask=...) makes
        hand_pose a leaf tensor (hand_model.py:846-851), and in the next iteration autograd accumulates the new gradient
        IN PLACE into that leaf's .grad -- the tensor MalaStar keeps as old_grad_hand_pose (optimizer.py:266) -- so the
        rows rejected in that iteration get old + new gradient back (optimizer.py:333-338).  Launched eagerly with the
        stand-alone propose / accept kernels; pinned by tests/golden/mala_ext_*.npz."""
        self.flush()
        self.draw(draws)
        st = _C.stream_ptr()
        self._propose(st)
        self._evaluate(self.pose_new, self.idx_new, st)
        self._accept(st)
        # ``_reinit`` (0-dim bool, device): whether the reset iteration re-initialised any row at all -- after an empty mask
        # no leaf pose was created (fit.py:412) and this is an ordinary iteration
        rej = ((self.accept == 0) & self._reinit).unsqueeze(1)
        self.grad.add_(torch.where(rej, self.grad_new, torch.zeros_like(self.grad_new)))
        self._slot_ctr += 1
        self._after_reset = False

    def step_reset(self, reset_mask, new_pose, new_idx, draws=None, z_threshold=None):
        """One iteration of fit.py:399-458 in which the rows of ``reset_mask`` are re-initialised (fit.py:408-422): after
        the proposal their pose / contact indices are replaced by ``new_pose`` / ``new_idx`` (what the reference's
        initialize_convex_hull writes, full batch size; see ``graspqp_amd.core.initializations``), MalaStar.reset_envs zeroes their step
        counter, gradient EMA and old gradient and makes the new pose the "old" one, and the accept step accepts them
        unconditionally.  ``reset_mask=None`` takes the reference's rule z_score > ``z_threshold`` (fit.py:409) from the
        z-scores the proposal kernel has just computed -- on the device, without a host round trip.  Launched eagerly with
        the stand-alone propose / accept kernels (this happens every few hundred iterations); the device-side draw-slot
        counter is advanced by hand so that graph replays stay in step.  A mask that selects no row (possible when
        batch_size_each is small: the largest z-score of b rows is (b-1)/sqrt(b)) makes this an ordinary iteration, as in
        the reference (fit.py:412); pinned by steps R_s4 / R_s5 of tests/golden/mala_ext_*.npz."""
        self.flush()
        self.draw(draws)
        st = _C.stream_ptr()
        self._propose(st)
        if reset_mask is None:
            m = self.z > float(z_threshold)  # NaN z (one-row objects) never resets, like the reference's comparison
        else:
            m = reset_mask.to(self.dev, torch.bool)
        self.reset_mask = m
        mc = m.unsqueeze(1)
        # fit.py:412 ``if reset_mask.sum() > 0``: with an EMPTY mask nothing is re-initialised and the iteration is an ordinary
        # one (the masked merges below are no-ops then; the evaluation must use the proposal's indices and the next
        # iteration must not see a leaf pose).  Kept on the device as a 0-dim flag: no host round trip.
        self._reinit = m.any()
        idx_all = torch.where(self._reinit, new_idx.to(self.dev), self.idx_new).contiguous()
        # masked merges with torch.where: boolean-mask indexing would synchronise with the host
        self.pose_new.copy_(torch.where(mc, new_pose.to(self.dev, torch.float32), self.pose_new))
        self.idx_new.copy_(torch.where(mc, idx_all, self.idx_new))
        mala = self.optimizer == "mala_star"
        if mala:  # MalaStar.reset_envs (optimizer.py:275-284); AnnealingDexGraspNet.reset_envs is a no-op (:148-149)
            self.step_count.masked_fill_(m, 0)
            self.ema.masked_fill_(mc, 0.0)
            self.hand_pose.copy_(torch.where(mc, self.pose_new, self.hand_pose))
            self.contact_idx.copy_(torch.where(mc, self.idx_new, self.contact_idx))
            self.grad.masked_fill_(mc, 0.0)
        # reference quirk (hand_model.py:815-831): set_parameters(..., env_mask) gathers the contact points of ALL rows with
        # the freshly drawn indices it is handed (initializations.py:190-193), while the rows outside the mask keep
        # their proposal's indices as state -- so this iteration's energies are evaluated at ``new_idx`` everywhere
        # (``idx_all`` is the proposal's ``idx_new`` when the mask is empty, see above)
        self._evaluate(self.pose_new, idx_all, st)
        rm = m.to(torch.uint8).contiguous() if mala else None  # AnnealingDexGraspNet.accept_step ignores reset_mask
        B, D, n, mc = self.B, self.D, self.n, self.mala
        _C.call("gq_mala_accept", _C.f32(self.total_new), _C.f32(self._cur[2]), _C.f32(self.z) if mala else None, _C.u8(rm),
                _C.i64(self.step_count), _C.f32(self.pose_new), _C.i64(self.idx_new), _C.f32(self.grad_new), B, D, n,
                float(mc["starting_temperature"]), float(mc["temperature_decay"]), int(mc["annealing_period"]),
                _C.f32(self.energy), _C.f32(self.hand_pose), _C.i64(self.contact_idx), _C.f32(self.grad),
                _C.u8(self.accept), _C.f32(self.temperature), 5, _C.f32(self.terms_new), _C.f32(self.terms), st)
        self._slot_ctr += 1
        self._after_reset = True

    def realign_draws(self):
        """Run queued iterations, then restart the 64-slot buffer of random draws at slot 0 (the next ``step`` refills
        it): afterwards a multi-iteration graph replays whole groups until the buffer wraps, so a timed region that
        starts here contains no eager remainder iterations (bench.py)."""
        self.flush()
        self._slot_ctr.zero_()
        self._draw_pos = 0

    # ---- on-device (re-)initialisation: initialize_convex_hull, scripts/fit.py:315,408-422 ------------------------------
    def set_hulls(self, hulls, init_args=None):
        """``hulls`` = ObjectModel.convex_hulls() (device triangles, area table, offsets); ``init_args`` = the ranges of
        scripts/fit.py:59-71 (Namespace / dict; reference defaults otherwise)."""
        self._hulls, self._init_args = hulls, init_args

    def fresh_state(self):
        """hand_pose (B,D) and contact indices (B,n) of initialize_convex_hull for ALL rows, drawn with the stepper's
        generator -- device tensors, nothing touches the host."""
        from .core.initializations import convex_hull_poses

        if getattr(self, "_hulls", None) is None:
            raise RuntimeError("GraspStepper.set_hulls(object_model.convex_hulls()) must be called first")
        pose = convex_hull_poses(self.hand.spec, self._hulls, self.n_obj, self.be, self._init_args, self.gen, self.dev)
        idx = torch.randint(self.hand.spec.n_contact_candidates, (self.B, self.n), device=self.dev, generator=self.gen)
        return pose, idx

    def initialize(self):
        """fit.py:315 + 381-396: initialize_convex_hull for every row, then the first energy evaluation."""
        pose, idx = self.fresh_state()
        self.reset(pose, idx)

    def run(self, n_iter, reset_epochs=600, z_score_threshold=1.0, callback=None):
        """The reference's schedule (fit.py:399-458): ``n_iter`` MALA* iterations; every ``reset_epochs`` iterations
        (while step < n_iter - 2 * reset_epochs) the rows whose per-object z-score exceeds ``z_score_threshold`` are
        re-initialised on the convex hull.  Ordinary iterations replay from the captured hipGraph, reset iterations are
        launched eagerly; the host never waits for the device.  ``callback(step)`` (e.g. export_poses every 500
        iterations, fit.py:518-521) is called after the iteration ``step``."""
        for step in range(1, n_iter + 1):
            if reset_epochs is not None and step % reset_epochs == 0 and step < n_iter - 2 * reset_epochs:
                pose, idx = self.fresh_state()
                self.step_reset(None, pose, idx, z_threshold=z_score_threshold)
            else:
                self.step()
            if callback is not None:
                self.flush()
                callback(step)
        self.flush()

    def flush(self):
        """Run the iterations that ``step`` has queued for a multi-iteration graph but not yet replayed."""
        if self._graph_pending:
            st = _C.stream_ptr()
            for _ in range(self._graph_pending):
                self._iteration(st, fused=self._can_fuse)
            self._graph_pending = 0

    def capture(self, fork=None, fused=None, iters=1):
        """Capture one iteration into a hipGraph: FK forward (column means of the squared gradient, the proposal, the
        kinematics and the object SDF of the contacts in one launch), the two stage launches that hold the force-closure
        and the penetration branch side by side (``fused``), FK backward (with the energies and the accept step as its
        tail) -- four launches, no host involvement.  ``fork`` = every role its own launch, the two branches as parallel graph
        branches: per-role occupancy instead of one register budget for both roles.  Left at None the mode follows the
        batch: one grid below 384 rows (latency: 3.14 vs 2.61 M evals/s at 256 rows, 3.07 vs 3.00 at 320), graph branches
        from 384 rows on (throughput: 3.67 vs 3.52 M at 384, 4.5 vs 4.1 M at 512, 8.0 vs 6.5 M at 2048, 8.7 vs 6.9 M at
        4096; tools/ab_fork.sh).  ``iters``
        > 1 captures that many consecutive iterations in one graph (every kernel finds its random draws through the
        device-side slot counter), which removes the graph-launch gap between iterations; ``step`` then replays once
        per ``iters`` calls and ``flush`` runs a remainder.  The state is saved and restored around the warm-up +
        capture passes, so capturing does not advance the chain."""
        keep = (self.hand_pose, self.contact_idx, self.grad, self.energy, self.ema, self.step_count, self.terms,
                self._span_acc, self._slot_ctr)
        saved = [t.clone() for t in keep]
        rng = (self.gen.get_state(), self._draw_pos)
        if fused is None:
            fused = self.B < 384 if fork is None else not fork
        if fork is None:
            fork = not fused
        fused = fused and self._can_fuse
        fork = fork and not fused
        self.graph_mode = "one grid" if fused else "graph branches" if fork else "serial"
        if fork and self._side is None:
            self._side = torch.cuda.Stream()
        s = torch.cuda.Stream()
        self.draw()  # may refill the draw buffers on the current stream: order it before the side stream's warm-up
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._iteration(_C.stream_ptr(), fork=fork, fused=fused)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        iters = iters if self._fuse_loop else 1
        assert 64 % iters == 0, "iters must divide the 64-slot draw buffer"
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            if self._fuse_loop:
                for _ in range(iters):
                    self._iteration(_C.stream_ptr(), fork=fork, fused=fused)
            else:
                self._evaluate(self.pose_new, self.idx_new, _C.stream_ptr(), fork=fork, fused=fused)
        torch.cuda.synchronize()
        for t, v in zip(keep, saved):
            t.copy_(v)
        self.gen.set_state(rng[0])
        self._draw_pos = rng[1]
        self._graph, self._graph_iters, self._graph_pending = g, iters, 0
        return g
