"""GraspStepper -- one MALA* iteration (reference scripts/fit.py:399-458) as a fixed sequence of HIP launches.

No autograd, no host synchronisation, no allocation inside ``step``: all buffers are created once, so the
whole iteration is capturable into a hipGraph (``capture()``).  It runs exactly the kernels that the autograd
route (``graspqp_amd.core``) runs, in the reference's order:

    propose -> FK+contacts -> object SDF -> contact terms (E_dis grads) -> E_fc fwd -> hand penetration ->
    self penetration -> row energies (+total) -> E_fc bwd -> penetration bwd -> FK bwd -> z-score -> accept

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
                 penetration_only: bool = True):
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
        self.fc_ws = ops._ws(self.fc_nb, self.dev)
        self.pen_nb = ops._size_call("gq_hand_pen_workspace_bytes", ctypes.c_int64(B), ctypes.c_int64(P), self.L)
        self.pen_ws = torch.zeros(self.pen_nb, dtype=torch.uint8, device=self.dev)  # queue counters start at zero
        self._graph = None
        self.kernel_events = None
        self._span = None
        self.penetration_only = int(penetration_only)  # E_pen only needs dis > 0 (energy.py:59-61)

    # ---- energy + gradient of the pose in (pose, idx) -> terms_new (5,B), total_new (B), grad_new (B,D) ----------
    def _eval_pre(self, pose, idx, st):
        B, n, P, w, fc = self.B, self.n, self.P, self.w, self.fc
        C, f32, i32, i64 = _C.call, _C.f32, _C.i32, _C.i64
        e_fc = self.terms_new[1]
        C("gq_fk_forward", self.hand.handle, f32(pose), i64(idx), B, n, f32(self.Rg), f32(self.link_T), f32(self.cpts),
          f32(self.cnrm), f32(self.spheres) if self.S > 0 else None, _C.ptr(self.fk_ws), self.fk_nb, st)
        C("gq_sdf_forward_meshset", self.objs.handle, f32(self.cpts), B * n, self.be * n, f32(self.d2), i32(self.sgn),
          f32(self.onrm), f32(self.closest), st)
        C("gq_contact_terms", f32(self.d2), i32(self.sgn), f32(self.onrm), f32(self.closest), f32(self.cpts),
          f32(self.cnrm), B, n, float(w["E_dis"]), f32(self.obj_normal), f32(self.g_cpts), f32(self.g_cnrm), st)
        C("gq_fc_forward", f32(self.cpts), f32(self.obj_normal), f32(self.cog), B, n, int(fc["n_cone_vecs"]),
          float(fc["friction"]), float(fc["torque_weight"]), float(fc["max_limit"]), float(fc["svd_gain"]),
          float(fc["values_gain"]), float(fc["eps"]), int(fc["max_iter"]), f32(e_fc), f32(self.x_sum), i32(self.n_iter),
          _C.ptr(self.fc_ws), self.fc_nb, st)

    def _eval_pen(self, pose, st, timer=None, span=None):
        """The dominant kernel, kept as a launch of its own so bench.py can time it live: ``timer`` = HIP event pair
        (hipExtLaunchKernelGGL start/stop events), ``span`` = device pointer to {min block start, max block end} in
        100 MHz s_memrealtime ticks (the kernel's own execution span, what rocprofv3 reports)."""
        _C.call("gq_hand_pen_forward", self.hand.links.handle, _C.f32(self.surf), self.n_obj, self.P, self.be,
                _C.f32(pose), self.D, _C.f32(self.Rg), _C.f32(self.link_T), int(self.penetration_only),
                _C.f32(self.pen_dis), _C.i32(self.pen_link), _C.f32(self.pen_gvec),
                _C.ptr(self.pen_ws) if self.penetration_only == 1 else None, self.pen_nb, timer, span, st)

    def _eval_post(self, pose, idx, st):
        B, n, P, w, fc = self.B, self.n, self.P, self.w, self.fc
        C, f32, i32, i64 = _C.call, _C.f32, _C.i32, _C.i64
        e_dis, e_fc, e_pen, e_spen, e_joints = (self.terms_new[i] for i in range(5))
        if self.S > 0:
            C("gq_self_pen_forward", self.hand.handle, f32(self.spheres), B, float(w["E_spen"]), f32(e_spen),
              f32(self.g_sph_w), st)
        else:
            C("gq_fill", f32(e_spen), 0.0, B, st)
        C("gq_row_energy", f32(self.d2), i32(self.sgn), f32(self.onrm), f32(self.cnrm), f32(pose), f32(self.jlo),
          f32(self.jhi), f32(e_fc), f32(self.pen_dis), f32(e_spen), B, n, self.J, P, float(w["E_dis"]), float(w["E_fc"]),
          float(w["E_pen"]), float(w["E_spen"]), float(w["E_joints"]), f32(e_dis), f32(e_joints), f32(e_pen),
          f32(self.total_new), f32(self.g_theta), f32(self.g_pen), st)
        # backward
        C("gq_fc_backward", f32(self.cpts), f32(self.obj_normal), f32(self.cog), f32(self.w_fc_vec), B, n,
          int(fc["n_cone_vecs"]), float(fc["friction"]), float(fc["torque_weight"]), float(fc["svd_gain"]),
          float(fc["values_gain"]), 1, f32(self.g_cpts), _C.ptr(self.fc_ws), self.fc_nb, st)
        C("gq_hand_pen_backward", self.L, f32(self.surf), self.n_obj, P, self.be, f32(pose), self.D, f32(self.Rg),
          f32(self.g_pen), i32(self.pen_link), f32(self.pen_gvec), f32(self.wrench), f32(self.gRt), st)
        C("gq_fk_backward", self.hand.handle, f32(pose), i64(idx), B, n, f32(self.Rg), f32(self.link_T), f32(self.g_cpts),
          f32(self.g_cnrm), f32(self.g_sph_w) if self.S > 0 else None, f32(self.wrench), f32(self.gRt), f32(self.g_theta),
          None, f32(self.grad_new), _C.ptr(self.fk_ws), self.fk_nb, st)

    def _evaluate(self, pose, idx, st):
        self._eval_pre(pose, idx, st)
        self._eval_pen(pose, st)
        self._eval_post(pose, idx, st)

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

    def draw(self):
        """Random draws of one iteration (optimizer.py:253-257,305): full-size draws + select (no host sync),
        generated 64 iterations at a time and sliced per iteration (3 generator launches per 64 iterations)."""
        k = self._draw_pos
        if k == 0:
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
        C("gq_mala_accept", f32(self.total_new), f32(self._cur[2]), f32(self.z), None,
          i64(self.step_count), f32(self.pose_new), i64(self.idx_new), f32(self.grad_new), B, D, n,
          float(m["starting_temperature"]), float(m["temperature_decay"]), int(m["annealing_period"]), f32(self.energy),
          f32(self.hand_pose), i64(self.contact_idx), f32(self.grad), _C.u8(self.accept), f32(self.temperature), 5,
          f32(self.terms_new), f32(self.terms), st)

    def _step_head(self, st):
        self._eval_pre(self.pose_new, self.idx_new, st)

    def _step_tail(self, st):
        self._eval_post(self.pose_new, self.idx_new, st)

    def _pen_timed(self, st):
        ev = self.kernel_events
        if ev is None:
            self._eval_pen(self.pose_new, st)
            return
        i = len(ev)
        if i >= self._span.shape[0]:
            self._eval_pen(self.pose_new, st)
            return
        t = ctypes.c_void_p(0)
        _C.call("gq_timer_create", ctypes.byref(t))
        self._eval_pen(self.pose_new, st, t, ctypes.c_void_p(self._span[i].data_ptr()))
        ev.append(t)

    def start_kernel_timing(self, max_launches=4096):
        """Time every hand-penetration launch from now on (bench.py)."""
        self._span = torch.zeros(max_launches, 64, 2, dtype=torch.int64, device=self.dev)
        self._span[:, :, 0] = -1  # 0xffff... as unsigned: atomicMin target (64 shards per launch)
        self.kernel_events = []

    def kernel_times_ms(self):
        """-> (event_ms, span_ms): per-launch durations from the HIP event pairs and from the in-kernel
        s_memrealtime span (100 MHz) of the launches timed since ``start_kernel_timing``."""
        ev_ms = []
        for t in self.kernel_events or []:
            ms = ctypes.c_float(0)
            _C.call("gq_timer_elapsed_ms", t, ctypes.byref(ms))
            _C.call("gq_timer_destroy", t)
            ev_ms.append(float(ms.value))
        n = len(ev_ms)
        sp = self._span[:n].cpu()
        used = sp[:, :, 1] > 0  # shards that saw a block
        start = torch.where(used, sp[:, :, 0], torch.full_like(sp[:, :, 0], 2**62)).min(dim=1).values
        end = sp[:, :, 1].max(dim=1).values
        span_ms = ((end - start).double() / 1e5).tolist()  # ticks of 10 ns -> ms
        self.kernel_events = None
        return ev_ms, span_ms

    def step(self, draws=None):
        """One MALA* iteration.  ``draws`` = (u_switch, new_idx, u_accept) to inject random numbers (tests)."""
        if draws is None:
            self.draw()
        else:
            self.u_switch.copy_(draws[0])
            self.new_idx.copy_(draws[1])
            self.u_accept.copy_(draws[2])
            self._cur = (self.u_switch, self.new_idx, self.u_accept)
        st = _C.stream_ptr()
        self._propose(st)
        if self._graph is not None:
            self._graph[0].replay()
            self._pen_timed(st)
            self._graph[1].replay()
        else:
            self._step_head(st)
            self._pen_timed(st)
            self._step_tail(st)
        self._accept(st)

    def capture(self):
        """Capture the energy + gradient evaluation of one iteration into two hipGraphs (the launches before / after the
        hand-penetration query).  Propose, the penetration query and accept stay ordinary launches: the first and last
        read the current slice of the pre-generated random numbers, the middle one is timed by bench.py.  The state is
        saved and restored around the warm-up + capture passes, so capturing does not advance the chain."""
        saved = [t.clone() for t in (self.hand_pose, self.contact_idx, self.grad, self.energy, self.ema,
                                     self.step_count, self.terms)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        self.draw()
        with torch.cuda.stream(s):
            st = _C.stream_ptr()
            self._propose(st)
            self._step_head(st)
            self._eval_pen(self.pose_new, st)
            self._step_tail(st)
            self._accept(st)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g0, g1 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0):
            self._step_head(_C.stream_ptr())
        self._eval_pen(self.pose_new, _C.stream_ptr())
        with torch.cuda.graph(g1):
            self._step_tail(_C.stream_ptr())
        torch.cuda.synchronize()
        for t, v in zip((self.hand_pose, self.contact_idx, self.grad, self.energy, self.ema, self.step_count,
                         self.terms), saved):
            t.copy_(v)
        self._graph = (g0, g1)
        return self._graph
