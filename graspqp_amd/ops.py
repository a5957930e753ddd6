"""Torch-facing wrappers of the HIP kernels (thin: allocate outputs, pass pointers, register autograd).

Every function here runs on the current HIP stream through the C ABI (``graspqp_amd._C``); none has a CPU path.
The autograd ``Function``s reproduce exactly the differentiability contract of the packages they replace:
TorchSDF (only ``dist_sq`` w.r.t. ``points``), qpth (implicit KKT backward), pytorch_kinematics (full FK).
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _C

# ----------------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------------


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def _size_call(name, *args) -> int:
    out = ctypes.c_size_t(0)
    _C.call(name, *args, ctypes.byref(out))
    return int(out.value)


def _c(t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


# ----------------------------------------------------------------------------------------------------------
# mesh sets (device-resident triangle soups)
# ----------------------------------------------------------------------------------------------------------
class MeshSet:
    """n_mesh triangle soups on the device (face records precomputed once)."""

    def __init__(self, face_verts_list):
        fvs = [np.ascontiguousarray(np.asarray(f, dtype=np.float32).reshape(-1, 3, 3)) for f in face_verts_list]
        self.n_mesh = len(fvs)
        self.offsets = np.zeros(self.n_mesh + 1, dtype=np.int32)
        self.offsets[1:] = np.cumsum([len(f) for f in fvs])
        allf = np.ascontiguousarray(np.concatenate(fvs, 0))
        self.n_faces = int(self.offsets[-1])
        h = ctypes.c_void_p(0)
        torch.cuda.current_device()  # make sure the HIP context exists
        _C.call(
            "gq_meshset_create",
            allf.ctypes.data_as(ctypes.c_void_p),
            self.offsets.ctypes.data_as(ctypes.c_void_p),
            self.n_mesh,
            ctypes.byref(h),
        )
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _C.lib().gq_meshset_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------
# TorchSDF-compatible ops
# ----------------------------------------------------------------------------------------------------------
def index_vertices_by_faces(verts: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """torchsdf.index_vertices_by_faces: verts (V,3), faces (F,3) int64 -> (F,3,3)."""
    return verts[faces.long()]


class _ComputeSDF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, face_verts):
        pts = _c(points.detach())
        fv = _c(face_verts.detach())
        if pts.dim() != 2 or pts.shape[1] != 3:
            raise ValueError(f"compute_sdf: points must be (N,3), got {tuple(pts.shape)}")
        if fv.dim() != 3 or fv.shape[1:] != (3, 3):
            raise ValueError(f"compute_sdf: face_verts must be (F,3,3), got {tuple(fv.shape)}")
        N, F = pts.shape[0], fv.shape[0]
        dev = pts.device
        d2 = torch.empty(N, device=dev)
        sgn = torch.empty(N, dtype=torch.int32, device=dev)
        nrm = torch.empty(N, 3, device=dev)
        cls = torch.empty(N, 3, device=dev)
        if N > 0:
            nb = _size_call("gq_sdf_workspace_bytes", ctypes.c_int64(F))
            ws = _ws(nb, dev)
            _C.call("gq_sdf_forward", _C.f32(pts), N, _C.f32(fv), F, _C.f32(d2), _C.i32(sgn), _C.f32(nrm), _C.f32(cls),
                    _C.ptr(ws), nb, _C.stream_ptr())
        ctx.save_for_backward(pts, cls)
        ctx.mark_non_differentiable(sgn, nrm, cls)
        return d2, sgn, nrm, cls

    @staticmethod
    def backward(ctx, g_d2, g_sgn, g_nrm, g_cls):
        pts, cls = ctx.saved_tensors
        N = pts.shape[0]
        gp = torch.empty_like(pts)
        if N > 0:
            _C.call("gq_sdf_backward", _C.f32(_c(g_d2)), _C.f32(pts), _C.f32(cls), N, _C.f32(gp), _C.stream_ptr())
        return gp, None


def compute_sdf(points: torch.Tensor, face_verts: torch.Tensor):
    """torchsdf.compute_sdf drop-in -> (dist_sq, sign int32, normal, closest)."""
    return _ComputeSDF.apply(points, face_verts)


class _SdfMeshSet(torch.autograd.Function):
    """compute_sdf of n_mesh groups of queries against a MeshSet (object_model.py:217-220 without the loop)."""

    @staticmethod
    def forward(ctx, points, meshset, queries_per_mesh):
        pts = _c(points.detach()).reshape(-1, 3)
        N = pts.shape[0]
        dev = pts.device
        d2 = torch.empty(N, device=dev)
        sgn = torch.empty(N, dtype=torch.int32, device=dev)
        nrm = torch.empty(N, 3, device=dev)
        cls = torch.empty(N, 3, device=dev)
        _C.call("gq_sdf_forward_meshset", meshset.handle, _C.f32(pts), N, int(queries_per_mesh), _C.f32(d2), _C.i32(sgn),
                _C.f32(nrm), _C.f32(cls), _C.stream_ptr())
        ctx.save_for_backward(pts, cls)
        ctx.mark_non_differentiable(sgn, nrm, cls)
        return d2, sgn, nrm, cls

    @staticmethod
    def backward(ctx, g_d2, g_sgn, g_nrm, g_cls):
        pts, cls = ctx.saved_tensors
        gp = torch.empty_like(pts)
        _C.call("gq_sdf_backward", _C.f32(_c(g_d2)), _C.f32(pts), _C.f32(cls), pts.shape[0], _C.f32(gp), _C.stream_ptr())
        return gp, None, None


def sdf_meshset(points, meshset: MeshSet, queries_per_mesh: int):
    return _SdfMeshSet.apply(points, meshset, queries_per_mesh)


# ----------------------------------------------------------------------------------------------------------
# box QP (qpth.qp.QPFunction on G = [I; -I]) and the least-squares form used by SQPLsqSolver
# ----------------------------------------------------------------------------------------------------------
def _qp_ws(B, nz, max_iter, dev):
    nb = _size_call("gq_boxqp_workspace_bytes", ctypes.c_int64(B), int(nz), int(max_iter))
    return _ws(nb, dev), nb


class _BoxQP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, p, lower, upper, eps, max_iter, not_improved_lim):
        Qc, pc, lc, uc = _c(Q.detach()), _c(p.detach()), _c(lower.detach()), _c(upper.detach())
        B, nz = pc.shape
        dev = Qc.device
        x = torch.empty(B, nz, device=dev)
        lam = torch.empty(B, 2 * nz, device=dev)
        slack = torch.empty(B, 2 * nz, device=dev)
        nit = torch.zeros(1, dtype=torch.int32, device=dev)
        ws, nb = _qp_ws(B, nz, max_iter, dev)
        _C.call("gq_boxqp_forward", _C.f32(Qc), _C.f32(pc), _C.f32(lc), _C.f32(uc), 0.0, 0.0, B, nz, float(eps),
                int(max_iter), int(not_improved_lim), _C.f32(x), _C.f32(lam), _C.f32(slack), None, _C.i32(nit),
                _C.ptr(ws), nb, _C.stream_ptr())
        ctx.save_for_backward(Qc, x, lam, slack)
        ctx.n_iter = nit
        ctx.mark_non_differentiable(lam, slack)
        return x, lam, slack

    @staticmethod
    def backward(ctx, gx, g_lam, g_slack):
        Qc, x, lam, slack = ctx.saved_tensors
        B, nz = x.shape
        dx = torch.empty_like(x)
        dlam = torch.empty_like(lam)
        _C.call("gq_boxqp_backward", _C.f32(Qc), _C.f32(lam), _C.f32(slack), _C.f32(_c(gx)), B, nz, _C.f32(dx),
                _C.f32(dlam), _C.stream_ptr())
        gQ = 0.5 * (dx.unsqueeze(2) * x.unsqueeze(1) + x.unsqueeze(2) * dx.unsqueeze(1))
        # h = [upper; -lower]; grad_h = -dlam
        g_upper = -dlam[:, :nz]
        g_lower = dlam[:, nz:]
        return gQ, dx, g_lower, g_upper, None, None, None


def box_qp(Q, p, lower, upper, eps=5e-2, max_iter=12, not_improved_lim=3):
    """argmin 1/2 x'Qx + p'x, lower <= x <= upper -> (x, lam, slack); differentiable (qpth semantics)."""
    return _BoxQP.apply(Q, p, lower, upper, eps, max_iter, not_improved_lim)


class _LsqBoxQP(torch.autograd.Function):
    """x = argmin 1/2 x'(A'A + ridge I)x - (A'b)'x in the box; backward to A and b (qp_solver.py:101-126)."""

    @staticmethod
    def forward(ctx, A, b, lower_s, upper_s, ridge, eps, max_iter):
        Ac = _c(A.detach())
        bc = None if b is None else _c(b.detach())
        B, m, nz = Ac.shape
        dev = Ac.device
        x = torch.empty(B, nz, device=dev)
        lam = torch.empty(B, 2 * nz, device=dev)
        slack = torch.empty(B, 2 * nz, device=dev)
        nit = torch.zeros(1, dtype=torch.int32, device=dev)
        ws, nb = _qp_ws(B, nz, max_iter, dev)
        _C.call("gq_lsq_boxqp_forward", _C.f32(Ac), _C.f32(bc), None, None, float(lower_s), float(upper_s), B, m, nz,
                float(ridge), float(eps), int(max_iter), 3, _C.f32(x), _C.f32(lam), _C.f32(slack), None, _C.i32(nit),
                _C.ptr(ws), nb, _C.stream_ptr())
        ctx.save_for_backward(Ac, bc if bc is not None else torch.zeros(B, m, device=dev), x, lam, slack)
        ctx.ridge = ridge
        ctx.has_b = b is not None
        ctx.n_iter = nit
        return x

    @staticmethod
    def backward(ctx, gx):
        Ac, bc, x, lam, slack = ctx.saved_tensors
        B, m, nz = Ac.shape
        dx = torch.empty_like(x)
        dlam = torch.empty_like(lam)
        _C.call("gq_lsq_boxqp_backward", _C.f32(Ac), _C.f32(lam), _C.f32(slack), _C.f32(_c(gx)), B, m, nz,
                float(ctx.ridge), _C.f32(dx), _C.f32(dlam), _C.stream_ptr())
        # Q = A'A + ridge I -> grad_A = A (dx x' + x dx');  p = -A'b -> grad_A += -b dx', grad_b = -A dx
        Adx = (Ac @ dx.unsqueeze(-1)).squeeze(-1)
        Ax = (Ac @ x.unsqueeze(-1)).squeeze(-1)
        gA = Adx.unsqueeze(2) * x.unsqueeze(1) + Ax.unsqueeze(2) * dx.unsqueeze(1) - bc.unsqueeze(2) * dx.unsqueeze(1)
        gb = -Adx if ctx.has_b else None
        return gA, gb, None, None, None, None, None


def lsq_box_qp(A, b, lower, upper, ridge=1e-4, eps=5e-2, max_iter=12):
    return _LsqBoxQP.apply(A, b, lower, upper, ridge, eps, max_iter)


# ----------------------------------------------------------------------------------------------------------
# fused force-closure energy
# ----------------------------------------------------------------------------------------------------------
class _FcEnergy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, contact_pts, contact_normals, cog, cfg):
        cp, cn, cg = _c(contact_pts.detach()), _c(contact_normals.detach()), _c(cog.detach())
        B, n, _ = cp.shape
        k = int(cfg["n_cone_vecs"])
        dev = cp.device
        e = torch.empty(B, device=dev)
        xs = torch.empty(B, n, device=dev)
        nit = torch.zeros(1, dtype=torch.int32, device=dev)
        nb = _size_call("gq_fc_workspace_bytes", ctypes.c_int64(B), n, k, int(cfg["max_iter"]))
        ws = _ws(nb, dev)
        _C.call("gq_fc_forward", _C.f32(cp), _C.f32(cn), _C.f32(cg), B, n, k, float(cfg["friction"]),
                float(cfg["torque_weight"]), float(cfg["max_limit"]), float(cfg["svd_gain"]), float(cfg["values_gain"]),
                float(cfg["eps"]), int(cfg["max_iter"]), _C.f32(e), _C.f32(xs), _C.i32(nit), _C.ptr(ws), nb,
                _C.stream_ptr())
        ctx.save_for_backward(cp, cn, cg, ws)
        ctx.cfg = dict(cfg)
        ctx.nb = nb
        ctx.mark_non_differentiable(xs)
        return e, xs

    @staticmethod
    def backward(ctx, ge, g_xs):
        cp, cn, cg, ws = ctx.saved_tensors
        cfg = ctx.cfg
        B, n, _ = cp.shape
        gp = torch.empty_like(cp)
        _C.call("gq_fc_backward", _C.f32(cp), _C.f32(cn), _C.f32(cg), _C.f32(_c(ge)), B, n, int(cfg["n_cone_vecs"]),
                float(cfg["friction"]), float(cfg["torque_weight"]), float(cfg["svd_gain"]), float(cfg["values_gain"]), 0,
                _C.f32(gp), _C.ptr(ws), ctx.nb, _C.stream_ptr())
        return gp, None, None, None


FC_DEFAULTS = dict(friction=0.2, n_cone_vecs=4, torque_weight=5.0, max_limit=20.0, svd_gain=0.1, values_gain=2.0,
                   eps=5e-2, max_iter=12)


def fc_energy(contact_pts, contact_normals, cog, **cfg):
    """E_fc (B,) and per-contact force sums (B,n); gradient flows to contact_pts only (normals are SDF constants)."""
    c = dict(FC_DEFAULTS)
    c.update(cfg)
    return _FcEnergy.apply(contact_pts, contact_normals, cog, c)


def fc_peek(ws, B, n, k):
    """(F, x, val, svd) views of the last fc_energy forward on workspace ``ws`` (tests)."""
    outs = [ctypes.c_void_p(0) for _ in range(4)]
    _C.call("gq_fc_peek", _C.ptr(ws), ws.numel(), B, n, k, *[ctypes.byref(o) for o in outs])
    return outs


# ----------------------------------------------------------------------------------------------------------
# hand handle + kinematics
# ----------------------------------------------------------------------------------------------------------
class HandHandle:
    """Device copy of a HandSpec's reduced kinematic tree + its link meshes."""

    def __init__(self, spec):
        self.spec = spec
        torch.cuda.current_device()
        keep = []

        def arr(a, dt):
            a = np.ascontiguousarray(np.asarray(a, dtype=dt))
            keep.append(a)
            return a.ctypes.data_as(ctypes.c_void_p)

        d = _C.HandDesc()
        d.n_dofs, d.n_links = spec.n_dofs, spec.n_links
        d.n_cand, d.n_spheres = spec.n_contact_candidates, spec.n_spheres
        d.node_parent = arr(spec.node_parent, np.int32)
        d.node_type = arr(spec.node_type, np.int32)
        d.node_pre = arr(spec.node_pre[:, :3, :], np.float32)
        d.node_axis = arr(spec.node_axis, np.float32)
        d.link_node = arr(spec.link_node, np.int32)
        d.link_offset = arr(spec.link_offset[:, :3, :], np.float32)
        d.cand_pos = arr(spec.cand_pos, np.float32)
        d.cand_nrm = arr(spec.cand_nrm, np.float32)
        d.cand_link = arr(spec.cand_link, np.int32)
        d.sphere = arr(spec.sphere, np.float32)
        d.sphere_link = arr(spec.sphere_link, np.int32)
        d.joints_lower = arr(spec.joints_lower, np.float32)
        d.joints_upper = arr(spec.joints_upper, np.float32)
        h = ctypes.c_void_p(0)
        _C.call("gq_hand_create", ctypes.byref(d), ctypes.byref(h))
        self.handle = h
        self.links = MeshSet([spec.link_faces(l) for l in range(spec.n_links)])
        _C.call("gq_meshset_build_occupancy", self.links.handle)
        self.J, self.L, self.S = spec.n_dofs, spec.n_links, spec.n_spheres

    def fk_ws(self, B, dev):
        nb = _size_call("gq_fk_workspace_bytes", self.handle, ctypes.c_int64(B))
        return _ws(nb, dev), nb

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _C.lib().gq_hand_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _FK(torch.autograd.Function):
    """hand_pose, contact idx -> (Rg (B,3,3), link_T (B,L,3,4), contact_points, contact_normals, sphere_centers)."""

    @staticmethod
    def forward(ctx, hand_pose, idx, hand):
        hp = _c(hand_pose.detach())
        ix = _c(idx, torch.int64)
        B, n = ix.shape
        dev = hp.device
        Rg = torch.empty(B, 3, 3, device=dev)
        LT = torch.empty(B, hand.L, 3, 4, device=dev)
        cp = torch.empty(B, n, 3, device=dev)
        cn = torch.empty(B, n, 3, device=dev)
        sc = torch.empty(B, max(hand.S, 1), 3, device=dev)
        ws, nb = hand.fk_ws(B, dev)
        _C.call("gq_fk_forward", hand.handle, _C.f32(hp), _C.i64(ix), B, n, _C.f32(Rg), _C.f32(LT), _C.f32(cp),
                _C.f32(cn), _C.f32(sc) if hand.S > 0 else None, 0.0, None, None, None, None, _C.ptr(ws), nb, _C.stream_ptr())
        ctx.save_for_backward(hp, ix, Rg, LT, ws)
        ctx.hand = hand
        ctx.nb = nb
        ctx.mark_non_differentiable(ws)
        return Rg, LT, cp, cn, sc[:, : hand.S], ws

    @staticmethod
    def backward(ctx, gRg, gLT, gcp, gcn, gsc, gws):
        hp, ix, Rg, LT, ws = ctx.saved_tensors
        hand = ctx.hand
        B, n = ix.shape
        if gLT is not None and bool((gLT != 0).any()):
            raise RuntimeError("gradient w.r.t. link transforms must come through hand_pen (link wrench), not link_T")
        return _fk_backward(hand, hp, ix, Rg, LT, ws, ctx.nb, gcp, gcn, gsc, None, None, None, gRg), None, None


def _fk_backward(hand, hp, ix, Rg, LT, ws, nb, gcp, gcn, gsc, wrench, gRt, gtheta, gR):
    B, n = ix.shape
    gp = torch.empty_like(hp)
    _C.call("gq_fk_backward", hand.handle, _C.f32(hp), _C.i64(ix), B, n, _C.f32(Rg), _C.f32(LT),
            _C.f32(None if gcp is None else _c(gcp)), _C.f32(None if gcn is None else _c(gcn)),
            _C.f32(None if (gsc is None or hand.S == 0) else _c(gsc)), _C.f32(None if wrench is None else _c(wrench)),
            _C.f32(None if gRt is None else _c(gRt)), _C.f32(None if gtheta is None else _c(gtheta)),
            _C.f32(None if gR is None else _c(gR)), _C.f32(gp), None, None, _C.ptr(ws), nb, _C.stream_ptr())
    return gp


def fk_contacts(hand_pose, idx, hand: HandHandle):
    """-> (Rg (B,3,3), link_T (B,L,3,4), contact_points, contact_normals, sphere_centers, fk workspace)."""
    return _FK.apply(hand_pose, idx, hand)


# ----------------------------------------------------------------------------------------------------------
# export-time kinematics (scripts/fit.py:224-300): explicit Jacobians, damped pseudo-inverse, root pose
# ----------------------------------------------------------------------------------------------------------
def link_jacobian(hand: HandHandle, link_T, fk_workspace):
    """(B,L,6,J) geometric Jacobian [J_v; J_w] of every mesh link in the hand frame (HandModel.jacobian)."""
    LT = _c(link_T.detach())
    B = LT.shape[0]
    out = torch.empty(B, hand.L, 6, hand.J, device=LT.device)
    _C.call("gq_link_jacobian", hand.handle, ctypes.c_int64(B), _C.f32(LT), _C.f32(out), _C.ptr(fk_workspace),
            fk_workspace.numel(), _C.stream_ptr())
    return out


def contact_jacobian(hand: HandHandle, contact_idx, link_T, fk_workspace):
    """(B,n,3,J) linear contact Jacobian J_v + J_w x r (hand_model.py:1176-1196), hand frame."""
    LT = _c(link_T.detach())
    ix = _c(contact_idx, torch.int64)
    B, n = ix.shape
    out = torch.empty(B, n, 3, hand.J, device=LT.device)
    _C.call("gq_contact_jacobian", hand.handle, _C.i64(ix), ctypes.c_int64(B), n, _C.f32(LT), _C.f32(out),
            _C.ptr(fk_workspace), fk_workspace.numel(), _C.stream_ptr())
    return out


def joint_velocities(jac, directions, Rg=None, damping=1e-3):
    """theta = pinv_damped(J) d (hand_model.py:46-54,1198-1218).  jac (B,m,J), directions (B,m) -- world frame when
    Rg (B,3,3) is given.  -> (theta (B,J), residual (B,m), ee_vel (B,m))."""
    Jc = _c(jac.detach())
    d = _c(directions.detach())
    B, m, J = Jc.shape
    theta = torch.empty(B, J, device=Jc.device)
    res = torch.empty(B, m, device=Jc.device)
    ee = torch.empty(B, m, device=Jc.device)
    R = None if Rg is None else _c(Rg.detach()).reshape(B, 9)
    _C.call("gq_joint_velocities", _C.f32(Jc), _C.f32(d), _C.f32(R), ctypes.c_int64(B), m, J, float(damping),
            _C.f32(theta), _C.f32(res), _C.f32(ee), _C.stream_ptr())
    return theta, res, ee


def root_pose_wxyz(hand_pose):
    """(B,7) = [translation, unit quaternion (w,x,y,z)] of hand_pose[:, :9] (fit.py:260-263)."""
    hp = _c(hand_pose.detach())
    out = torch.empty(hp.shape[0], 7, device=hp.device)
    _C.call("gq_root_pose_wxyz", _C.f32(hp), ctypes.c_int64(hp.shape[0]), hp.shape[1], _C.f32(out), _C.stream_ptr())
    return out


class _HandPen(torch.autograd.Function):
    """max-over-links signed distance (inside positive) of object surface points; differentiable w.r.t. hand_pose.

    The kinematic state (Rg, link_T, fk workspace) is passed in detached; the gradient is routed to ``hand_pose``
    directly through the analytic FK backward (link wrenches), which is what autograd through
    pytorch_kinematics computes in the reference (hand_model.py:875-987).
    """

    @staticmethod
    def forward(ctx, hand_pose, surface_points, batch_each, hand, idx, Rg, LT, ws, nb, penetration_only=False):
        hp = _c(hand_pose.detach())
        sp = _c(surface_points)
        n_obj, P, _ = sp.shape
        B = hp.shape[0]
        dev = hp.device
        dis = torch.empty(B, P, device=dev)
        link = torch.zeros(B, P, dtype=torch.int32, device=dev)  # mode 1 writes link / gvec only where dis > 0
        gvec = torch.zeros(B, P, 3, device=dev)
        pws, pnb = None, 0
        if int(penetration_only) == 3:  # queue path without candidate lists (kept for A/B tests)
            pnb = _size_call("gq_hand_pen_workspace_bytes", ctypes.c_int64(B), ctypes.c_int64(P), hand.L)
            pws = torch.zeros(pnb, dtype=torch.uint8, device=dev)  # queue counters must start at zero
        _C.call("gq_hand_pen_forward", hand.links.handle, _C.f32(sp), n_obj, P, int(batch_each), _C.f32(hp), hp.shape[1],
                _C.f32(Rg), _C.f32(LT), int(penetration_only), _C.f32(dis), _C.i32(link), _C.f32(gvec), _C.ptr(pws), pnb,
                None, None, _C.stream_ptr())
        ctx.save_for_backward(hp, sp, idx, Rg, LT, ws, link, gvec)
        ctx.hand, ctx.batch_each, ctx.nb = hand, int(batch_each), nb
        return dis

    @staticmethod
    def backward(ctx, g):
        hp, sp, idx, Rg, LT, ws, link, gvec = ctx.saved_tensors
        hand = ctx.hand
        n_obj, P, _ = sp.shape
        B = hp.shape[0]
        dev = hp.device
        wrench = torch.empty(B, hand.L, 6, device=dev)
        gRt = torch.empty(B, 12, device=dev)
        _C.call("gq_hand_pen_backward", hand.L, _C.f32(sp), n_obj, P, ctx.batch_each, _C.f32(hp), hp.shape[1], _C.f32(Rg),
                _C.f32(_c(g)), _C.i32(link), _C.f32(gvec), _C.f32(wrench), _C.f32(gRt), None, 0.0, None, None, None,
                _C.stream_ptr())
        gp = _fk_backward(hand, hp, idx, Rg, LT, ws, ctx.nb, None, None, None, wrench, gRt, None, None)
        return gp, None, None, None, None, None, None, None, None, None


def hand_pen(hand_pose, surface_points, batch_each, hand, idx, Rg, LT, ws, nb, penetration_only=False):
    return _HandPen.apply(hand_pose, surface_points, batch_each, hand, idx, Rg, LT, ws, nb, penetration_only)


class _SelfPen(torch.autograd.Function):
    @staticmethod
    def forward(ctx, centers, hand):
        c = _c(centers.detach())
        B = c.shape[0]
        e = torch.empty(B, device=c.device)
        g = torch.empty_like(c)
        _C.call("gq_self_pen_forward", hand.handle, _C.f32(c), B, 1.0, _C.f32(e), _C.f32(g), _C.stream_ptr())
        ctx.save_for_backward(g)
        return e

    @staticmethod
    def backward(ctx, ge):
        (g,) = ctx.saved_tensors
        return g * ge.view(-1, 1, 1), None


def self_pen(centers, hand: HandHandle):
    if hand.S == 0:
        return torch.zeros(centers.shape[0], device=centers.device)
    return _SelfPen.apply(centers, hand)
