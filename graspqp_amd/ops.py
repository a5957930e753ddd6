"""Torch-facing wrappers of the HIP kernels (thin: allocate outputs, pass pointers, register autograd).

Every function here runs on the current HIP stream through the C ABI (``graspqp_amd._C``); none has a CPU path.
The ops are registered with the dispatcher as ``torch.ops.graspqp_amd.*`` (``torch.library.custom_op`` for the CUDA/HIP
device only, fake kernels for tracing, ``register_autograd`` for the backward -- itself a registered op), so they are
visible to ``torch.compile`` / the profiler like any ATen op.  Their differentiability contracts are exactly those of
the packages they replace: TorchSDF (only ``dist_sq`` w.r.t. ``points``), qpth (implicit KKT backward),
pytorch_kinematics (full FK).  Opaque device objects (mesh sets, hands) cross the dispatcher as integer ids.

Eager calls do not take the round trip through the dispatcher: a registered Python op costs ~40 us of host time per call
(schema matching, re-dispatch, the autograd wrapper of torch.library), a ``fit.py``-shaped loop on the class surface makes
about a dozen such calls per iteration, forward and backward, and is host-bound.  ``_Eager.<op>`` runs the SAME forward
body, ``setup_context`` and backward that are registered -- through a plain ``torch.autograd.Function`` where the op has a
gradient -- unless a trace / ``torch.compile`` is in progress or ``GRASPQP_DISPATCH=dispatcher`` / ``use_dispatcher(True)``
asks for the registered route (tests compare the two bit for bit).
"""

import ctypes
import os
import weakref
from typing import List, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _C

_custom_op = torch.library.custom_op
_ROUTE = {"dispatcher": os.environ.get("GRASPQP_DISPATCH", "eager") == "dispatcher"}


def use_dispatcher(flag: bool) -> bool:
    """Route eager calls through the registered ``torch.ops.graspqp_amd.*`` (True) or past the dispatcher (False, the
    default; module docstring).  Returns the previous setting."""
    old, _ROUTE["dispatcher"] = _ROUTE["dispatcher"], bool(flag)
    return old


class _Eager:
    """Namespace of the eager routes, one per registered op (filled by ``_eager`` next to each registration)."""


def _eager(name, opdef, bwd=None, setup=None):
    body = opdef._init_fn  # the undecorated forward body
    if bwd is None:
        direct = body
    else:
        class _Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *args):
                out = body(*args)
                setup(ctx, args, out)
                return out

            @staticmethod
            def backward(ctx, *grads):
                return bwd(ctx, *grads)

        _Fn.__name__ = _Fn.__qualname__ = "graspqp_amd_" + name
        direct = _Fn.apply
    registered = getattr(torch.ops.graspqp_amd, name)

    def call(*args):
        if _ROUTE["dispatcher"] or torch.compiler.is_compiling():
            return registered(*args)
        return direct(*args)

    call.__name__ = name
    setattr(_Eager, name, staticmethod(call))
_HANDLES = weakref.WeakValueDictionary()  # id -> MeshSet / HandHandle (ops take the id: only tensors and scalars may
_next_id = [1]                            # cross the dispatcher)


def _register_handle(obj) -> int:
    i = _next_id[0]
    _next_id[0] += 1
    _HANDLES[i] = obj
    return i


def _handle(i: int):
    try:
        return _HANDLES[int(i)]
    except KeyError:
        raise RuntimeError(f"graspqp_amd: device object {i} no longer exists") from None

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
        self.hid = _register_handle(self)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _C.lib().gq_meshset_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Bvh:
    """Implicit 4-ary box hierarchy over one mesh (csrc/bvh.hip): the acceleration data of compute_sdf for large query
    counts."""

    def __init__(self, face_verts):
        fv = np.ascontiguousarray(np.asarray(face_verts, dtype=np.float32).reshape(-1, 3, 3))
        self.n_faces = int(fv.shape[0])
        h = ctypes.c_void_p(0)
        torch.cuda.current_device()
        _C.call("gq_bvh_create", fv.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(self.n_faces), ctypes.byref(h))
        self.handle = h
        self.hid = _register_handle(self)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _C.lib().gq_bvh_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def surface_fps(face_verts_list, n_keep: int, oversample: int = 100, generator=None, draws=None, device="cuda") -> torch.Tensor:
    """(n_obj, n_keep, 3) surface samples of every mesh, drawn on the device (reference core/object_model.py:163-178):
    ``oversample * n_keep`` area-weighted samples per mesh, farthest-point sampling from sample 0 down to ``n_keep``.
    ``draws`` = (u_face (n_obj,M), u_len (n_obj,M,2)) injects the uniforms (tests)."""
    dev = torch.device(device)
    fvs = [torch.as_tensor(f, dtype=torch.float32).reshape(-1, 3, 3).to(dev) for f in face_verts_list]
    n_obj, M = len(fvs), int(oversample) * int(n_keep)
    cdfs, off = [], [0]
    for f in fvs:
        area = 0.5 * torch.linalg.cross(f[:, 1] - f[:, 0], f[:, 2] - f[:, 0]).double().norm(dim=1)
        c = torch.cumsum(area, 0)
        cdfs.append((c / c[-1]).float())
        off.append(off[-1] + f.shape[0])
    fv, cdf = torch.cat(fvs).contiguous(), torch.cat(cdfs).contiguous()
    offs = torch.tensor(off, dtype=torch.int32, device=dev)
    if draws is None:
        draws = (torch.rand(n_obj, M, device=dev, generator=generator), torch.rand(n_obj, M, 2, device=dev, generator=generator))
    u_face, u_len = (_c(d.to(dev)) for d in draws)
    out = torch.empty(n_obj, int(n_keep), 3, device=dev)
    nb = _size_call("gq_init_workspace_bytes", ctypes.c_int64(n_obj), ctypes.c_int64(M), ctypes.c_int64(int(n_keep)))
    ws = _ws(nb, dev)
    _C.call("gq_surface_fps", _C.f32(fv), _C.f32(cdf), _C.i32(offs), ctypes.c_int64(n_obj), ctypes.c_int64(M),
            ctypes.c_int64(int(n_keep)), _C.f32(u_face), _C.f32(u_len), _C.f32(out), _C.ptr(ws), nb, _C.stream_ptr())
    return out


def morton_sort_points(points: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """(n_obj,P,3) -> the same points, every object's set ordered along a 3-D Morton curve (neighbouring indices are
    spatial neighbours: the 64 points of a wavefront of the penetration query then meet the same hand links)."""
    lo, hi = points.amin(dim=1, keepdim=True), points.amax(dim=1, keepdim=True)
    q = ((points - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).to(torch.int64)
    code = torch.zeros(points.shape[:2], dtype=torch.int64, device=points.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[..., a] >> b) & 1) << (3 * b + a)
    order = torch.argsort(code, dim=1, stable=True)
    return torch.gather(points, 1, order.unsqueeze(-1).expand(-1, -1, 3)).contiguous()


class PointGrid:
    """Coarse uniform grid over the surface points of every object (n_obj,P,3): set-up data of the link-driven
    penetration query (gq_hand_pen_forward_cells)."""

    def __init__(self, surface_points, cells_per_axis: int = 0):
        sp = np.ascontiguousarray(np.asarray(surface_points.detach().cpu() if torch.is_tensor(surface_points) else surface_points,
                                             dtype=np.float32))
        self.n_obj, self.P = int(sp.shape[0]), int(sp.shape[1])
        h = ctypes.c_void_p(0)
        torch.cuda.current_device()
        _C.call("gq_pointgrid_create", sp.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(self.n_obj), ctypes.c_int64(self.P),
                int(cells_per_axis), ctypes.byref(h))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _C.lib().gq_pointgrid_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------
# TorchSDF-compatible ops
# ----------------------------------------------------------------------------------------------------------
def index_vertices_by_faces(verts: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """torchsdf.index_vertices_by_faces: verts (V,3), faces (F,3) int64 -> (F,3,3)."""
    return verts[faces.long()]


@_custom_op("graspqp_amd::sdf_backward", mutates_args=(), device_types="cuda")
def _sdf_backward(g_d2: Tensor, points: Tensor, closest: Tensor) -> Tensor:
    gp = torch.empty_like(points)
    if points.shape[0] > 0:
        _C.call("gq_sdf_backward", _C.f32(_c(g_d2)), _C.f32(points), _C.f32(closest), points.shape[0], _C.f32(gp),
                _C.stream_ptr())
    return gp


@_sdf_backward.register_fake
def _(g_d2, points, closest):
    return torch.empty_like(points)


def _sdf_outputs(pts):
    N, dev = pts.shape[0], pts.device
    return (torch.empty(N, device=dev), torch.empty(N, dtype=torch.int32, device=dev), torch.empty(N, 3, device=dev),
            torch.empty(N, 3, device=dev))


@_custom_op("graspqp_amd::compute_sdf", mutates_args=(), device_types="cuda")
def _compute_sdf_op(points: Tensor, face_verts: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    pts, fv = _c(points), _c(face_verts)
    d2, sgn, nrm, cls = _sdf_outputs(pts)
    N, F = pts.shape[0], fv.shape[0]
    if N > 0:
        nb = _size_call("gq_sdf_workspace_bytes", ctypes.c_int64(F))
        ws = _ws(nb, pts.device)
        _C.call("gq_sdf_forward", _C.f32(pts), N, _C.f32(fv), F, _C.f32(d2), _C.i32(sgn), _C.f32(nrm), _C.f32(cls),
                _C.ptr(ws), nb, _C.stream_ptr())
    return d2, sgn, nrm, cls


@_compute_sdf_op.register_fake
def _(points, face_verts):
    return _sdf_outputs(points)


def _sdf_setup(ctx, inputs, output):
    ctx.save_for_backward(_c(inputs[0]), output[3])
    ctx.mark_non_differentiable(output[1], output[2], output[3])  # TorchSDF: only dist_sq is differentiable


def _sdf_bwd(ctx, g_d2, g_sgn, g_nrm, g_cls):
    pts, cls = ctx.saved_tensors
    return _Eager.sdf_backward(g_d2, pts, cls), None


torch.library.register_autograd("graspqp_amd::compute_sdf", _sdf_bwd, setup_context=_sdf_setup)


# Per-mesh set-up of the drop-in.  The reference hands the SAME face_verts tensor to every call (hand_model.py:351-353 keeps
# one per link, object_model.py:146-148 one per object), so the acceleration data of a mesh -- Morton-sorted face records
# + oriented 64-face cluster boxes (gq_meshset_create) -- is built on the first call with a tensor and kept while that
# tensor is alive and unmodified (weak reference + data pointer + version counter; a dead tensor drops its entry).
_MESH_CACHE = {}              # (id(face_verts), kind) -> (weakref, data_ptr, _version, MeshSet | Bvh)
_MESH_CACHE_MIN_FACES = 1024  # cluster search (one wavefront per query): meshes from this size on
_BVH_MIN_QUERIES = 32768      # box hierarchy (one query per lane): query counts from this size on
_BVH_MIN_FACES, _BVH_MAX_FACES = 32, 65536


def _cached(face_verts, kind):
    key = (id(face_verts), kind)
    ent = _MESH_CACHE.get(key)
    if ent is not None and ent[0]() is face_verts and ent[1] == face_verts.data_ptr() and ent[2] == face_verts._version:
        return ent[3]
    fv = face_verts.detach().to(torch.float32).cpu().numpy()  # one device->host copy, once per mesh and kind
    obj = MeshSet([fv]) if kind == "clusters" else Bvh(fv)
    ref = weakref.ref(face_verts, lambda _r, k=key: _MESH_CACHE.pop(k, None))
    _MESH_CACHE[key] = (ref, face_verts.data_ptr(), face_verts._version, obj)
    return obj


def _cached_meshset(face_verts):
    return _cached(face_verts, "clusters")


def compute_sdf(points: torch.Tensor, face_verts: torch.Tensor):
    """torchsdf.compute_sdf drop-in -> (dist_sq, sign int32, normal, closest).

    Three device paths behind the one signature, all exact with the same winner rule (smallest ranking distance, ties to
    the smallest face index): (i) N >= 32768 queries: one query per lane through the mesh's box hierarchy (gq_sdf_forward_bvh;
    the per-link calls of HandModel.cal_distance); (ii) fewer queries against >= 1024 faces: best-first search over
    oriented 64-face cluster boxes, one wavefront per query (the contact queries of ObjectModel.cal_distance); (iii)
    otherwise the face loop of gq_sdf_forward (no set-up).  (i) and (ii) use acceleration data built on the first call
    with a ``face_verts`` tensor and kept while that tensor is alive and unmodified."""
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"compute_sdf: points must be (N,3), got {tuple(points.shape)}")
    if face_verts.dim() != 3 or tuple(face_verts.shape[1:]) != (3, 3):
        raise ValueError(f"compute_sdf: face_verts must be (F,3,3), got {tuple(face_verts.shape)}")
    if not points.is_cuda:
        raise RuntimeError("graspqp_amd ops need CUDA (ROCm) tensors; got a CPU tensor")
    N, F = points.shape[0], face_verts.shape[0]
    if N > 0 and not face_verts.requires_grad:
        if N >= _BVH_MIN_QUERIES and _BVH_MIN_FACES <= F <= _BVH_MAX_FACES:
            return _Eager.sdf_bvh(points, _cached(face_verts, "bvh").hid)
        if F >= _MESH_CACHE_MIN_FACES:
            return _Eager.sdf_meshset(points, _cached_meshset(face_verts).hid, N)
    return _Eager.compute_sdf(points, face_verts)


@_custom_op("graspqp_amd::sdf_meshset", mutates_args=(), device_types="cuda")
def _sdf_meshset_op(points: Tensor, meshset: int, queries_per_mesh: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """compute_sdf of n_mesh groups of queries against a MeshSet (object_model.py:217-220 without the loop)."""
    pts = _c(points).reshape(-1, 3)
    d2, sgn, nrm, cls = _sdf_outputs(pts)
    _C.call("gq_sdf_forward_meshset", _handle(meshset).handle, _C.f32(pts), pts.shape[0], int(queries_per_mesh), _C.f32(d2),
            _C.i32(sgn), _C.f32(nrm), _C.f32(cls), _C.stream_ptr())
    return d2, sgn, nrm, cls


@_sdf_meshset_op.register_fake
def _(points, meshset, queries_per_mesh):
    return _sdf_outputs(points.reshape(-1, 3))


def _sdf_ms_setup(ctx, inputs, output):
    ctx.save_for_backward(_c(inputs[0]).reshape(-1, 3), output[3])
    ctx.in_shape = inputs[0].shape
    ctx.mark_non_differentiable(output[1], output[2], output[3])


def _sdf_ms_bwd(ctx, g_d2, g_sgn, g_nrm, g_cls):
    pts, cls = ctx.saved_tensors
    return _Eager.sdf_backward(g_d2, pts, cls).reshape(ctx.in_shape), None, None


torch.library.register_autograd("graspqp_amd::sdf_meshset", _sdf_ms_bwd, setup_context=_sdf_ms_setup)


@_custom_op("graspqp_amd::sdf_bvh", mutates_args=(), device_types="cuda")
def _sdf_bvh_op(points: Tensor, bvh: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """compute_sdf of many points against one mesh through its box hierarchy (one query per lane)."""
    pts = _c(points).reshape(-1, 3)
    d2, sgn, nrm, cls = _sdf_outputs(pts)
    _C.call("gq_sdf_forward_bvh", _handle(bvh).handle, _C.f32(pts), pts.shape[0], _C.f32(d2), _C.i32(sgn), _C.f32(nrm),
            _C.f32(cls), _C.stream_ptr())
    return d2, sgn, nrm, cls


@_sdf_bvh_op.register_fake
def _(points, bvh):
    return _sdf_outputs(points.reshape(-1, 3))


def _sdf_bvh_bwd(ctx, g_d2, g_sgn, g_nrm, g_cls):
    pts, cls = ctx.saved_tensors
    return _Eager.sdf_backward(g_d2, pts, cls).reshape(ctx.in_shape), None


torch.library.register_autograd("graspqp_amd::sdf_bvh", _sdf_bvh_bwd, setup_context=_sdf_ms_setup)


def sdf_meshset(points, meshset: MeshSet, queries_per_mesh: int):
    return _Eager.sdf_meshset(points, meshset.hid, int(queries_per_mesh))


# ----------------------------------------------------------------------------------------------------------
# box QP (qpth.qp.QPFunction on G = [I; -I]) and the least-squares form used by SQPLsqSolver
# ----------------------------------------------------------------------------------------------------------
def _qp_ws(B, nz, max_iter, dev):
    nb = _size_call("gq_boxqp_workspace_bytes", ctypes.c_int64(B), int(nz), int(max_iter))
    return _ws(nb, dev), nb


@_custom_op("graspqp_amd::box_qp", mutates_args=(), device_types="cuda")
def _box_qp_op(Q: Tensor, p: Tensor, lower: Tensor, upper: Tensor, eps: float, max_iter: int,
               not_improved_lim: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    Qc, pc, lc, uc = _c(Q), _c(p), _c(lower), _c(upper)
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
    return x, lam, slack, nit


@_box_qp_op.register_fake
def _(Q, p, lower, upper, eps, max_iter, not_improved_lim):
    B, nz = p.shape
    return (p.new_empty(B, nz), p.new_empty(B, 2 * nz), p.new_empty(B, 2 * nz), p.new_empty(1, dtype=torch.int32))


@_custom_op("graspqp_amd::box_qp_backward", mutates_args=(), device_types="cuda")
def _box_qp_bwd_op(Q: Tensor, lam: Tensor, slack: Tensor, gx: Tensor) -> Tuple[Tensor, Tensor]:
    B, nz = gx.shape
    dx = torch.empty(B, nz, device=gx.device)
    dlam = torch.empty(B, 2 * nz, device=gx.device)
    _C.call("gq_boxqp_backward", _C.f32(_c(Q)), _C.f32(lam), _C.f32(slack), _C.f32(_c(gx)), B, nz, _C.f32(dx), _C.f32(dlam),
            _C.stream_ptr())
    return dx, dlam


@_box_qp_bwd_op.register_fake
def _(Q, lam, slack, gx):
    return torch.empty_like(gx), torch.empty_like(lam)


def _box_qp_setup(ctx, inputs, output):
    ctx.save_for_backward(_c(inputs[0]), output[0], output[1], output[2])
    ctx.mark_non_differentiable(output[1], output[2], output[3])


def _box_qp_bwd(ctx, gx, g_lam, g_slack, g_nit):
    Qc, x, lam, slack = ctx.saved_tensors
    nz = x.shape[1]
    dx, dlam = _Eager.box_qp_backward(Qc, lam, slack, gx)
    gQ = 0.5 * (dx.unsqueeze(2) * x.unsqueeze(1) + x.unsqueeze(2) * dx.unsqueeze(1))
    # h = [upper; -lower]; grad_h = -dlam
    return gQ, dx, dlam[:, nz:], -dlam[:, :nz], None, None, None


torch.library.register_autograd("graspqp_amd::box_qp", _box_qp_bwd, setup_context=_box_qp_setup)


def box_qp(Q, p, lower, upper, eps=5e-2, max_iter=12, not_improved_lim=3):
    """argmin 1/2 x'Qx + p'x, lower <= x <= upper -> (x, lam, slack); differentiable (qpth semantics)."""
    x, lam, slack, _ = _Eager.box_qp(Q, p, lower, upper, float(eps), int(max_iter), int(not_improved_lim))
    return x, lam, slack


@_custom_op("graspqp_amd::lsq_box_qp", mutates_args=(), device_types="cuda")
def _lsq_box_qp_op(A: Tensor, b: Tensor, lower_s: float, upper_s: float, ridge: float, eps: float,
                   max_iter: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """x = argmin 1/2 x'(A'A + ridge I)x - (A'b)'x in the box (qp_solver.py:101-126) -> (x, lam, slack, n_iter)."""
    Ac, bc = _c(A), _c(b)
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
    return x, lam, slack, nit


@_lsq_box_qp_op.register_fake
def _(A, b, lower_s, upper_s, ridge, eps, max_iter):
    B, m, nz = A.shape
    return (A.new_empty(B, nz), A.new_empty(B, 2 * nz), A.new_empty(B, 2 * nz), A.new_empty(1, dtype=torch.int32))


@_custom_op("graspqp_amd::lsq_box_qp_backward", mutates_args=(), device_types="cuda")
def _lsq_box_qp_bwd_op(A: Tensor, lam: Tensor, slack: Tensor, gx: Tensor, ridge: float) -> Tuple[Tensor, Tensor]:
    B, m, nz = A.shape
    dx = torch.empty(B, nz, device=A.device)
    dlam = torch.empty(B, 2 * nz, device=A.device)
    _C.call("gq_lsq_boxqp_backward", _C.f32(_c(A)), _C.f32(lam), _C.f32(slack), _C.f32(_c(gx)), B, m, nz, float(ridge),
            _C.f32(dx), _C.f32(dlam), _C.stream_ptr())
    return dx, dlam


@_lsq_box_qp_bwd_op.register_fake
def _(A, lam, slack, gx, ridge):
    return torch.empty_like(gx), torch.empty_like(lam)


def _lsq_setup(ctx, inputs, output):
    ctx.save_for_backward(_c(inputs[0]), _c(inputs[1]), output[0], output[1], output[2])
    ctx.ridge = inputs[4]
    ctx.mark_non_differentiable(output[1], output[2], output[3])


def _lsq_bwd(ctx, gx, g_lam, g_slack, g_nit):
    Ac, bc, x, lam, slack = ctx.saved_tensors
    dx, _ = _Eager.lsq_box_qp_backward(Ac, lam, slack, gx, ctx.ridge)
    # Q = A'A + ridge I -> grad_A = A (dx x' + x dx');  p = -A'b -> grad_A += -b dx', grad_b = -A dx
    Adx = (Ac @ dx.unsqueeze(-1)).squeeze(-1)
    Ax = (Ac @ x.unsqueeze(-1)).squeeze(-1)
    gA = Adx.unsqueeze(2) * x.unsqueeze(1) + Ax.unsqueeze(2) * dx.unsqueeze(1) - bc.unsqueeze(2) * dx.unsqueeze(1)
    return gA, -Adx, None, None, None, None, None


torch.library.register_autograd("graspqp_amd::lsq_box_qp", _lsq_bwd, setup_context=_lsq_setup)


def lsq_box_qp(A, b, lower, upper, ridge=1e-4, eps=5e-2, max_iter=12, return_n_iter=False):
    """x (B,nz); with ``return_n_iter`` also the (1,) int32 iteration count of qpth's batch-global stop rule."""
    if b is None:
        b = torch.zeros(A.shape[0], A.shape[1], device=A.device, dtype=A.dtype)
    x, _, _, nit = _Eager.lsq_box_qp(A, b, float(lower), float(upper), float(ridge), float(eps), int(max_iter))
    return (x, nit) if return_n_iter else x


# ----------------------------------------------------------------------------------------------------------
# fused force-closure energy
# ----------------------------------------------------------------------------------------------------------
@_custom_op("graspqp_amd::fc_energy", mutates_args=(), device_types="cuda")
def _fc_energy_op(contact_pts: Tensor, contact_normals: Tensor, cog: Tensor, n_cone_vecs: int, friction: float,
                  torque_weight: float, max_limit: float, svd_gain: float, values_gain: float, eps: float,
                  max_iter: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (E_fc (B), force sums (B,n), n_iter (1) int32, workspace kept for the backward)."""
    cp, cn, cg = _c(contact_pts), _c(contact_normals), _c(cog)
    B, n, _ = cp.shape
    dev = cp.device
    e = torch.empty(B, device=dev)
    xs = torch.empty(B, n, device=dev)
    nit = torch.zeros(1, dtype=torch.int32, device=dev)
    nb = _size_call("gq_fc_workspace_bytes", ctypes.c_int64(B), n, int(n_cone_vecs), int(max_iter))
    ws = _ws(nb, dev)
    _C.call("gq_fc_forward", _C.f32(cp), _C.f32(cn), _C.f32(cg), B, n, int(n_cone_vecs), float(friction),
            float(torque_weight), float(max_limit), float(svd_gain), float(values_gain), float(eps), int(max_iter),
            _C.f32(e), _C.f32(xs), _C.i32(nit), _C.ptr(ws), nb, _C.stream_ptr())
    return e, xs, nit, ws


@_fc_energy_op.register_fake
def _(contact_pts, contact_normals, cog, n_cone_vecs, friction, torque_weight, max_limit, svd_gain, values_gain, eps, max_iter):
    B, n, _ = contact_pts.shape
    nb = _size_call("gq_fc_workspace_bytes", ctypes.c_int64(B), n, int(n_cone_vecs), int(max_iter))  # host-only helper
    return (contact_pts.new_empty(B), contact_pts.new_empty(B, n), contact_pts.new_empty(1, dtype=torch.int32),
            contact_pts.new_empty(nb, dtype=torch.uint8))


@_custom_op("graspqp_amd::fc_energy_backward", mutates_args=("ws",), device_types="cuda")  # scratch inside the workspace
def _fc_energy_bwd_op(contact_pts: Tensor, contact_normals: Tensor, cog: Tensor, ge: Tensor, ws: Tensor, n_cone_vecs: int,
                      friction: float, torque_weight: float, svd_gain: float, values_gain: float) -> Tensor:
    gp = torch.empty_like(contact_pts)
    B, n, _ = contact_pts.shape
    _C.call("gq_fc_backward", _C.f32(contact_pts), _C.f32(contact_normals), _C.f32(cog), _C.f32(_c(ge)), B, n,
            int(n_cone_vecs), float(friction), float(torque_weight), float(svd_gain), float(values_gain), 0, _C.f32(gp),
            _C.ptr(ws), ws.numel(), _C.stream_ptr())
    return gp


@_fc_energy_bwd_op.register_fake
def _(contact_pts, contact_normals, cog, ge, ws, n_cone_vecs, friction, torque_weight, svd_gain, values_gain):
    return torch.empty_like(contact_pts)


def _fc_setup(ctx, inputs, output):
    ctx.save_for_backward(_c(inputs[0]), _c(inputs[1]), _c(inputs[2]), output[3])
    ctx.cfg = inputs[3:]
    ctx.mark_non_differentiable(output[1], output[2], output[3])


def _fc_bwd(ctx, ge, g_xs, g_nit, g_ws):
    cp, cn, cg, ws = ctx.saved_tensors
    k, mu, tw, _ml, sg, vg, _eps, _mi = ctx.cfg
    gp = _Eager.fc_energy_backward(cp, cn, cg, ge, ws, k, mu, tw, sg, vg)
    return (gp,) + (None,) * 10


torch.library.register_autograd("graspqp_amd::fc_energy", _fc_bwd, setup_context=_fc_setup)


FC_DEFAULTS = dict(friction=0.2, n_cone_vecs=4, torque_weight=5.0, max_limit=20.0, svd_gain=0.1, values_gain=2.0,
                   eps=5e-2, max_iter=12)


def fc_energy(contact_pts, contact_normals, cog, return_n_iter=False, **cfg):
    """E_fc (B,) and per-contact force sums (B,n); gradient flows to contact_pts only (normals are SDF constants)."""
    c = dict(FC_DEFAULTS)
    c.update(cfg)
    e, xs, nit, _ = _Eager.fc_energy(
        contact_pts, contact_normals.detach(), cog.detach(), int(c["n_cone_vecs"]), float(c["friction"]),
        float(c["torque_weight"]), float(c["max_limit"]), float(c["svd_gain"]), float(c["values_gain"]), float(c["eps"]),
        int(c["max_iter"]))
    return (e, xs, nit) if return_n_iter else (e, xs)


# ----------------------------------------------------------------------------------------------------------
# the reference's other force-closure energies: dexgrasp (||G'n||^2) and TDG (grasp-wrench-space directions)
# ----------------------------------------------------------------------------------------------------------
@_custom_op("graspqp_amd::dexgrasp_energy", mutates_args=(), device_types="cuda")
def _dexgrasp_op(contact_pts: Tensor, contact_normals: Tensor, cog: Tensor, torque_weight: float) -> Tuple[Tensor, Tensor]:
    """-> (E (B), dE/d contact_pts (B,n,3)); metrics/ops/dexgrasp.py:4-34."""
    cp, cn, cg = _c(contact_pts), _c(contact_normals), _c(cog)
    B, n, _ = cp.shape
    e = torch.empty(B, device=cp.device)
    g = torch.empty_like(cp)
    _C.call("gq_dexgrasp_energy", _C.f32(cp), _C.f32(cn), _C.f32(cg), ctypes.c_int64(B), n, float(torque_weight), None, 1.0, 0,
            _C.f32(e), _C.f32(g), _C.stream_ptr())
    return e, g


@_dexgrasp_op.register_fake
def _(contact_pts, contact_normals, cog, torque_weight):
    return contact_pts.new_empty(contact_pts.shape[0]), torch.empty_like(contact_pts)


@_custom_op("graspqp_amd::tdg_energy", mutates_args=(), device_types="cuda")
def _tdg_op(contact_pts: Tensor, contact_normals: Tensor, cog: Tensor, directions: Tensor, friction: float, obb_length: float,
            enable_density: bool, scale: float) -> Tuple[Tensor, Tensor]:
    """-> (E (B), dE/d contact_pts (B,n,3)); metrics/ops/tdg.py:147-239."""
    cp, cn, cg, dr = _c(contact_pts), _c(contact_normals), _c(cog), _c(directions)
    B, n, _ = cp.shape
    e = torch.empty(B, device=cp.device)
    g = torch.empty_like(cp)
    _C.call("gq_tdg_energy", _C.f32(cp), _C.f32(cn), _C.f32(cg), _C.f32(dr), dr.shape[0], ctypes.c_int64(B), n, float(friction),
            float(obb_length), int(bool(enable_density)), float(scale), None, 1.0, 0, _C.f32(e), _C.f32(g), _C.stream_ptr())
    return e, g


@_tdg_op.register_fake
def _(contact_pts, contact_normals, cog, directions, friction, obb_length, enable_density, scale):
    return contact_pts.new_empty(contact_pts.shape[0]), torch.empty_like(contact_pts)


def _alt_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.n_in = len(inputs)
    ctx.mark_non_differentiable(output[1])


def _alt_bwd(ctx, ge, gg):
    (g,) = ctx.saved_tensors
    return (g * ge.view(-1, 1, 1),) + (None,) * (ctx.n_in - 1)


torch.library.register_autograd("graspqp_amd::dexgrasp_energy", _alt_bwd, setup_context=_alt_setup)
torch.library.register_autograd("graspqp_amd::tdg_energy", _alt_bwd, setup_context=_alt_setup)


def dexgrasp_energy(contact_pts, contact_normals, cog, torque_weight=0.0):
    """(B,) DexGraspNet force-closure term; gradient to contact_pts (the normals are SDF constants)."""
    return _Eager.dexgrasp_energy(contact_pts, contact_normals.detach(), cog.detach(), float(torque_weight))[0]


def tdg_energy(contact_pts, contact_normals, cog, directions, friction=0.2, obb_length=0.2, enable_density=True, scale=100.0):
    return _Eager.tdg_energy(contact_pts, contact_normals.detach(), cog.detach(), directions, float(friction),
                                            float(obb_length), bool(enable_density), float(scale))[0]


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
        d.n_dofs, d.n_links = spec.n_nodes, spec.n_links  # tree joints; the pose carries spec.n_dofs actuated ones
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
        if spec.is_coupled:  # theta_tree = coupling theta_actuated + offset (ability_hand, panda)
            d.n_actuated = spec.n_dofs
            d.coupling = arr(spec.coupling, np.float32)
            d.coupling_offset = arr(spec.coupling_offset, np.float32)
        h = ctypes.c_void_p(0)
        _C.call("gq_hand_create", ctypes.byref(d), ctypes.byref(h))
        self.handle = h
        self.links = MeshSet([spec.link_faces(l) for l in range(spec.n_links)])
        _C.call("gq_meshset_build_occupancy", self.links.handle)
        self.J, self.L, self.S = spec.n_dofs, spec.n_links, spec.n_spheres
        self.hid = _register_handle(self)

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


@_custom_op("graspqp_amd::fk_contacts", mutates_args=(), device_types="cuda")
def _fk_op(hand_pose: Tensor, idx: Tensor, hand: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """hand_pose, contact idx -> (Rg (B,3,3), link_T (B,L,3,4), contact_points, contact_normals, sphere_centers,
    FK workspace holding the per-joint frames for the backward)."""
    h = _handle(hand)
    hp = _c(hand_pose)
    ix = _c(idx, torch.int64)
    B, n = ix.shape
    dev = hp.device
    Rg = torch.empty(B, 3, 3, device=dev)
    LT = torch.empty(B, h.L, 3, 4, device=dev)
    cp = torch.empty(B, n, 3, device=dev)
    cn = torch.empty(B, n, 3, device=dev)
    sc = torch.empty(B, max(h.S, 1), 3, device=dev)
    ws, nb = h.fk_ws(B, dev)
    _C.call("gq_fk_forward", h.handle, _C.f32(hp), _C.i64(ix), B, n, _C.f32(Rg), _C.f32(LT), _C.f32(cp),
            _C.f32(cn), _C.f32(sc) if h.S > 0 else None, 0.0, None, None, None, None, _C.ptr(ws), nb, _C.stream_ptr())
    return Rg, LT, cp, cn, sc[:, : h.S].contiguous(), ws


@_fk_op.register_fake
def _(hand_pose, idx, hand):
    h = _handle(hand)
    B, n = idx.shape
    e = hand_pose.new_empty
    nb = _size_call("gq_fk_workspace_bytes", h.handle, ctypes.c_int64(B))
    return (e(B, 3, 3), e(B, h.L, 3, 4), e(B, n, 3), e(B, n, 3), e(B, h.S, 3), e(nb, dtype=torch.uint8))


@_custom_op("graspqp_amd::fk_backward", mutates_args=(), device_types="cuda")
def _fk_bwd_op(hand: int, hand_pose: Tensor, idx: Tensor, Rg: Tensor, LT: Tensor, ws: Tensor, gcp: Tensor, gcn: Tensor,
               gsc: Tensor, wrench: Tensor, gRt: Tensor, gR: Tensor, has: List[bool]) -> Tensor:
    """Analytic FK backward (replaces autograd through pytorch_kinematics).  ``has`` flags which of the six gradient
    inputs (contact points, contact normals, sphere centres, link wrench, g_Rt, g_R) are present; absent ones are
    passed as empty tensors (only tensors may cross the dispatcher)."""
    h = _handle(hand)
    B, n = idx.shape
    gp = torch.empty_like(hand_pose)
    opt = lambda t, on: _C.f32(_c(t)) if on else None
    _C.call("gq_fk_backward", h.handle, _C.f32(hand_pose), _C.i64(idx), B, n, _C.f32(_c(Rg)), _C.f32(_c(LT)),
            opt(gcp, has[0]), opt(gcn, has[1]), opt(gsc, has[2] and h.S > 0), opt(wrench, has[3]), opt(gRt, has[4]), None,
            opt(gR, has[5]), _C.f32(gp), None, None, _C.ptr(ws), ws.numel(), _C.stream_ptr())
    return gp


@_fk_bwd_op.register_fake
def _(hand, hand_pose, idx, Rg, LT, ws, gcp, gcn, gsc, wrench, gRt, gR, has):
    return torch.empty_like(hand_pose)


def _fk_backward(hand, hp, ix, Rg, LT, ws, gcp=None, gcn=None, gsc=None, wrench=None, gRt=None, gR=None):
    z = hp.new_empty(0)
    args = [gcp, gcn, gsc, wrench, gRt, gR]
    return _Eager.fk_backward(hand.hid, hp, ix, Rg, LT, ws, *[z if a is None else a for a in args],
                                             [a is not None for a in args])


def _fk_setup(ctx, inputs, output):
    ctx.save_for_backward(_c(inputs[0]), _c(inputs[1], torch.int64), output[0], output[1], output[5])
    ctx.hand = inputs[2]
    ctx.mark_non_differentiable(output[5])


def _fk_bwd(ctx, gRg, gLT, gcp, gcn, gsc, gws):
    hp, ix, Rg, LT, ws = ctx.saved_tensors
    # d / d link_T is routed through hand_pen's link wrenches, never through link_T itself: autograd hands a zero (or
    # no) gradient here, which is ignored
    return _fk_backward(_handle(ctx.hand), hp, ix, Rg, LT, ws, gcp, gcn, gsc, None, None, gRg), None, None


torch.library.register_autograd("graspqp_amd::fk_contacts", _fk_bwd, setup_context=_fk_setup)


def fk_contacts(hand_pose, idx, hand: HandHandle):
    """-> (Rg (B,3,3), link_T (B,L,3,4), contact_points, contact_normals, sphere_centers, fk workspace)."""
    if not hand_pose.is_cuda:
        raise RuntimeError("graspqp_amd ops need CUDA (ROCm) tensors; got a CPU tensor")
    return _Eager.fk_contacts(hand_pose, idx, hand.hid)


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


class _JointVelocityResiduals(torch.autograd.Function):
    """residuals (B,3n) = (J theta - d_h)^2 with theta = pinv_damped(J) d_h, d_h = R' d (hand_model.py:1155-1218, coupled form),
    differentiable w.r.t. hand_pose (joint angles through the contact Jacobian -- analytic kinematic Hessian,
    gq_contact_jacobian_backward -- and the root rotation through d_h) and w.r.t. the moving directions d: what autograd
    through pytorch_kinematics gives the reference for E_manipulativity (core/energy.py:80-87).  The kinematic state
    (link transforms, FK workspace, root rotation) is the one of ``hand_pose``, passed in detached."""

    @staticmethod
    def forward(ctx, hand_pose, directions, hand, idx, Rg, LT, ws, damping):
        hp, d, R = _c(hand_pose.detach()), _c(directions.detach()), _c(Rg.detach())
        ix = _c(idx, torch.int64)
        B, n = ix.shape
        jc = contact_jacobian(hand, ix, LT, ws).reshape(B, 3 * n, hand.J)
        theta, res, _ = joint_velocities(jc, d.reshape(B, 3 * n), R, damping)
        d_h = (R.transpose(1, 2).unsqueeze(1) @ d.unsqueeze(-1)).squeeze(-1).reshape(B, 3 * n)
        r = (jc @ theta.unsqueeze(-1)).squeeze(-1) - d_h  # signed residual, hand frame
        ctx.save_for_backward(hp, d, R, ix, _c(LT.detach()), ws, jc, theta, r)
        ctx.hand, ctx.damping = hand, damping
        ctx.mark_non_differentiable(theta)
        return theta, res

    @staticmethod
    def backward(ctx, _g_theta, g_res):
        hp, d, R, ix, LT, ws, jc, theta, r = ctx.saved_tensors
        hand = ctx.hand
        B, n = ix.shape
        g = 2.0 * g_res * r                                           # d E / d r
        u, _, _ = joint_velocities(jc, g, None, ctx.damping)          # (J'J + lambda I)^-1 J' g
        w = g - (jc @ u.unsqueeze(-1)).squeeze(-1)                    # (I - J M^-1 J') g
        GJ = (w.unsqueeze(-1) * theta.unsqueeze(1) - r.unsqueeze(-1) * u.unsqueeze(1)).contiguous()  # d E / d J  (B,3n,J)
        g_dh = (-w).reshape(B, n, 3)                                  # d E / d d_h
        g_d = (R.unsqueeze(1) @ g_dh.unsqueeze(-1)).squeeze(-1)       # d_h = R' d
        gR = (d.unsqueeze(-1) * g_dh.unsqueeze(-2)).sum(1).contiguous()  # (B,3,3): dE/dR[a,c] = sum_i d[i,a] g_dh[i,c]
        g_th = torch.empty(B, hand.J, device=hp.device)
        _C.call("gq_contact_jacobian_backward", hand.handle, _C.i64(ix), ctypes.c_int64(B), n, _C.f32(LT), _C.f32(GJ),
                _C.f32(g_th), _C.ptr(ws), ws.numel(), _C.stream_ptr())
        ghp = _fk_backward(hand, hp, ix, R, LT, ws, None, None, None, None, None, gR).clone()
        ghp[:, hp.shape[1] - hand.J:] += g_th
        return ghp, g_d, None, None, None, None, None, None


def joint_velocity_residuals(hand_pose, directions, hand: HandHandle, idx, Rg, LT, ws, damping=1e-3):
    """-> (theta (B,J), residuals (B,3n)); see _JointVelocityResiduals."""
    return _JointVelocityResiduals.apply(hand_pose, directions, hand, idx, Rg, LT, ws, float(damping))


def root_pose_wxyz(hand_pose):
    """(B,7) = [translation, unit quaternion (w,x,y,z)] of hand_pose[:, :9] (fit.py:260-263)."""
    hp = _c(hand_pose.detach())
    out = torch.empty(hp.shape[0], 7, device=hp.device)
    _C.call("gq_root_pose_wxyz", _C.f32(hp), ctypes.c_int64(hp.shape[0]), hp.shape[1], _C.f32(out), _C.stream_ptr())
    return out


@_custom_op("graspqp_amd::hand_pen", mutates_args=(), device_types="cuda")
def _hand_pen_op(hand_pose: Tensor, surface_points: Tensor, batch_each: int, hand: int, Rg: Tensor, LT: Tensor,
                 penetration_only: int) -> Tuple[Tensor, Tensor, Tensor]:
    """max-over-links signed distance (inside positive) of object surface points -> (dis (B,P), argmax link, d dis / d x_h).

    The kinematic state (Rg, link_T) is passed in detached; the gradient is routed to ``hand_pose`` directly through the
    analytic FK backward (link wrenches), which is what autograd through pytorch_kinematics computes in the reference
    (hand_model.py:875-987)."""
    h = _handle(hand)
    hp = _c(hand_pose)
    sp = _c(surface_points)
    n_obj, P, _ = sp.shape
    B = hp.shape[0]
    dev = hp.device
    dis = torch.empty(B, P, device=dev)
    link = torch.zeros(B, P, dtype=torch.int32, device=dev)  # mode 1 writes link / gvec only where dis > 0
    gvec = torch.zeros(B, P, 3, device=dev)
    pws, pnb = None, 0
    if int(penetration_only) == 3:  # queue path without candidate lists (kept for A/B tests)
        pnb = _size_call("gq_hand_pen_workspace_bytes", ctypes.c_int64(B), ctypes.c_int64(P), h.L)
        pws = torch.zeros(pnb, dtype=torch.uint8, device=dev)  # queue counters must start at zero
    _C.call("gq_hand_pen_forward", h.links.handle, _C.f32(sp), n_obj, P, int(batch_each), _C.f32(hp), hp.shape[1],
            _C.f32(_c(Rg)), _C.f32(_c(LT)), int(penetration_only), _C.f32(dis), _C.i32(link), _C.f32(gvec), _C.ptr(pws), pnb,
            None, None, None, _C.stream_ptr())
    return dis, link, gvec


@_hand_pen_op.register_fake
def _(hand_pose, surface_points, batch_each, hand, Rg, LT, penetration_only):
    B, P = hand_pose.shape[0], surface_points.shape[1]
    return (hand_pose.new_empty(B, P), hand_pose.new_empty(B, P, dtype=torch.int32), hand_pose.new_empty(B, P, 3))


@_custom_op("graspqp_amd::hand_pen_backward", mutates_args=(), device_types="cuda")
def _hand_pen_bwd_op(n_links: int, surface_points: Tensor, batch_each: int, hand_pose: Tensor, Rg: Tensor, g: Tensor,
                     link: Tensor, gvec: Tensor) -> Tuple[Tensor, Tensor]:
    sp = _c(surface_points)
    n_obj, P, _ = sp.shape
    B = hand_pose.shape[0]
    wrench = torch.empty(B, n_links, 6, device=g.device)
    gRt = torch.empty(B, 12, device=g.device)
    _C.call("gq_hand_pen_backward", int(n_links), _C.f32(sp), n_obj, P, int(batch_each), _C.f32(hand_pose), hand_pose.shape[1],
            _C.f32(_c(Rg)), _C.f32(_c(g)), _C.i32(link), _C.f32(gvec), _C.f32(wrench), _C.f32(gRt), None, 0.0, None, None, None,
            _C.stream_ptr())
    return wrench, gRt


@_hand_pen_bwd_op.register_fake
def _(n_links, surface_points, batch_each, hand_pose, Rg, g, link, gvec):
    B = hand_pose.shape[0]
    return hand_pose.new_empty(B, n_links, 6), hand_pose.new_empty(B, 12)


def hand_pen(hand_pose, surface_points, batch_each, hand, idx, Rg, LT, ws, nb=None, penetration_only=False):
    """(B,P) max-over-links signed distance, differentiable w.r.t. hand_pose (see the op's docstring)."""
    return _HandPen.apply(hand_pose, surface_points, int(batch_each), hand, idx, Rg.detach(), LT.detach(), ws,
                          int(penetration_only))


class _HandPen(torch.autograd.Function):
    """Glue between two registered ops (hand_pen + fk_backward): the backward needs the hand's FK workspace and contact
    indices, which are not inputs of the distance query itself."""

    @staticmethod
    def forward(ctx, hand_pose, surface_points, batch_each, hand, idx, Rg, LT, ws, penetration_only):
        hp = _c(hand_pose.detach())
        sp = _c(surface_points)
        dis, link, gvec = _Eager.hand_pen(hp, sp, batch_each, hand.hid, Rg, LT, penetration_only)
        ctx.save_for_backward(hp, sp, idx, Rg, LT, ws, link, gvec)
        ctx.hand, ctx.batch_each = hand, batch_each
        return dis

    @staticmethod
    def backward(ctx, g):
        hp, sp, idx, Rg, LT, ws, link, gvec = ctx.saved_tensors
        hand = ctx.hand
        wrench, gRt = _Eager.hand_pen_backward(hand.L, sp, ctx.batch_each, hp, Rg, g, link, gvec)
        gp = _fk_backward(hand, hp, idx, Rg, LT, ws, None, None, None, wrench, gRt, None)
        return gp, None, None, None, None, None, None, None, None


@_custom_op("graspqp_amd::self_pen", mutates_args=(), device_types="cuda")
def _self_pen_op(centers: Tensor, hand: int) -> Tuple[Tensor, Tensor]:
    """E_spen (B,) of world sphere centres (B,S,3) and dE/dcentres (hand_model.py:989-1040)."""
    c = _c(centers)
    B = c.shape[0]
    e = torch.empty(B, device=c.device)
    g = torch.empty_like(c)
    _C.call("gq_self_pen_forward", _handle(hand).handle, _C.f32(c), B, 1.0, _C.f32(e), _C.f32(g), _C.stream_ptr())
    return e, g


@_self_pen_op.register_fake
def _(centers, hand):
    return centers.new_empty(centers.shape[0]), torch.empty_like(centers)


def _self_pen_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.mark_non_differentiable(output[1])


def _self_pen_bwd(ctx, ge, gg):
    (g,) = ctx.saved_tensors
    return g * ge.view(-1, 1, 1), None


torch.library.register_autograd("graspqp_amd::self_pen", _self_pen_bwd, setup_context=_self_pen_setup)


def self_pen(centers, hand: HandHandle):
    if hand.S == 0:
        return torch.zeros(centers.shape[0], device=centers.device)
    return _Eager.self_pen(centers, hand.hid)[0]


# ----------------------------------------------------------------------------------------------------------
# energy terms of the class surface, one launch each (csrc/terms.hip): the forward launch also writes the term's
# derivative, the backward is a broadcast multiply with the upstream row gradient
# ----------------------------------------------------------------------------------------------------------
@_custom_op("graspqp_amd::signed_distance", mutates_args=(), device_types="cuda")
def _signed_distance_op(dist_sq: Tensor, sign: Tensor, normal: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """ObjectModel.cal_distance (object_model.py:222-227): (sqrt(dist_sq + 1e-8) * (-sign), normal * sign, d dis / d dist_sq)."""
    d2, sg, nr = _c(dist_sq), _c(sign, torch.int32), _c(normal)
    dis, nout, g = torch.empty_like(d2), torch.empty_like(nr), torch.empty_like(d2)
    _C.call("gq_signed_distance", _C.f32(d2), _C.i32(sg), _C.f32(nr), d2.numel(), _C.f32(dis), _C.f32(nout), _C.f32(g),
            _C.stream_ptr())
    return dis, nout, g


@_signed_distance_op.register_fake
def _(dist_sq, sign, normal):
    return torch.empty_like(dist_sq), torch.empty_like(normal), torch.empty_like(dist_sq)


def _signed_distance_setup(ctx, inputs, output):
    ctx.save_for_backward(output[2])
    ctx.mark_non_differentiable(output[1], output[2])


def _signed_distance_bwd(ctx, g_dis, g_n, g_g):
    (g,) = ctx.saved_tensors
    return g_dis * g, None, None


torch.library.register_autograd("graspqp_amd::signed_distance", _signed_distance_bwd, setup_context=_signed_distance_setup)


def signed_distance(dist_sq, sign, normal):
    dis, nout, _ = _Eager.signed_distance(dist_sq, sign, normal)
    return dis, nout


@_custom_op("graspqp_amd::energy_dis", mutates_args=(), device_types="cuda")
def _energy_dis_op(distance: Tensor, obj_normal: Tensor, hand_normal: Tensor, with_normals: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """E_dis (energy.py:25-28): with_normals ("gendexgrasp") sum_j exp(1 - (-n_obj . n_hand)) |d|, else sum_j |d|;
    -> (e (B,), d e / d distance (B,n), d e / d hand_normal (B,n,3))."""
    d = _c(distance)
    B, n = d.shape
    e, gd = torch.empty(B, device=d.device), torch.empty_like(d)
    if with_normals:
        on, hn = _c(obj_normal), _c(hand_normal)
        gh = torch.empty_like(hn)
        _C.call("gq_energy_dis", _C.f32(d), _C.f32(on), _C.f32(hn), B, n, _C.f32(e), _C.f32(gd), _C.f32(gh), _C.stream_ptr())
    else:
        gh = d.new_zeros(B, n, 3)
        _C.call("gq_energy_dis", _C.f32(d), None, None, B, n, _C.f32(e), _C.f32(gd), None, _C.stream_ptr())
    return e, gd, gh


@_energy_dis_op.register_fake
def _(distance, obj_normal, hand_normal, with_normals):
    B, n = distance.shape
    return distance.new_empty(B), torch.empty_like(distance), distance.new_empty(B, n, 3)


def _energy_dis_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], output[2])
    ctx.with_normals = bool(inputs[3])
    ctx.mark_non_differentiable(output[1], output[2])


def _energy_dis_bwd(ctx, ge, g1, g2):
    gd, gh = ctx.saved_tensors
    return ge.unsqueeze(-1) * gd, None, (ge.view(-1, 1, 1) * gh) if ctx.with_normals else None, None


torch.library.register_autograd("graspqp_amd::energy_dis", _energy_dis_bwd, setup_context=_energy_dis_setup)


def energy_dis(distance, obj_normal, hand_normal, with_normals=True):
    return _Eager.energy_dis(distance, obj_normal.detach(), hand_normal, bool(with_normals))[0]


@_custom_op("graspqp_amd::energy_joints", mutates_args=(), device_types="cuda")
def _energy_joints_op(hand_pose: Tensor, lower: Tensor, upper: Tensor) -> Tuple[Tensor, Tensor]:
    """E_joints (energy.py:47-52) over the last len(lower) columns of hand_pose -> (e (B,), d e / d hand_pose (B,D))."""
    hp, lo, hi = _c(hand_pose), _c(lower), _c(upper)
    B, D = hp.shape
    e, g = torch.empty(B, device=hp.device), torch.empty_like(hp)
    _C.call("gq_energy_joints", _C.f32(hp), _C.f32(lo), _C.f32(hi), B, D, lo.numel(), _C.f32(e), _C.f32(g), _C.stream_ptr())
    return e, g


@_energy_joints_op.register_fake
def _(hand_pose, lower, upper):
    return hand_pose.new_empty(hand_pose.shape[0]), torch.empty_like(hand_pose)


def _energy_joints_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.mark_non_differentiable(output[1])


def _energy_joints_bwd(ctx, ge, g1):
    (g,) = ctx.saved_tensors
    return ge.unsqueeze(-1) * g, None, None


torch.library.register_autograd("graspqp_amd::energy_joints", _energy_joints_bwd, setup_context=_energy_joints_setup)


def energy_joints(hand_pose, lower, upper):
    return _Eager.energy_joints(hand_pose, lower, upper)[0]


@_custom_op("graspqp_amd::energy_pen", mutates_args=(), device_types="cuda")
def _energy_pen_op(distances: Tensor) -> Tensor:
    """E_pen (energy.py:58-61): sum over the surface points of where(distances <= 0, 0, distances) -> (B,)."""
    d = _c(distances)
    B, P = d.shape
    e = torch.empty(B, device=d.device)
    _C.call("gq_energy_pen", _C.f32(d), B, P, _C.f32(e), _C.stream_ptr())
    return e


@_energy_pen_op.register_fake
def _(distances):
    return distances.new_empty(distances.shape[0])


def _energy_pen_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _energy_pen_bwd(ctx, ge):
    (d,) = ctx.saved_tensors
    return torch.where(d > 0, ge.unsqueeze(-1), ge.new_zeros(()))


torch.library.register_autograd("graspqp_amd::energy_pen", _energy_pen_bwd, setup_context=_energy_pen_setup)


def energy_pen(distances):
    return _Eager.energy_pen(distances)


# eager routes of the registered ops (see the module docstring)
_eager("sdf_backward", _sdf_backward)
_eager("compute_sdf", _compute_sdf_op, _sdf_bwd, _sdf_setup)
_eager("sdf_meshset", _sdf_meshset_op, _sdf_ms_bwd, _sdf_ms_setup)
_eager("sdf_bvh", _sdf_bvh_op, _sdf_bvh_bwd, _sdf_ms_setup)
_eager("box_qp", _box_qp_op, _box_qp_bwd, _box_qp_setup)
_eager("box_qp_backward", _box_qp_bwd_op)
_eager("lsq_box_qp", _lsq_box_qp_op, _lsq_bwd, _lsq_setup)
_eager("lsq_box_qp_backward", _lsq_box_qp_bwd_op)
_eager("fc_energy", _fc_energy_op, _fc_bwd, _fc_setup)
_eager("fc_energy_backward", _fc_energy_bwd_op)
_eager("dexgrasp_energy", _dexgrasp_op, _alt_bwd, _alt_setup)
_eager("tdg_energy", _tdg_op, _alt_bwd, _alt_setup)
_eager("fk_contacts", _fk_op, _fk_bwd, _fk_setup)
_eager("fk_backward", _fk_bwd_op)
_eager("hand_pen", _hand_pen_op)
_eager("hand_pen_backward", _hand_pen_bwd_op)
_eager("self_pen", _self_pen_op, _self_pen_bwd, _self_pen_setup)
_eager("signed_distance", _signed_distance_op, _signed_distance_bwd, _signed_distance_setup)
_eager("energy_dis", _energy_dis_op, _energy_dis_bwd, _energy_dis_setup)
_eager("energy_joints", _energy_joints_op, _energy_joints_bwd, _energy_joints_setup)
_eager("energy_pen", _energy_pen_op, _energy_pen_bwd, _energy_pen_setup)
