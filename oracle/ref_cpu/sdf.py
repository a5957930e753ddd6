"""Oracle mesh signed-distance query with the TorchSDF output contract.  TEST INFRASTRUCTURE.

TorchSDF (renezurbruegg fork, no pinned commit; submodule directory empty in the reference) is
absent, so this restates its documented behaviour from the reference call sites
(``object_model.py:220,241-246``; ``hand_model.py:953,975-976``) and the Kaolin
``unbatched_triangle_distance`` algorithm it derives from: brute force over all faces, exact
closest point on each triangle (vertex / edge / interior regions), first strict minimum wins.

  compute_sdf(points (N,3), face_verts (F,3,3)) ->
      dist_sq (N,)   squared distance to the closest point            [differentiable wrt points]
      sign    (N,)   int32, +1 outside / -1 inside, from dot(p - closest, normal of closest face) >= 0
      normal  (N,3)  unit (p - closest)/|p - closest| (face normal when the point lies on the surface)
      closest (N,3)  closest point on the mesh
  d(dist_sq)/d(points) = 2 (p - closest); no gradient to faces, none through sign/normal/closest.

PARITY UNPINNED: the reference has no test of this op and its source is not in the tree; the
sign rule at edge/vertex ties and the on-surface normal are this restatement's choices.
"""

import torch


def index_vertices_by_faces(verts: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """verts (V,3), faces (F,3) long -> (F,3,3)  (torchsdf.index_vertices_by_faces)."""
    return verts[faces]


def closest_point_on_triangles(p: torch.Tensor, tri: torch.Tensor) -> torch.Tensor:
    """p (N,1,3), tri (1,F,3,3) -> closest points (N,F,3).  Ericson, Real-Time Collision Detection 5.1.5."""
    a, b, c = tri[..., 0, :], tri[..., 1, :], tri[..., 2, :]
    ab, ac = b - a, c - a
    ap = p - a
    d1 = (ab * ap).sum(-1)
    d2 = (ac * ap).sum(-1)
    bp = p - b
    d3 = (ab * bp).sum(-1)
    d4 = (ac * bp).sum(-1)
    cp = p - c
    d5 = (ab * cp).sum(-1)
    d6 = (ac * cp).sum(-1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4

    def safe(n, d):
        return n / torch.where(d == 0, torch.ones_like(d), d)

    # interior
    den = va + vb + vc
    v = safe(vb, den)
    w = safe(vc, den)
    q = a + ab * v[..., None] + ac * w[..., None]
    # edge BC
    m = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)
    wbc = safe(d4 - d3, (d4 - d3) + (d5 - d6))
    q = torch.where(m[..., None], b + (c - b) * wbc[..., None], q)
    # edge AC
    m = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
    wac = safe(d2, d2 - d6)
    q = torch.where(m[..., None], a + ac * wac[..., None], q)
    # vertex C
    m = (d6 >= 0) & (d5 <= d6)
    q = torch.where(m[..., None], c.expand_as(q), q)
    # edge AB
    m = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
    vab = safe(d1, d1 - d3)
    q = torch.where(m[..., None], a + ab * vab[..., None], q)
    # vertex B
    m = (d3 >= 0) & (d4 <= d3)
    q = torch.where(m[..., None], b.expand_as(q), q)
    # vertex A
    m = (d1 <= 0) & (d2 <= 0)
    q = torch.where(m[..., None], a.expand_as(q), q)
    return q


def _sdf_forward(points: torch.Tensor, face_verts: torch.Tensor, chunk: int = 2048):
    N = points.shape[0]
    dt = points.dtype
    fv = face_verts.to(dt)
    fn = torch.linalg.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=-1)  # (F,3) un-normalised
    d2o = torch.empty(N, dtype=dt)
    sgn = torch.empty(N, dtype=torch.int32)
    nrm = torch.empty(N, 3, dtype=dt)
    cls = torch.empty(N, 3, dtype=dt)
    # keep the (chunk x F) temporaries around ~64 MB
    F = max(int(fv.shape[0]), 1)
    chunk = max(64, min(chunk, int(2_000_000 // F) or 64))
    for s in range(0, N, chunk):
        p = points[s : s + chunk]
        q = closest_point_on_triangles(p[:, None, :], fv[None])  # (n,F,3)
        d2 = ((p[:, None, :] - q) ** 2).sum(-1)
        d2 = torch.where(torch.isnan(d2), torch.full_like(d2, float("inf")), d2)
        best = torch.argmin(d2, dim=1)  # first minimum wins
        ar = torch.arange(p.shape[0])
        qb = q[ar, best]
        diff = p - qb
        db = d2[ar, best]
        nb = fn[best]
        s_ = torch.where((diff * nb).sum(-1) >= 0, 1, -1).to(torch.int32)
        dist = torch.sqrt(db)
        unit_fn = nb / torch.linalg.norm(nb, dim=-1, keepdim=True).clamp_min(1e-30)
        n_ = torch.where((db > 0)[:, None], diff / dist.clamp_min(1e-30)[:, None], unit_fn)
        d2o[s : s + chunk] = db
        sgn[s : s + chunk] = s_
        nrm[s : s + chunk] = n_
        cls[s : s + chunk] = qb
    return d2o, sgn, nrm, cls


class _ComputeSDF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, face_verts):
        d2, sgn, nrm, cls = _sdf_forward(points.detach(), face_verts.detach())
        ctx.save_for_backward(points.detach(), cls)
        ctx.mark_non_differentiable(sgn, nrm, cls)
        return d2, sgn, nrm, cls

    @staticmethod
    def backward(ctx, g_d2, g_sgn, g_nrm, g_cls):
        points, cls = ctx.saved_tensors
        return 2.0 * (points - cls) * g_d2[:, None], None


def compute_sdf(points: torch.Tensor, face_verts: torch.Tensor):
    return _ComputeSDF.apply(points, face_verts)
