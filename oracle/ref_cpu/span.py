"""Oracle force-closure metric: friction cone, grasp matrix, box-QP, svd scale, E_fc.  TEST INFRASTRUCTURE.

Restates ``OverallFrictionConeSpanMetric.forward`` (reference ``metrics/ops/span.py:313-415``),
``get_friction_cone`` (``span.py:263-295``) and ``SpanMetricWrapper.forward``
(``metrics/ops/registry.py:31-89``).  Pinned against those reference files executed in the build
container with the reference's ScipyLsqSolver (fixtures ``tests/golden/span_*.npz``).
"""

import math

import torch

from . import qp as _qp


def friction_cone_edges(normals: torch.Tensor, mu: float = 0.2, k: int = 4) -> torch.Tensor:
    """normals (B,n,3) -> cone edge directions (B, n*k, 3), contact-major / edge-minor, each / k.

    span.py:263-295 -- note the tangents are NOT normalised and the helper vector b1 flips its y
    component when it is within ~26 deg of the normal.
    """
    B, n, _ = normals.shape
    dt = normals.dtype
    b1 = torch.ones(B, n, 3, dtype=dt) / math.sqrt(3)
    dot = (b1 * normals).sum(-1) / (torch.linalg.norm(normals, dim=-1) + 1e-6)
    b1 = b1.clone()
    b1[..., 1] = b1[..., 1] - 2 * (dot > 0.9).to(dt)
    t1 = torch.linalg.cross(normals, b1, dim=-1)
    t2 = torch.linalg.cross(normals, t1, dim=-1)
    c = math.sqrt(1 - mu**2)
    if k == 4:
        edges = [mu * t1 + c * normals, mu * t2 + c * normals, -mu * t1 + c * normals, -mu * t2 + c * normals]
    else:
        edges = []
        for i in range(k):
            a = 2 * math.pi / k * i
            edges.append(mu * (math.cos(a) * t1 + math.sin(a) * t2) + c * normals)
    return torch.stack(edges, dim=-2).flatten(-3, -2) / len(edges)


def grasp_matrix(contact_pts, contact_normals, cog, mu=0.2, k=4, torque_weight=5.0):
    """F (B,6,n*k) = [f ; w (r x f)]'   (span.py:341-346)."""
    r = contact_pts - cog.unsqueeze(1)
    f = friction_cone_edges(contact_normals, mu, k)
    r = r.repeat_interleave(k, dim=-2)
    tau = torch.linalg.cross(r, f, dim=-1) * torque_weight
    return torch.cat([f, tau], dim=-1).transpose(1, 2)


def svd_scale(F):
    """(prod of singular values of F)^(1/6)  (span.py:402)."""
    return torch.linalg.svdvals(F).prod(-1) ** (1.0 / F.shape[-2])


def span_metric(
    contact_pts,
    contact_normals,
    cog,
    mu=0.2,
    k=4,
    max_limit=20.0,
    torque_weight=5.0,
    eps=5e-2,
    maxIter=12,
    box_form=False,
    solver=None,
):
    """-> (val (B,), svd (B,), x (B,nz), F).  Bounds 1 <= x <= max_limit + 1, b = 0 (span.py:333,348-349)."""
    F = grasp_matrix(contact_pts, contact_normals, cog, mu, k, torque_weight)
    b = torch.zeros(F.shape[0], 6, dtype=F.dtype)
    if solver is None:
        val, x = _qp.lsq_box_qp(F, b, 1.0, max_limit + 1.0, eps, maxIter, box_form)
    else:
        val, x = solver(F, b, 1.0, max_limit + 1.0)
    return val, svd_scale(F), x, F


def e_fc(contact_pts, contact_normals, cog, svd_gain=0.1, values_gain=2.0, **kw):
    """E_fc = 2 (val + 0.01) exp(-svd_gain * svd)  and the per-contact force sums (registry.py:82-87)."""
    k = kw.get("k", 4)
    val, svd, x, _ = span_metric(contact_pts, contact_normals, cog, **kw)
    e = values_gain * (val + 1e-2) * torch.exp(-svd_gain * svd)
    return e, x.view(x.shape[0], -1, k).sum(-1)
