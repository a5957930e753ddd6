"""Oracle for the reference's other force-closure energies.  TEST INFRASTRUCTURE.

``dexgrasp_e_fc`` restates ``metrics/ops/dexgrasp.py:4-34`` + ``DexgraspSpanMetric.forward`` (:46-68),
``tdg_energy`` restates ``TDGEnergy.forward / GWS / construct_grasp_matrix / utils_1axis_to_3axes / estimate_density``
(``metrics/ops/tdg.py:75-216``) behind ``TDGSpanMetric.forward`` (:233-239).  Both reference modules are plain torch
and import in the build container: PINNED by fixtures ``alt_metrics_*.npz`` (tools/make_golden.py::gen_alt_metrics).
"""

import math

import torch


def dexgrasp_e_fc(contact_pts, contact_normals, cog, torque_weight=0.0):
    r = contact_pts - cog.unsqueeze(1)
    f = contact_normals.sum(1)
    tau = torque_weight * torch.linalg.cross(contact_normals, r, dim=-1).sum(1)
    return (f**2).sum(-1) + (tau**2).sum(-1)


def _normalize(v):
    return v / torch.clamp(v.norm(dim=-1, p=2, keepdim=True), min=1e-12)


def tdg_energy(contact_pts, contact_normals, cog, directions, miu=0.2, obb_length=0.2, enable_density=True, scale=100.0):
    """contact_pts (B,n,3), contact_normals (B,n,3), cog (B,3), directions (P,3) -> (B,)"""
    dt = contact_pts.dtype
    pos = contact_pts - cog.unsqueeze(1)
    a0 = contact_normals
    b1 = torch.tensor([0.0, 1.0, 0.0], dtype=dt).view(1, 1, 3)
    b2 = torch.tensor([0.0, 0.0, 1.0], dtype=dt).view(1, 1, 3)
    proj = (a0 * b1).sum(-1, keepdim=True).abs()
    a1 = torch.where(proj > 0.99, b2, b1)
    a1 = _normalize(a1 - (a1 * a0).sum(-1, keepdim=True) * a0).detach()
    a1 = _normalize(a1 - (a1 * a0).sum(-1, keepdim=True) * a0)
    a2 = torch.linalg.cross(a0, a1, dim=-1)
    r = pos / obb_length
    G = torch.stack([torch.cat([a, torch.linalg.cross(r, a, dim=-1)], dim=-1) for a in (a0, a1, a2)], dim=-1)  # (B,n,6,3)
    u6 = torch.cat([directions, torch.zeros_like(directions)], dim=-1).unsqueeze(0)  # (1,P,6)
    dF = _normalize((u6.unsqueeze(1) @ G).transpose(2, 1))  # (B,P,n,3)
    c = torch.tensor([1.0, 0.0, 0.0], dtype=dt).view(1, 1, 1, 3)
    proj_cn = (dF * c).sum(-1, keepdim=True)
    perp = dF - proj_cn * c
    ang = torch.acos(torch.clamp(proj_cn, min=-1, max=1))
    ba = math.atan(miu)
    r1, r2, r3 = ang <= ba, (ang > ba) & (ang <= math.pi / 2), ang > math.pi / 2
    pn = perp.norm(dim=-1, keepdim=True)
    help3 = pn / (pn - 2 * miu * torch.clamp(proj_cn, max=0))
    help2 = c + miu * _normalize(perp)
    arg = r1 * (c + perp / torch.clamp(proj_cn, min=math.cos(ba) / 2)) + r2 * help2 + r3 * help3 * help2
    w = (G.unsqueeze(1) @ arg.unsqueeze(-1)).squeeze(-1)  # (B,P,n,6)
    if enable_density:
        cos_t = (a0.unsqueeze(-2) * a0.unsqueeze(-3)).sum(-1)
        rho = (1 / torch.clamp(torch.clamp(cos_t, min=0).sum(-1), min=1e-4)).detach()
        W = (w * rho.unsqueeze(1).unsqueeze(-1)).sum(2)
    else:
        W = w.sum(2)
    cos_wt = (_normalize(W) * u6).sum(-1)
    return scale * (1 - cos_wt).mean(-1)
