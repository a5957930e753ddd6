"""Oracle energy composition.  TEST INFRASTRUCTURE.

Restates ``calculate_energy`` (reference ``core/energy.py:6-89``, method="gendexgrasp") and the
weighted sum of ``scripts/fit.py:363-371,434-438``.  Pinned: the reference's own energy.py is run
on the same oracle models by tools/make_golden.py and must agree to round-off.
"""

import torch

from . import span

DEFAULT_WEIGHTS = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}  # fit.py:51-55


def calculate_energy(hand, obj, svd_gain=0.1, mu=0.2, k=4, max_limit=20.0, box_form=False, fc_solver=None, e_fc_fn=None):
    """-> dict of (B,) tensors E_dis, E_fc, E_joints, E_pen, E_spen, plus '_x' (QP solution sums).
    ``e_fc_fn(contact_points, object_normals, cog) -> (B,)`` replaces the graspqp metric (energy types dexgrasp / tdg)."""
    losses = {}
    distance, contact_normal = obj.cal_distance(hand.contact_points)
    nH = hand.contact_normals
    # energy.py:25-28: (1 - sum((-vC) * nH)).exp() * |d|
    losses["E_dis"] = ((1 - ((-contact_normal) * nH).sum(-1)).exp() * distance.abs()).sum(-1)
    if e_fc_fn is not None:
        e_fc, xs = e_fc_fn(hand.contact_points, contact_normal, obj.cog), None
    else:
        e_fc, xs = span.e_fc(
            hand.contact_points, contact_normal, obj.cog, svd_gain=svd_gain, mu=mu, k=k, max_limit=max_limit,
            box_form=box_form, solver=fc_solver,
        )
    losses["E_fc"] = e_fc
    th = hand.hand_pose[:, 9:]
    losses["E_joints"] = ((th > hand.joints_upper) * (th - hand.joints_upper)).sum(-1) + (
        (th < hand.joints_lower) * (hand.joints_lower - th)
    ).sum(-1)
    scale = obj.object_scale_tensor.flatten().unsqueeze(1).unsqueeze(2)
    pts = obj.surface_points_tensor * scale
    d = hand.cal_distance(pts)
    d = torch.where(d <= 0, torch.zeros_like(d), d)  # energy.py:60 (in-place zeroing: no gradient there)
    losses["E_pen"] = d.sum(-1)
    losses["E_spen"] = hand.self_penetration()
    losses["_x"] = xs
    return losses


def optional_terms(hand):
    """energy.py:68-78: E_prior = 1 - (R grasp_axis).(0,0,-1); E_wall = sum |min(z, 0)| over the hand's surface samples."""
    fwd = (hand.global_rotation @ hand.grasp_axis.view(1, -1, 1)).view(-1, 3)
    e_prior = 1 - (fwd * torch.tensor([0.0, 0.0, -1.0], dtype=fwd.dtype)).sum(-1)
    e_wall = hand.get_surface_points()[..., -1].clamp(max=0.0).abs().sum(-1)
    return {"E_prior": e_prior, "E_wall": e_wall}


def total_energy(losses, weights=None):
    w = DEFAULT_WEIGHTS if weights is None else weights
    e = 0
    for name, wt in w.items():
        if wt > 0 and name in losses:
            e = e + wt * losses[name]
    return e
