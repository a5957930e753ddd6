"""``calculate_energy`` with the reference's signature (core/energy.py:6-89), composed from the HIP-backed ops.

This is the autograd ("plugin surface") route: every term is a torch tensor whose backward runs the HIP
backward kernels, so a ``fit.py``-style loop (``new_energy.sum().backward()``) works unchanged.  The
``GraspStepper`` in ``graspqp_amd.stepper`` runs the same kernels without autograd for throughput.
"""

import os

import torch

from .. import ops


FUSED = os.environ.get("GRASPQP_FUSED_ENERGY", "1") != "0"  # False: always compose the terms from the separate ops


class _FusedTerms(torch.autograd.Function):
    """The five terms of scripts/fit.py's energy as ONE autograd node on ``hand_pose``: the forward bodies of the registered
    ops back to back (object SDF at the contacts, signed distance, E_dis, E_fc, E_joints, hand penetration, E_pen, E_spen),
    and one backward that feeds every term's derivative -- scaled by its upstream row gradient -- into a single analytic FK
    backward.  Same kernels and same numbers as the term-by-term composition below; a fit.py-shaped loop is host-bound and
    a dozen autograd nodes with Python backwards cost more host time than the kernels take."""

    @staticmethod
    def forward(ctx, hand_pose, hand_model, object_model, fc_cfg, with_normals):
        hm, om, hand = hand_model, object_model, hand_model._hand
        hp = ops._c(hand_pose.detach())
        cp, hn = ops._c(hm.contact_points.detach()), ops._c(hm.contact_normals.detach())
        Rg, LT = ops._c(hm.global_rotation.detach()), ops._c(hm.current_status.detach())
        B, n, _ = cp.shape
        body = lambda op: op._init_fn
        flat = cp.reshape(-1, 3)
        d2, sgn, nrm, cls = body(ops._sdf_meshset_op)(flat, om._meshset.hid, om.batch_size_each * n)
        dis, onrm, g_sd = body(ops._signed_distance_op)(d2, sgn, nrm)
        distance, contact_normal = dis.reshape(B, n), onrm.reshape(B, n, 3)
        e_dis, gd, gh = body(ops._energy_dis_op)(distance, contact_normal, hn, with_normals)
        c = dict(ops.FC_DEFAULTS)
        c.update(fc_cfg)
        cfg = (int(c["n_cone_vecs"]), float(c["friction"]), float(c["torque_weight"]), float(c["max_limit"]), float(c["svd_gain"]),
               float(c["values_gain"]), float(c["eps"]), int(c["max_iter"]))
        cog = ops._c(om.cog.detach())
        e_fc, xs, _nit, fc_ws = body(ops._fc_energy_op)(cp, contact_normal, cog, *cfg)
        e_j, g_pose = body(ops._energy_joints_op)(hp, hm.joints_lower, hm.joints_upper)
        surf = om.surface_points_each
        pen_dis, link, gvec = body(ops._hand_pen_op)(hp, surf, om.batch_size_each, hand.hid, Rg, LT, 1)
        e_pen = body(ops._energy_pen_op)(pen_dis)
        if hand.S > 0:
            e_spen, g_sc = body(ops._self_pen_op)(hm._sphere_centers.detach(), hand.hid)
        else:
            e_spen, g_sc = torch.zeros(B, device=cp.device), None
        ctx.save_for_backward(hp, cp, flat, cls, g_sd, gd, gh, contact_normal, cog, fc_ws, g_pose, pen_dis, link, gvec, Rg, LT,
                              hm._fk_ws, hm._fk_idx, surf, *([g_sc] if g_sc is not None else []))
        ctx.hand, ctx.cfg, ctx.be, ctx.with_normals = hand, cfg, om.batch_size_each, with_normals
        ctx.mark_non_differentiable(distance, contact_normal, xs)
        return e_dis, e_fc, e_j, e_pen, e_spen, distance, contact_normal, xs

    @staticmethod
    def backward(ctx, g_dis, g_fc, g_j, g_pen, g_spen, _g1, _g2, _g3):
        (hp, cp, flat, cls, g_sd, gd, gh, contact_normal, cog, fc_ws, g_pose, pen_dis, link, gvec, Rg, LT, fk_ws, idx, surf,
         *rest) = ctx.saved_tensors
        hand = ctx.hand
        body = lambda op: op._init_fn
        B, n, _ = cp.shape
        k, mu, tw, _ml, sg, vg, _eps, _mi = ctx.cfg
        # contacts: E_fc + E_dis (through the signed distance and the squared distance of the SDF)
        gcp = body(ops._fc_energy_bwd_op)(cp, contact_normal, cog, g_fc, fc_ws, k, mu, tw, sg, vg)
        g_d2 = (g_dis.unsqueeze(-1) * gd).reshape(-1) * g_sd
        gcp = gcp + body(ops._sdf_backward)(g_d2, flat, cls).reshape(B, n, 3)
        gcn = (g_dis.view(-1, 1, 1) * gh) if ctx.with_normals else None
        gsc = (rest[0] * g_spen.view(-1, 1, 1)) if rest else None
        # hand penetration: upstream on the penetrating surface points -> link wrenches
        g = torch.where(pen_dis > 0, g_pen.unsqueeze(-1), g_pen.new_zeros(()))
        wrench, gRt = body(ops._hand_pen_bwd_op)(hand.L, surf, ctx.be, hp, Rg, g, link, gvec)
        z = hp.new_empty(0)
        args = [gcp, gcn, gsc, wrench, gRt, None]
        ghp = body(ops._fk_bwd_op)(hand.hid, hp, idx, Rg, LT, fk_ws, *[z if a is None else a for a in args],
                                   [a is not None for a in args])
        return ghp + g_j.unsqueeze(-1) * g_pose, None, None, None, None


def _fusable(hand_model, object_model, energy_fnc, method, energy_names):
    from ..metrics.ops.registry import SpanMetricWrapper

    return (FUSED and not ops._ROUTE["dispatcher"] and not torch.compiler.is_compiling() and isinstance(energy_fnc, SpanMetricWrapper)
            and "E_manipulativity" not in energy_names  # its directions carry a gradient to `distance`
            and method in ("gendexgrasp", "dexgraspnet") and getattr(object_model, "_meshset", None) is not None
            and object_model.surface_points_each is not None and getattr(hand_model, "_fk_ws", None) is not None)


def calculate_energy(hand_model, object_model, energy_fnc=None, energy_names=[], method="gendexgrasp", svd_gain=0.1):
    losses = {}
    if _fusable(hand_model, object_model, energy_fnc, method, energy_names):
        object_model.attach(hand_model)
        (losses["E_dis"], losses["E_fc"], losses["E_joints"], losses["E_pen"], losses["E_spen"], distance, contact_normal,
         _lambda) = _FusedTerms.apply(hand_model.hand_pose, hand_model, object_model, energy_fnc.fc_config(svd_gain=svd_gain),
                                      method == "gendexgrasp")
        return _extra_terms(losses, hand_model, energy_names, distance, contact_normal)
    distance, contact_normal = object_model.cal_distance(hand_model.contact_points)
    # (each term below is one launch that also writes its derivative, csrc/terms.hip: the torch expressions of the
    # reference, energy.py:25-28,47-52,58-61, cost a dozen launches each, forward and backward, in a host-bound loop)
    if method == "dexgraspnet":  # sum |d|
        losses["E_dis"] = ops.energy_dis(distance, contact_normal, hand_model.contact_normals, with_normals=False)
    elif method == "gendexgrasp":  # sum exp(1 - (-n_obj . n_hand)) |d|
        losses["E_dis"] = ops.energy_dis(distance, contact_normal, hand_model.contact_normals, with_normals=True)
    else:
        raise ValueError(f"Unknown method: {method}")

    E_fc, _lambda = energy_fnc(contact_pts=hand_model.contact_points, contact_normals=contact_normal, sdf=distance,
                               cog=object_model.cog, with_solution=True, svd_gain=svd_gain)
    losses["E_fc"] = E_fc

    losses["E_joints"] = ops.energy_joints(hand_model.hand_pose, hand_model.joints_lower, hand_model.joints_upper)

    object_model.attach(hand_model)
    distances = hand_model.cal_distance(object_model.surface_points_each, penetration_only=True)  # only dis > 0 is used
    losses["E_pen"] = ops.energy_pen(distances)  # sum of where(distances <= 0, 0, distances)
    losses["E_spen"] = hand_model.self_penetration()

    return _extra_terms(losses, hand_model, energy_names, distance, contact_normal)


def _extra_terms(losses, hand_model, energy_names, distance, contact_normal):
    if "E_prior" in energy_names:  # energy.py:68-74: the grasp axis should point down
        forward_axis = (hand_model.global_rotation @ hand_model.grasp_axis.view(1, -1, 1)).view(-1, 3)
        axis_prior = torch.tensor([0, 0, -1], dtype=torch.float, device=forward_axis.device).view(1, 3)
        losses["E_prior"] = 1 - torch.sum(forward_axis * axis_prior, dim=-1)

    if "E_wall" in energy_names:  # energy.py:76-78: hand surface samples below the table plane z = 0
        z_height = hand_model.get_surface_points()[..., -1].clamp(max=0.0)
        losses["E_wall"] = z_height.abs().sum(-1)

    if "E_manipulativity" in energy_names:
        # energy.py:80-87: mean squared contact velocity the joints cannot produce when the contacts move along
        # normal * max(|distance|, 5 mm); differentiable (ops.joint_velocity_residuals: analytic kinematic Hessian)
        E_jacobian = hand_model.get_manipulability(
            contact_normal * distance.unsqueeze(-1).abs().clamp(min=5e-3), hand_model.contact_point_indices)
        losses["E_manipulativity"] = E_jacobian.mean(-1)
    return losses
