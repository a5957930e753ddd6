"""``calculate_energy`` with the reference's signature (core/energy.py:6-89), composed from the HIP-backed ops.

This is the autograd ("plugin surface") route: every term is a torch tensor whose backward runs the HIP
backward kernels, so a ``fit.py``-style loop (``new_energy.sum().backward()``) works unchanged.  The
``GraspStepper`` in ``graspqp_amd.stepper`` runs the same kernels without autograd for throughput.
"""

import torch

from .. import ops


def calculate_energy(hand_model, object_model, energy_fnc=None, energy_names=[], method="gendexgrasp", svd_gain=0.1):
    losses = {}
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

    if "E_prior" in energy_names:  # energy.py:68-74: the grasp axis should point down
        forward_axis = (hand_model.global_rotation @ hand_model.grasp_axis.view(1, -1, 1)).view(-1, 3)
        axis_prior = torch.tensor([0, 0, -1], dtype=torch.float, device=forward_axis.device).view(1, 3)
        losses["E_prior"] = 1 - torch.sum(forward_axis * axis_prior, dim=-1)

    if "E_wall" in energy_names:  # energy.py:76-78: hand surface samples below the table plane z = 0
        z_height = hand_model.get_surface_points()[..., -1].clamp(max=0.0)
        losses["E_wall"] = z_height.abs().sum(-1)

    if "E_manipulativity" in energy_names:
        # energy.py:80-87.  Value only: the reference differentiates it through the Jacobian and the pseudo-inverse, but
        # scripts/fit.py cannot select it (no weight for it: fit.py:363-371 raises on the unknown name), so no backward here
        if hand_model.hand_pose.requires_grad:
            import warnings

            warnings.warn("graspqp_amd: E_manipulativity is VALUE-ONLY (no gradient reaches hand_pose through it; the reference "
                          "differentiates it through the contact Jacobian and its pseudo-inverse, core/energy.py:80-87). "
                          "Weighting it in an energy that is back-propagated adds nothing to the gradient.", RuntimeWarning,
                          stacklevel=2)
        E_jacobian = hand_model.get_manipulability(
            contact_normal * distance.detach().unsqueeze(-1).abs().clamp(min=5e-3), hand_model.contact_point_indices)
        losses["E_manipulativity"] = E_jacobian.mean(-1)
    return losses
