"""export_poses -- the reference's grasp snapshot (scripts/fit.py:224-300) on the HIP kernels.

Writes one ``<object><suffix>.dexgrasp.pt`` per object with exactly the keys the reference's consumer reads
(graspqp_isaaclab/src/graspqp_isaaclab/utils/data.py:105-140): ``values`` (B,), ``parameters`` {joint name -> (B,),
"root_pose" -> (B,7) = [t, q_wxyz]}, ``grasp_velocities`` / ``full_grasp_velocities`` / ``grasp_velocities_off``
{joint name -> (B,)}, ``contact_idx`` (B,n), ``grasp_type``, ``contact_links``.  All tensors on the CPU, float32 /
int64, like the reference's.

The three velocity sets are joint velocities that close the hand along the object normals at the contacts
(``HandModel.get_req_joint_velocities``: linear contact Jacobian + damped pseudo-inverse): at the selected contacts
scaled by 5 |d|, at ALL contact candidates, and at the selected contacts scaled by 5 (|d| + 0.005).
"""

from __future__ import annotations

import os

import torch

from . import ops


def get_result_path(data_root_path, object_code, hand_name, n_contact, energy_name, grasp_type=None):
    """fit.py:203-221"""
    path = os.path.join(data_root_path, object_code, "grasp_predictions", hand_name, f"{n_contact}_contacts", energy_name,
                        "default" if grasp_type in (None, "all") else grasp_type)
    os.makedirs(path, exist_ok=True)
    return path


def grasp_snapshot(hand_model, energy, object_model):
    """The tensors of fit.py:226-252 for the whole batch (device tensors): root_pose (B,7), joint positions (B,J) and the
    three (B,J) velocity sets.  The hand model's contact indices are left as they were."""
    old_idx = hand_model.contact_point_indices.clone()
    with torch.no_grad():
        distance, normal = object_model.cal_distance(hand_model.contact_points)
        d = 5 * (normal * distance.unsqueeze(-1).abs())
        delta, _, _ = hand_model.get_req_joint_velocities(-d, hand_model.contact_point_indices, return_ee_vel=True)
        hand_model._set_contact_idxs("all")
        distance_f, normal_f = object_model.cal_distance(hand_model.contact_points)
        d_f = 5 * (normal_f * distance_f.unsqueeze(-1).abs())
        delta_full, _ = hand_model.get_req_joint_velocities(-d_f, hand_model.contact_point_indices)
        hand_model._set_contact_idxs(old_idx)
        distance, normal = object_model.cal_distance(hand_model.contact_points)
        d_o = 5 * normal * (distance.unsqueeze(-1).abs() + 0.005)
        delta_off, _, _ = hand_model.get_req_joint_velocities(-d_o, hand_model.contact_point_indices, return_ee_vel=True)
        root = ops.root_pose_wxyz(hand_model.hand_pose)
    return {"root_pose": root, "joint_positions": hand_model.hand_pose.detach()[:, 9:], "grasp_velocities": delta,
            "full_grasp_velocities": delta_full, "grasp_velocities_off": delta_off,
            "values": energy.detach(), "contact_idx": hand_model.contact_point_indices.detach()}


def snapshot_dicts(hand_model, energy, object_model, n_objects, batch_size, grasp_type=None):
    """One dict per object in the reference's on-disk layout (CPU tensors)."""
    snap = {k: v.cpu() for k, v in grasp_snapshot(hand_model, energy, object_model).items()}
    names = list(hand_model._actuated_joints_names)
    out = []
    for a in range(n_objects):
        s, e = a * batch_size, (a + 1) * batch_size
        params = {names[i]: snap["joint_positions"][s:e, i].clone() for i in range(len(names))}
        params["root_pose"] = snap["root_pose"][s:e].clone()
        data = {"values": snap["values"][s:e].clone(), "parameters": params}
        for key in ("grasp_velocities", "full_grasp_velocities", "grasp_velocities_off"):
            data[key] = {names[i]: snap[key][s:e, i].clone() for i in range(len(names))}
        data["contact_idx"] = snap["contact_idx"][s:e].clone()
        data["grasp_type"] = grasp_type
        data["contact_links"] = hand_model._contact_links
        out.append(data)
    return out


def export_poses(hand_model, energy, object_model, object_code_list, batch_size, data_root_path, hand_name, n_contact,
                 energy_name, suffix="None", grasp_type=None):
    """fit.py:224-300 -> list of the files written."""
    files = []
    dicts = snapshot_dicts(hand_model, energy, object_model, len(object_code_list), batch_size, grasp_type)
    for code, data in zip(object_code_list, dicts):
        path = os.path.join(get_result_path(data_root_path, code, hand_name, n_contact, energy_name, grasp_type),
                            code + f"{suffix}.dexgrasp.pt")
        torch.save(data, path)
        files.append(path)
    return files
