"""Oracle for the export path (reference ``scripts/fit.py:224-300`` ``export_poses``).  TEST INFRASTRUCTURE.

Restates ``pinv`` (``core/hand_model.py:46-54``), ``HandModel.jacobian`` (``:772-777`` -> the pytorch_kinematics
fork's tree ``Chain.jacobian``: the geometric Jacobian [J_v; J_w] of every mesh link in the hand base frame at the
link-frame origin), ``HandModel.get_req_joint_velocities`` (``:1155-1218``) and the rot6d -> (w,x,y,z) conversion of
``fit.py:260-263`` (``roma.rotmat_to_unitquat``, which follows scipy's ``Rotation.from_matrix``).

PARITY UNPINNED for the two third-party pieces (pytorch_kinematics fork, roma: not in the reference tree, not
installable).  What the tests pin instead: the analytic Jacobian against autograd through the oracle FK (an
independent derivation), the quaternion against ``scipy.spatial.transform.Rotation`` (installed), and the dict layout
against the keys the reference's own consumer reads (``graspqp_isaaclab/.../utils/data.py:105-140``).
"""

import torch

from . import kin


def pinv(A: torch.Tensor, l: float = 1e-3) -> torch.Tensor:
    """hand_model.py:46-54: damped right inverse for m < n, damped left inverse otherwise."""
    m, n = A.shape[-2:]
    if m < n:
        return A.mT @ torch.linalg.inv(A @ A.mT + l * torch.eye(m, dtype=A.dtype))
    return torch.linalg.inv(A.mT @ A + l * torch.eye(n, dtype=A.dtype)) @ A.mT


def _frame_worlds(spec, theta):
    """World (hand-base) transform of every URDF frame, list of (B,4,4) (same walk as kin.forward_kinematics)."""
    B, dt = theta.shape[0], theta.dtype
    theta = kin.full_joint_angles(spec, theta)
    origin = torch.as_tensor(spec.frame_origin, dtype=dt)
    axis = torch.as_tensor(spec.frame_axis, dtype=dt)
    world = []
    for f in range(len(spec.frame_names)):
        p = int(spec.frame_parent[f])
        W = (torch.eye(4, dtype=dt).expand(B, 4, 4) if p < 0 else world[p]) @ origin[f]
        jt = int(spec.frame_joint_type[f])
        if jt != 0:
            q = theta[:, int(spec.frame_dof[f])]
            M = torch.eye(4, dtype=dt).repeat(B, 1, 1)
            if jt == 1:
                M[:, :3, :3] = kin._axis_angle(axis[f], q)
            else:
                M[:, :3, 3] = axis[f][None, :] * q[:, None]
            W = W @ M
        world.append(W)
    return world


def link_jacobian(spec, theta: torch.Tensor) -> torch.Tensor:
    """(B,L,6,J): column j of link l = [a_j x (o_l - p_j); a_j] for a revolute joint j on the path base -> l,
    [a_j; 0] for a prismatic one, 0 for joints off the path (textbook geometric Jacobian, base frame)."""
    B, dt = theta.shape[0], theta.dtype
    world = _frame_worlds(spec, theta)
    axis = torch.as_tensor(spec.frame_axis, dtype=dt)
    J = spec.n_nodes
    out = torch.zeros(B, len(spec.link_frame), 6, J, dtype=dt)
    for li, lf in enumerate(spec.link_frame):
        o = world[int(lf)][:, :3, 3]
        f = int(lf)
        while f >= 0:
            jt = int(spec.frame_joint_type[f])
            if jt != 0:
                j = int(spec.frame_dof[f])
                a = world[f][:, :3, :3] @ axis[f]
                if jt == 1:
                    out[:, li, :3, j] = torch.linalg.cross(a, o - world[f][:, :3, 3], dim=-1)
                    out[:, li, 3:, j] = a
                else:
                    out[:, li, :3, j] = a
            f = int(spec.frame_parent[f])
    # coupled hands: the reference's jacobian_fnc folds the tree-joint columns into the actuated ones (ability_hand.py:
    # 33-40: active = J[..., [0,2,4,6,8,9]] + mult * J[..., [1,3,5,7]]; panda.py:17-26: J[...,0] + J[...,1]) = J_tree C
    return out @ torch.as_tensor(spec.coupling, dtype=dt)


def contact_points_hand_frame(spec, theta, idx):
    """Selected contact candidates in the hand base frame, (B,n,3) -- differentiable w.r.t. theta."""
    LT = kin.forward_kinematics(spec, theta)
    cl = torch.as_tensor(spec.cand_link, dtype=torch.long)[idx]
    c = torch.as_tensor(spec.cand_pos, dtype=theta.dtype)[idx]
    T = LT[torch.arange(theta.shape[0])[:, None], cl]
    return (T[..., :3, :3] @ c.unsqueeze(-1)).squeeze(-1) + T[..., :3, 3]


def contact_jacobian(spec, theta, idx):
    """hand_model.py:1176-1196: J_v + J_w x r with r = R_l c (candidate offset from the link origin), (B,n,3,J)."""
    B, n = idx.shape
    Jl = link_jacobian(spec, theta)
    LT = kin.forward_kinematics(spec, theta)
    cl = torch.as_tensor(spec.cand_link, dtype=torch.long)[idx]  # (B,n)
    c = torch.as_tensor(spec.cand_pos, dtype=theta.dtype)[idx]
    rows = torch.arange(B)[:, None]
    r = (LT[rows, cl][..., :3, :3] @ c.unsqueeze(-1)).squeeze(-1)  # (B,n,3)
    Jc = Jl[rows, cl]  # (B,n,6,J)
    return Jc[..., :3, :] + torch.linalg.cross(Jc[..., 3:, :], r.unsqueeze(-1).expand(-1, -1, -1, Jc.shape[-1]), dim=-2)


def get_req_joint_velocities(hand, moving_directions, contact_point_indices=None, coupled=True, return_ee_vel=False):
    """hand_model.py:1155-1218 on an OracleHand (state set by set_parameters)."""
    spec = hand.spec
    R = hand.global_rotation
    md = (R.mT.unsqueeze(1) @ moving_directions.unsqueeze(-1)).squeeze(-1)
    B = hand.hand_pose.shape[0]
    if contact_point_indices is None:
        contact_point_indices = torch.arange(spec.n_contact_candidates).unsqueeze(0).expand(B, -1)
    j = contact_jacobian(spec, hand.hand_pose[:, 9:], contact_point_indices)
    n = j.shape[1]
    if coupled:
        j = j.flatten(1, 2)
        md = md.flatten(1, 2)
    md = md.unsqueeze(-1)
    theta = pinv(j) @ md
    ee = j @ theta
    resid = (ee - md) ** 2
    if return_ee_vel:
        if coupled:
            ee = ee.view(-1, n, 3)
        ee = (R.unsqueeze(1) @ ee.unsqueeze(-1)).squeeze(-1)
        return theta.squeeze(-1), resid.squeeze(-1), ee
    return theta.squeeze(-1), resid.squeeze(-1)


def rotmat_to_unitquat_xyzw(R: torch.Tensor) -> torch.Tensor:
    """roma.rotmat_to_unitquat (= scipy Rotation.from_matrix): largest of (R00, R11, R22, trace) picks the branch;
    normalised, not sign-canonicalised."""
    B = R.shape[0]
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    dec = torch.stack([R[:, 0, 0], R[:, 1, 1], R[:, 2, 2], tr], dim=1)
    ch = dec.argmax(dim=1)
    q = torch.zeros(B, 4, dtype=R.dtype)
    for b in range(B):
        c = int(ch[b])
        if c != 3:
            i, j, k = c, (c + 1) % 3, (c + 2) % 3
            q[b, i] = 1 - tr[b] + 2 * R[b, i, i]
            q[b, j] = R[b, j, i] + R[b, i, j]
            q[b, k] = R[b, k, i] + R[b, i, k]
            q[b, 3] = R[b, k, j] - R[b, j, k]
        else:
            q[b, 0] = R[b, 2, 1] - R[b, 1, 2]
            q[b, 1] = R[b, 0, 2] - R[b, 2, 0]
            q[b, 2] = R[b, 1, 0] - R[b, 0, 1]
            q[b, 3] = 1 + tr[b]
    return q / torch.linalg.norm(q, dim=1, keepdim=True)


def export_poses(hand, obj, energy, object_codes, batch_size, grasp_type=None, contact_links=None):
    """fit.py:224-300 on oracle models -> list of the dicts the reference torch.save()s, one per object."""
    full = hand.hand_pose.detach()
    old_idx = hand.contact_point_indices.clone()
    distance, normal = obj.cal_distance(hand.contact_points)
    normal = 5 * (normal * distance.unsqueeze(-1).abs())
    d_theta, _, _ = get_req_joint_velocities(hand, -normal, hand.contact_point_indices, return_ee_vel=True)
    hand._set_contact_idxs("all")
    dist_f, normal_f = obj.cal_distance(hand.contact_points)
    normal_f = 5 * (normal_f * dist_f.unsqueeze(-1).abs())
    d_theta_full, _ = get_req_joint_velocities(hand, -normal_f, hand.contact_point_indices)
    hand._set_contact_idxs(old_idx)
    distance, normal = obj.cal_distance(hand.contact_points)
    normal = 5 * normal * (distance.unsqueeze(-1).abs() + 0.005)
    d_theta_off, _, _ = get_req_joint_velocities(hand, -normal, hand.contact_point_indices, return_ee_vel=True)
    names = list(hand.spec.joint_names)
    out = []
    for a in range(len(object_codes)):
        s, e = a * batch_size, (a + 1) * batch_size
        Rm = kin.special_gramschmidt(full[s:e, 3:9])
        q = rotmat_to_unitquat_xyzw(Rm)[:, [3, 0, 1, 2]]
        params = {names[i]: full[s:e, 9 + i] for i in range(len(names))}
        params["root_pose"] = torch.cat([full[s:e, :3], q], dim=1)
        out.append({
            "values": energy.detach()[s:e],
            "parameters": params,
            "grasp_velocities": {names[i]: d_theta[s:e, i].detach() for i in range(len(names))},
            "full_grasp_velocities": {names[i]: d_theta_full[s:e, i].detach() for i in range(len(names))},
            "grasp_velocities_off": {names[i]: d_theta_off[s:e, i].detach() for i in range(len(names))},
            "contact_idx": hand.contact_point_indices[s:e].detach(),
            "grasp_type": grasp_type,
            "contact_links": contact_links,
        })
    return out
