"""Oracle kinematics: rot6d -> R, URDF tree FK, contact candidates, penetration spheres.

TEST INFRASTRUCTURE (see package docstring).  Everything is differentiable torch so that autograd
gives the reference's gradients (the reference back-propagates through pytorch_kinematics FK).
"""

import torch


def special_gramschmidt(six: torch.Tensor) -> torch.Tensor:
    """rot6d (B,6) = first two COLUMNS of R -> R (B,3,3).

    Reference: ``utils/transforms.py:5-13`` -> ``roma.special_gramschmidt(stack([x_raw, y_raw], -1))``
    (roma 1.5.4, epsilon=0): x = m0/|m0|; y = m1 - (x.m1)x; y /= |y|; z = x cross y; R = [x y z].
    """
    x = six[:, 0:3]
    y = six[:, 3:6]
    x = x / torch.linalg.norm(x, dim=-1, keepdim=True)
    y = y - (x * y).sum(-1, keepdim=True) * x
    y = y / torch.linalg.norm(y, dim=-1, keepdim=True)
    z = torch.linalg.cross(x, y, dim=-1)
    return torch.stack([x, y, z], dim=-1)


def _axis_angle(axis: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """Rodrigues: unit axis (3,), theta (B,) -> (B,3,3)."""
    ax, ay, az = axis[0], axis[1], axis[2]
    zero = torch.zeros((), dtype=theta.dtype)
    K = torch.stack(
        [torch.stack([zero, -az, ay]), torch.stack([az, zero, -ax]), torch.stack([-ay, ax, zero])]
    ).to(theta.dtype)
    s = torch.sin(theta)[:, None, None]
    c = torch.cos(theta)[:, None, None]
    eye = torch.eye(3, dtype=theta.dtype)
    return eye + s * K + (1 - c) * (K @ K)


def full_joint_angles(spec, theta: torch.Tensor) -> torch.Tensor:
    """Actuated joint values (B,J) -> values of all N moving joints of the tree.  Coupled hands restate the
    ``joint_calc_fnc`` of reference hands/ability_hand.py:9-30 (q2 = 1.05851325 q1) and hands/panda.py:6-14 (both fingers
    follow one value): theta_tree = C theta + c0; identity for the other hands."""
    C = torch.as_tensor(spec.coupling, dtype=theta.dtype)
    if C.shape[0] == C.shape[1] and torch.equal(C, torch.eye(C.shape[0], dtype=theta.dtype)):
        return theta
    return theta @ C.T + torch.as_tensor(spec.coupling_offset, dtype=theta.dtype)


def forward_kinematics(spec, theta: torch.Tensor) -> torch.Tensor:
    """Joint angles (B,J) -> mesh-link transforms (B,L,4,4) in the hand base frame.

    Restates ``pytorch_kinematics.Chain.forward_kinematics`` as used at reference
    ``hand_model.py:762-766``: world(frame) = world(parent) . joint_origin . joint_motion(theta);
    revolute = rotation about the axis, prismatic = translation along it.  Visual/collision origins
    are NOT part of the link transform (they are baked into the geometry, spec builder).
    """
    B = theta.shape[0]
    dt = theta.dtype
    theta = full_joint_angles(spec, theta)
    F = len(spec.frame_names)
    origin = torch.as_tensor(spec.frame_origin, dtype=dt)
    axis = torch.as_tensor(spec.frame_axis, dtype=dt)
    world = [None] * F
    for f in range(F):
        p = int(spec.frame_parent[f])
        Wp = torch.eye(4, dtype=dt).expand(B, 4, 4) if p < 0 else world[p]
        W = Wp @ origin[f]
        jt = int(spec.frame_joint_type[f])
        if jt != 0:
            q = theta[:, int(spec.frame_dof[f])]
            M = torch.eye(4, dtype=dt).repeat(B, 1, 1)
            if jt == 1:
                M[:, :3, :3] = _axis_angle(axis[f], q)
            else:
                M[:, :3, 3] = axis[f][None, :] * q[:, None]
            W = W @ M
        world[f] = W
    return torch.stack([world[int(f)] for f in spec.link_frame], dim=1)


def contact_candidates_world(spec, link_T, R, t):
    """All C candidates + normals in the world frame, (B,C,3) each.

    Reference ``hand_model.py:1220-1267``: p_w = R (T_l c) + t ; n_w = R R_l n_c.
    """
    dt = link_T.dtype
    cl = torch.as_tensor(spec.cand_link, dtype=torch.long)
    c = torch.as_tensor(spec.cand_pos, dtype=dt)
    n = torch.as_tensor(spec.cand_nrm, dtype=dt)
    Rl = link_T[:, cl, :3, :3]  # (B,C,3,3)
    tl = link_T[:, cl, :3, 3]
    ph = (Rl @ c[None, :, :, None]).squeeze(-1) + tl
    nh = (Rl @ n[None, :, :, None]).squeeze(-1)
    pw = ph @ R.transpose(1, 2) + t[:, None, :]
    nw = nh @ R.transpose(1, 2)
    return pw, nw


def sphere_centers_world(spec, link_T, R, t):
    """Penetration sphere centres in the world frame (B,S,3) (reference hand_model.py:1013-1021)."""
    dt = link_T.dtype
    sl = torch.as_tensor(spec.sphere_link, dtype=torch.long)
    c = torch.as_tensor(spec.sphere[:, :3], dtype=dt)
    Rl = link_T[:, sl, :3, :3]
    tl = link_T[:, sl, :3, 3]
    ph = (Rl @ c[None, :, :, None]).squeeze(-1) + tl
    return ph @ R.transpose(1, 2) + t[:, None, :]


def self_penetration(spec, centers: torch.Tensor) -> torch.Tensor:
    """E_spen (B,) from world sphere centres (B,S,3).

    Reference ``hand_model.py:989-1040``: for every sphere-carrying link except the last,
    min over (its spheres) x (all spheres of LATER links) of |a-b+1e-13| - r_a - r_b;
    E = -sum_links min(0, .).
    """
    B = centers.shape[0]
    dt = centers.dtype
    sl = spec.sphere_link
    r = torch.as_tensor(spec.sphere[:, 3], dtype=dt)
    S = len(sl)
    if S == 0:
        return torch.zeros(B, dtype=dt)
    # contiguous groups by link (sphere_link is non-decreasing)
    starts = [0] + [i for i in range(1, S) if sl[i] != sl[i - 1]] + [S]
    groups = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
    cols = []
    for a, b in groups[:-1]:
        mine = centers[:, a:b]  # (B,na,3)
        other = centers[:, b:]  # (B,no,3)
        dis = torch.linalg.norm(mine.unsqueeze(1) - other.unsqueeze(2) + 1e-13, dim=-1)  # (B,no,na)
        th = r[a:b].view(1, -1) + r[b:].view(-1, 1)
        pen = dis - th
        cols.append(pen.min(1)[0].min(1)[0])
    if not cols:
        return torch.zeros(B, dtype=dt)
    distances = torch.stack(cols, dim=1)
    return -distances.clamp(max=0).sum(1)
