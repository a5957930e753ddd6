"""Oracle for the (re-)initialisation of grasps (reference ``core/initializations.py:15-193``).  TEST INFRASTRUCTURE.

``initialize_convex_hull`` leans on trimesh (convex hull, ``sample_surface_even``, ``nearest.on_surface``), pytorch3d
(``sample_farthest_points``) and transforms3d (``euler2mat``) -- none of them in the reference tree or installable here,
so those pieces are restated from their documented behaviour (PARITY UNPINNED for them; random sampling anyway):

  * hull                scipy.spatial.ConvexHull (qhull, what trimesh uses too), faces oriented outward
  * surface samples     trimesh.sample.sample_surface: face by area CDF (searchsorted), then origin + l1 e1 + l2 e2 with
                        (l1, l2) reflected into the triangle; the "even" rejection of points closer than
                        sqrt(area / (3 count)) is NOT applied (the farthest-point sampling that follows makes it moot)
  * farthest points     pytorch3d.ops.sample_farthest_points with random_start_point=False: start at index 0, then
                        repeatedly the point with the largest distance to the chosen set (first index on ties)
  * closest point       on a convex hull the closest surface point of x + eps n_face is x itself
  * euler2mat 'rxyz'    intrinsic x-y'-z'' rotation = Rx(ai) Ry(aj) Rz(ak)   (pinned against scipy in the tests)
  * trunc_normal_       torch.nn.init.trunc_normal_: inverse-CDF sampling     (pinned against torch in the tests)

Everything else -- ``look_at``, the translation / rotation assembly, the rot6d layout, the per-joint truncated normal
around ``default_state`` -- is the reference's own arithmetic and IS pinned: tools/make_golden.py executes the reference's
``initialize_convex_hull`` with the pieces above standing in for the absent libraries (fixture ``init_*.npz``).
"""

import math

import numpy as np
import torch


def convex_hull_faces(verts) -> np.ndarray:
    """Outward-oriented triangles (F,3,3) float64 of the convex hull of ``verts`` (N,3); degenerate faces dropped."""
    from scipy.spatial import ConvexHull

    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    hull = ConvexHull(v)
    fv = v[hull.simplices]
    n = np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0])
    flip = (n * hull.equations[:, :3]).sum(1) < 0
    fv[flip] = fv[flip][:, [0, 2, 1]]
    area = 0.5 * np.linalg.norm(np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0]), axis=1)
    return fv[area > 1e-14]


def area_cdf(fv) -> np.ndarray:
    fv = np.asarray(fv, dtype=np.float64)
    area = 0.5 * np.linalg.norm(np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0]), axis=1)
    c = np.cumsum(area)
    return c / c[-1]


def sample_surface(fv, u_face, u_len):
    """trimesh.sample.sample_surface with the random numbers given: u_face (M,), u_len (M,2) -> points (M,3), face (M,)."""
    fv = torch.as_tensor(fv, dtype=torch.float64)
    cdf = torch.as_tensor(area_cdf(fv.numpy()))
    f = torch.searchsorted(cdf, torch.as_tensor(u_face, dtype=torch.float64)).clamp(max=fv.shape[0] - 1)
    l = torch.as_tensor(u_len, dtype=torch.float64).clone()
    over = l.sum(1) > 1.0
    l[over] -= 1.0
    l = l.abs()
    p = fv[f, 0] + (fv[f, 1] - fv[f, 0]) * l[:, :1] + (fv[f, 2] - fv[f, 0]) * l[:, 1:]
    return p, f


def face_normals(fv):
    fv = torch.as_tensor(fv, dtype=torch.float64)
    n = torch.linalg.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=-1)
    return n / torch.linalg.norm(n, dim=-1, keepdim=True)


def farthest_points(points, K):
    """Indices (K,) of pytorch3d.ops.sample_farthest_points(points[None], K=K, random_start_point=False)."""
    p = torch.as_tensor(points)
    sel = torch.zeros(K, dtype=torch.long)
    dist = torch.full((p.shape[0],), float("inf"), dtype=p.dtype)
    cur = 0
    for i in range(K):
        sel[i] = cur
        dist = torch.minimum(dist, ((p - p[cur]) ** 2).sum(-1))
        cur = int(torch.argmax(dist))
    return sel


def look_at(camera_positions, target_positions, forward_vector, up_vector):
    """initializations.py:83-117, restated line by line."""
    base_up = up_vector.to(camera_positions.dtype)
    up = base_up.clone().unsqueeze(0).repeat(camera_positions.shape[0], 1)
    fwd_v = forward_vector.to(camera_positions.dtype)
    forward = camera_positions - target_positions
    forward = forward / torch.norm(forward, dim=1, keepdim=True)
    prod = torch.sum(up * forward, dim=1, keepdim=True)
    up = torch.where(prod.abs() < 0.95, up, torch.tensor([0, 1, 0], dtype=up.dtype))
    right = torch.linalg.cross(up, forward, dim=1)
    right = right / torch.norm(right, dim=1, keepdim=True)
    upv = torch.linalg.cross(forward, right, dim=1)
    orientation = torch.stack([forward, upv, right], dim=-1)
    basis = torch.stack([fwd_v, -torch.linalg.cross(fwd_v, base_up, dim=-1), base_up], dim=-1)
    return orientation @ basis


def euler2mat_rxyz(ai, aj, ak):
    """transforms3d.euler.euler2mat(ai, aj, ak, axes='rxyz') for tensors (B,) -> (B,3,3) = Rx(ai) Ry(aj) Rz(ak)."""
    ci, si, cj, sj, ck, sk = torch.cos(ai), torch.sin(ai), torch.cos(aj), torch.sin(aj), torch.cos(ak), torch.sin(ak)
    o, z = torch.ones_like(ai), torch.zeros_like(ai)
    Rx = torch.stack([o, z, z, z, ci, -si, z, si, ci], -1).view(-1, 3, 3)
    Ry = torch.stack([cj, z, sj, z, o, z, -sj, z, cj], -1).view(-1, 3, 3)
    Rz = torch.stack([ck, -sk, z, sk, ck, z, z, z, o], -1).view(-1, 3, 3)
    return Rx @ Ry @ Rz


def trunc_normal_from_uniform(u, mean, std, a, b):
    """torch.nn.init.trunc_normal_ (torch/nn/init.py::_no_grad_trunc_normal_) with its uniform draw u in [0,1) given:
    v = 2l - 1 + u (2u_ - 2l); x = erfinv(v) * std * sqrt(2) + mean; clamp to [a, b]."""
    def ncdf(x):
        return (1.0 + torch.erf(x / math.sqrt(2.0))) / 2.0

    l, up = ncdf((a - mean) / std), ncdf((b - mean) / std)
    v = (2 * l - 1) + u * ((2 * up - 1) - (2 * l - 1))
    x = torch.erfinv(v) * std * math.sqrt(2.0) + mean
    return torch.maximum(torch.minimum(x, b), a)


DEFAULT_ARGS = dict(jitter_strength=0.1, distance_lower=0.05, distance_upper=0.1, rotate_lower=-math.pi, rotate_upper=math.pi,
                    pitch_lower=-15 * math.pi / 180, pitch_upper=15 * math.pi / 180, tilt_lower=-45 * math.pi / 180,
                    tilt_upper=45 * math.pi / 180)  # scripts/fit.py:59-71


def poses_from_samples(spec, p, n, u_pose, u_joint, args=None, dtype=torch.float64, joints=None):
    """initializations.py:79-186 from the point where p (B,3) = inflated hull samples and n (B,3) = unit direction from p
    to the hull are known.  u_pose (B,4) uniform draws for distance / rotate / pitch / tilt, u_joint (B,J) for the joints.
    -> hand_pose (B, 9 + J)."""
    a = dict(DEFAULT_ARGS)
    a.update(args or {})
    p, n = p.to(dtype), n.to(dtype)
    axes = {"x": [1.0, 0, 0], "y": [0, 1.0, 0], "z": [0, 0, 1.0]}
    fwd = torch.tensor(spec.forward_axis if not isinstance(spec.forward_axis, str) else axes[spec.forward_axis], dtype=dtype)
    upv = torch.tensor(spec.up_axis if not isinstance(spec.up_axis, str) else axes[spec.up_axis], dtype=dtype)
    Rg = look_at(p, p + n, fwd, upv)
    u = u_pose.to(dtype)
    distance = a["distance_lower"] + (a["distance_upper"] - a["distance_lower"]) * u[:, 0]
    rotate = a["rotate_lower"] + (a["rotate_upper"] - a["rotate_lower"]) * u[:, 1]
    pitch = a["pitch_lower"] + (a["pitch_upper"] - a["pitch_lower"]) * u[:, 2]
    tilt = a["tilt_lower"] + (a["tilt_upper"] - a["tilt_lower"]) * u[:, 3]
    Rl = euler2mat_rxyz(tilt, pitch, rotate)
    t = p - distance.unsqueeze(1) * n
    R = Rg @ Rl
    lo, hi = torch.as_tensor(spec.joints_lower, dtype=dtype), torch.as_tensor(spec.joints_upper, dtype=dtype)
    mu = torch.minimum(torch.maximum(torch.as_tensor(spec.default_state, dtype=dtype), lo), hi)
    sigma = a["jitter_strength"] * (hi - lo)
    if joints is not None:  # joint angles given (fixture of the reference's own trunc_normal_ draws)
        th = joints.to(dtype)
    else:
        th = trunc_normal_from_uniform(u_joint.to(dtype), mu[None], sigma[None], (lo - 1e-6)[None], (hi + 1e-6)[None])
    return torch.cat([t, R.transpose(1, 2)[:, :2].reshape(-1, 6), th], dim=1)


def initialize_convex_hull(spec, hull_fvs, batch_each, draws, args=None, samples_per_object=None, dtype=torch.float64,
                           joints=None):
    """Whole pipeline per object: sample M = 100 * batch_each points on the hull, inflate by 1 cm along the face normal,
    farthest-point-sample batch_each of them, n = -(face normal), then ``poses_from_samples``.
    draws: u_face (n_obj,M), u_len (n_obj,M,2), u_pose (B,4), u_joint (B,J).  -> hand_pose (B, 9+J), p (B,3), n (B,3)."""
    M = samples_per_object or 100 * batch_each
    ps, ns = [], []
    for i, fv in enumerate(hull_fvs):
        pts, f = sample_surface(fv, draws["u_face"][i][:M], draws["u_len"][i][:M])
        nrm = face_normals(fv)[f]
        pts = pts + 0.01 * nrm
        sel = farthest_points(pts, batch_each)
        ps.append(pts[sel])
        ns.append(-nrm[sel])
    p, n = torch.cat(ps), torch.cat(ns)
    return poses_from_samples(spec, p, n, draws["u_pose"], draws.get("u_joint"), args, dtype, joints), p, n
