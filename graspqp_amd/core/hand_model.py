"""HandModel with the reference's attribute/method surface (core/hand_model.py) on top of the HIP kernels.

Only the hot-path part of the reference class is mirrored: ``set_parameters``, ``fk``,
``_set_contact_idxs``, ``cal_distance``, ``self_penetration`` and the state attributes the optimizer and
``calculate_energy`` touch.  Construction goes through a :class:`HandSpec` (flat arrays) instead of
pytorch_kinematics / trimesh objects.
"""

from __future__ import annotations

import torch

from .. import ops
from ..hands import get_hand_spec


class HandModel:
    def __init__(self, spec, device="cuda", grasp_type=None, n_surface_points=512):
        if not str(device).startswith("cuda"):
            raise RuntimeError("graspqp_amd.HandModel runs on the GPU only (device='cuda'); there is no CPU path")
        self.spec = spec
        self.device = torch.device(device)
        self._hand = ops.HandHandle(spec)
        self.n_dofs = spec.n_dofs
        self.n_contact_candidates = spec.n_contact_candidates
        self._actuated_joints_names = list(spec.joint_names)
        self.joints_names = list(spec.joint_names)
        # reference hand_model.py:451: the grasp_type's link subset (a spec from get_hand_spec(..., grasp_type=...) carries
        # it), None for the default type
        self._contact_links = getattr(spec, "contact_links", None)
        self.grasp_type = getattr(spec, "grasp_type", grasp_type)
        self.joints_lower = torch.tensor(spec.joints_lower, device=self.device)
        self.joints_upper = torch.tensor(spec.joints_upper, device=self.device)
        self.default_state = torch.tensor(spec.default_state, device=self.device)
        self.forward_axis = torch.tensor(spec.forward_axis, device=self.device)
        self.up_axis = torch.tensor(spec.up_axis, device=self.device)
        self.grasp_axis = torch.tensor(spec.grasp_axis, device=self.device)
        self.global_index_to_link_index = torch.tensor(spec.cand_link, dtype=torch.long, device=self.device)
        self.hand_pose = None
        self.contact_point_indices = None
        self.global_translation = None
        self.global_rotation = None
        self.current_status = None  # (B,L,3,4) link transforms in the hand frame
        self.contact_points = None
        self.contact_normals = None
        self._sphere_centers = None
        self._fk_ws = None
        self._n_surface_points = int(n_surface_points)
        self._surface = None  # (handle whose "candidates" are the hand's surface samples, link-frame points, link ids)

    # reference hand_model.py:762-766 -- returns the (B,L,3,4) link transforms (the reference returns a dict of
    # Transform3d keyed by link name; ``link_matrix(name)`` gives the same 4x4 view).
    def fk(self, joint_angles):
        B = joint_angles.shape[0]
        hp = torch.zeros(B, 9 + self.n_dofs, device=self.device)
        hp[:, 3] = 1.0
        hp[:, 7] = 1.0
        hp[:, 9:] = joint_angles.detach()
        idx = torch.zeros(B, 0, dtype=torch.long, device=self.device)
        return ops.fk_contacts(hp, idx, self._hand)[1]

    def link_matrix(self, link_name):
        l = self.spec.link_names.index(link_name)
        T = self.current_status[:, l]
        bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], device=self.device).expand(T.shape[0], 1, 4)
        return torch.cat([T, bottom], dim=1)

    # reference hand_model.py:833-873
    def set_parameters(self, hand_pose, contact_point_indices=None, env_mask=None, _known_finite=False):
        """``_known_finite`` (internal, MalaStar.try_step): the pose comes straight from the proposal kernel, which zeroes every
        row that contains a NaN (optimizer.py:242-244) -- the reference's ``isnan().any()`` check (hand_model.py:862-863) cannot
        fire on it, and evaluating it would put a host-device synchronisation into every iteration."""
        if env_mask is not None:
            with torch.no_grad():
                self.hand_pose = torch.where(env_mask.unsqueeze(-1), hand_pose, self.hand_pose)
            self.hand_pose.requires_grad = True
            self.hand_pose.retain_grad()
        else:
            self.hand_pose = hand_pose.clone()
        if self.hand_pose.requires_grad:
            self.hand_pose.retain_grad()
        if not _known_finite and self.hand_pose.isnan().any():
            raise ValueError("nan in hand_pose")
        self.global_translation = self.hand_pose[:, 0:3]
        self._set_contact_idxs(contact_point_indices, env_mask=env_mask)

    # reference hand_model.py:787-831 -- only the n selected candidates are transformed (the reference
    # transforms all C and gathers n)
    def _set_contact_idxs(self, contact_point_indices, env_mask=None):
        if contact_point_indices is None:
            contact_point_indices = self.contact_point_indices
        if isinstance(contact_point_indices, str) and contact_point_indices == "all":
            contact_point_indices = (
                torch.arange(self.n_contact_candidates, dtype=torch.long, device=self.device)
                .unsqueeze(0)
                .expand(self.hand_pose.shape[0], -1)
            )
        if env_mask is None or self.contact_point_indices is None:
            self.contact_point_indices = contact_point_indices.clone()
        else:
            self.contact_point_indices = torch.where(env_mask.unsqueeze(-1), contact_point_indices, self.contact_point_indices)
        # reference hand_model.py:815-831: the contact points are gathered with the indices PASSED IN, also for the rows
        # outside env_mask (whose stored indices stay what they were) -- kept, it decides the energies of a reset iteration
        self._fk_idx = contact_point_indices.contiguous()  # the indices the kinematic state below belongs to
        Rg, LT, cp, cn, sc, ws = ops.fk_contacts(self.hand_pose, self._fk_idx, self._hand)
        self._fk_ws = ws
        self.global_rotation = Rg
        self.current_status = LT
        self.contact_points = cp
        self.contact_normals = cn
        self._sphere_centers = sc

    # reference hand_model.py:875-987
    def cal_distance(self, x, penetration_only=False):
        """x: (B,N,3) object surface points, identical for the rows of one object (object_model.py:182-184), or the
        un-expanded (n_obj,N,3).  Returns (B,N) max-over-links signed distance, inside positive."""
        B = self.hand_pose.shape[0]
        if x.shape[0] == B and B > 0:
            # recover the per-object tensor: rows of an object share their points
            be = getattr(self, "_batch_each_hint", None)
            if be is None:
                raise RuntimeError("call ObjectModel.attach(hand_model) or pass the (n_obj,N,3) surface tensor")
            surf = x.view(-1, be, x.shape[1], 3)[:, 0].contiguous()
        else:
            surf = x
            be = B // x.shape[0]
        # the kinematic state (Rg, link transforms, FK workspace) is the one written by the last set_parameters
        return ops.hand_pen(self.hand_pose, surf, be, self._hand, self.contact_point_indices,
                            self.global_rotation.detach(), self.current_status.detach(), self._fk_ws,
                            self._fk_ws.numel(), penetration_only)

    # reference hand_model.py:989-1040
    def self_penetration(self):
        return ops.self_pen(self._sphere_centers, self._hand)

    # reference hand_model.py:1220-1267: all C candidates (and normals) in the world frame
    def get_contact_candidates(self, with_normals=False):
        B = self.hand_pose.shape[0]
        all_idx = torch.arange(self.n_contact_candidates, dtype=torch.long, device=self.device).unsqueeze(0).expand(B, -1)
        _, _, cp, cn, _, _ = ops.fk_contacts(self.hand_pose.detach(), all_idx.contiguous(), self._hand)
        return (cp, cn) if with_normals else cp

    # reference hand_model.py:604-629 + 1042-1071: n_surface_points samples of the hand surface (per link proportional to
    # its area, 100x oversampled + farthest-point sampling, seed 42) in the world frame.  The reference draws them with
    # pytorch3d (absent: the sample SET differs, PARITY UNPINNED); here they are produced once on the host and pushed
    # through the FK kernels as extra "contact candidates", which also gives the analytic backward.
    def _surface_handle(self):
        if self._surface is None:
            import numpy as np

            from ..utils import meshes as mesh_utils

            spec = self.spec
            fvs = [spec.link_faces(l).astype(np.float64) for l in range(spec.n_links)]
            areas = [0.5 * np.linalg.norm(np.cross(f[:, 1] - f[:, 0], f[:, 2] - f[:, 0]), axis=1).sum() if len(f) else 0.0 for f in fvs]
            tot = sum(areas)
            counts = [int(a / tot * self._n_surface_points) for a in areas]
            counts[0] += self._n_surface_points - sum(counts)
            pts, lnk = [], []
            for l, (f, k) in enumerate(zip(fvs, counts)):
                if k == 0 or len(f) == 0:
                    continue
                dense = mesh_utils.sample_surface(f, 100 * k, seed=42)
                pts.append(mesh_utils.farthest_point_sampling(dense, k))
                lnk.append(np.full(k, l, dtype=np.int32))
            pts, lnk = np.concatenate(pts).astype(np.float32), np.concatenate(lnk)
            import copy

            s2 = copy.copy(spec)
            s2.cand_pos, s2.cand_nrm, s2.cand_link = pts, np.tile(np.array([[0, 0, 1.0]], dtype=np.float32), (len(pts), 1)), lnk
            self._surface = (ops.HandHandle(s2), pts, lnk)
        return self._surface

    def set_surface_points(self, points, link_ids):
        """Use the given link-frame samples (Ns,3) / link ids (Ns) instead of drawing them (tests, reproducible runs)."""
        import copy

        import numpy as np

        s2 = copy.copy(self.spec)
        pts = np.ascontiguousarray(points, dtype=np.float32)
        s2.cand_pos, s2.cand_link = pts, np.ascontiguousarray(link_ids, dtype=np.int32)
        s2.cand_nrm = np.tile(np.array([[0, 0, 1.0]], dtype=np.float32), (len(pts), 1))
        self._surface = (ops.HandHandle(s2), pts, s2.cand_link)

    def get_surface_points(self):
        """(B, n_surface_points, 3) hand surface samples in the world frame, differentiable w.r.t. hand_pose."""
        h, pts, _ = self._surface_handle()
        B = self.hand_pose.shape[0]
        idx = torch.arange(len(pts), dtype=torch.long, device=self.device).unsqueeze(0).expand(B, -1).contiguous()
        return ops.fk_contacts(self.hand_pose, idx, h)[2]

    # reference hand_model.py:1073-1077
    def get_manipulability(self, moving_directions, contact_point_indices=None, coupled=True):
        _, residuals = self.get_req_joint_velocities(moving_directions, contact_point_indices, coupled=coupled)
        if not coupled:
            residuals = residuals.mean(-1)
        return residuals

    # reference hand_model.py:772-777 (HandModel.jacobian -> Chain.jacobian of the pytorch_kinematics fork): geometric
    # Jacobian [J_v; J_w] of every mesh link, hand base frame, at the link-frame origin
    def jacobian(self, joint_angles):
        B = joint_angles.shape[0]
        hp = torch.zeros(B, 9 + self.n_dofs, device=self.device)
        hp[:, 3] = 1.0
        hp[:, 7] = 1.0
        hp[:, 9:] = joint_angles.detach()
        idx = torch.zeros(B, 0, dtype=torch.long, device=self.device)
        _, LT, _, _, _, ws = ops.fk_contacts(hp, idx, self._hand)
        return ops.link_jacobian(self._hand, LT, ws)

    # reference hand_model.py:1155-1218
    def get_req_joint_velocities(self, moving_directions, contact_point_indices=None, coupled=True, return_ee_vel=False):
        """Joint velocities that move the contact points along ``moving_directions`` (B,n,3, world frame): theta =
        pinv(J) d with the linear contact Jacobian J_v + J_w x r and the damped pseudo-inverse (lambda = 1e-3).
        -> (theta, residuals[, ee_vel]); coupled=False solves every contact on its own ((B,n,J), (B,n,3))."""
        B = self.hand_pose.shape[0]
        if contact_point_indices is None:
            contact_point_indices = (torch.arange(self.n_contact_candidates, dtype=torch.long, device=self.device)
                                     .unsqueeze(0).expand(B, -1))
        idx = contact_point_indices.contiguous()
        n = idx.shape[1]
        if coupled and not return_ee_vel and torch.is_grad_enabled() and (
                self.hand_pose.requires_grad or moving_directions.requires_grad):
            # differentiable route (E_manipulativity, core/energy.py:80-87): gradient to the joint angles through the contact
            # Jacobian, to the root rotation through R' d, and to the directions themselves
            return ops.joint_velocity_residuals(self.hand_pose, moving_directions.to(torch.float32), self._hand, idx,
                                                self.global_rotation, self.current_status, self._fk_ws)
        # link transforms / joint frames of the CURRENT pose (written by the last set_parameters)
        jc = ops.contact_jacobian(self._hand, idx, self.current_status, self._fk_ws)  # (B,n,3,J)
        R = self.global_rotation.detach()
        d = moving_directions.detach().to(torch.float32)
        if coupled:
            theta, res, ee = ops.joint_velocities(jc.reshape(B, 3 * n, self.n_dofs), d.reshape(B, 3 * n), R)
            ee = ee.view(B, n, 3)
        else:
            theta, res, ee = ops.joint_velocities(jc.reshape(B * n, 3, self.n_dofs), d.reshape(B * n, 3),
                                                  R.unsqueeze(1).expand(-1, n, -1, -1).reshape(B * n, 3, 3))
            theta, res, ee = theta.view(B, n, self.n_dofs), res.view(B, n, 3), ee.view(B, n, 3)
        if return_ee_vel:
            return theta, res, ee
        return theta, res

    @property
    def actuated_joints_names(self):
        return self._actuated_joints_names

    @property
    def n_actutated_joints(self):  # (sic) reference hand_model.py:779-781
        return self.n_dofs


def get_hand_model(hand_name: str, device="cuda", asset_dir=None, grasp_type=None, **kwargs) -> HandModel:
    """reference hands/__init__.py:27-28 (``grasp_type`` as in scripts/fit.py:304)"""
    return HandModel(get_hand_spec(hand_name, asset_dir, grasp_type=grasp_type), device=device, grasp_type=grasp_type, **kwargs)
