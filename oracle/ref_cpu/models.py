"""Oracle hand / object models.  TEST INFRASTRUCTURE.

Duck-typed to the attribute surface the reference's ``calculate_energy`` (core/energy.py) and
``MalaStar`` (core/optimizer.py) use, so those reference files can be run on these models in the
build container to produce golden fixtures.  Hot-path methods restate
``HandModel.set_parameters / _set_contact_idxs / cal_distance / self_penetration``
(reference hand_model.py:787-873, 875-987, 989-1040) and ``ObjectModel.cal_distance / cog``
(object_model.py:64-68, 186-255).
"""

import torch

from . import kin, sdf


class OracleHand:
    def __init__(self, spec, dtype=torch.float64):
        self.spec = spec
        self.dtype = dtype
        self.device = "cpu"
        self.n_dofs = spec.n_dofs
        self.n_contact_candidates = spec.n_contact_candidates
        self.joints_lower = torch.as_tensor(spec.joints_lower, dtype=dtype)
        self.joints_upper = torch.as_tensor(spec.joints_upper, dtype=dtype)
        self.default_state = torch.as_tensor(spec.default_state, dtype=dtype)
        self.link_faces = [torch.as_tensor(spec.link_faces(l), dtype=dtype) for l in range(spec.n_links)]
        self.grasp_axis = torch.as_tensor(spec.grasp_axis, dtype=dtype)
        self.surface_points = None  # (Ns,3) link-frame samples + self.surface_link (Ns): set by the test / fixture script
        self.surface_link = None
        self.hand_pose = None
        self.contact_point_indices = None
        self.global_translation = None
        self.global_rotation = None
        self.current_status = None  # (B,L,4,4) link transforms in the hand frame
        self.contact_points = None
        self.contact_normals = None

    # reference hand_model.py:762-766
    def fk(self, joint_angles):
        return kin.forward_kinematics(self.spec, joint_angles)

    # reference hand_model.py:833-873 (env_mask branch restated too)
    def set_parameters(self, hand_pose, contact_point_indices=None, env_mask=None):
        if env_mask is not None:
            with torch.no_grad():
                self.hand_pose = torch.where(env_mask.unsqueeze(-1), hand_pose, self.hand_pose)
            self.hand_pose.requires_grad = True
            self.hand_pose.retain_grad()
        else:
            self.hand_pose = hand_pose.clone()
        if self.hand_pose.requires_grad:
            self.hand_pose.retain_grad()
        self.global_translation = self.hand_pose[:, 0:3]
        if self.hand_pose.isnan().any():
            raise ValueError("nan in hand_pose")
        self.global_rotation = kin.special_gramschmidt(self.hand_pose[:, 3:9])
        self.current_status = self.fk(self.hand_pose[:, 9:])
        self._set_contact_idxs(contact_point_indices, env_mask=env_mask)

    # reference hand_model.py:787-831
    def _set_contact_idxs(self, contact_point_indices, env_mask=None):
        if contact_point_indices is None:
            return
        if isinstance(contact_point_indices, str) and contact_point_indices == "all":
            contact_point_indices = (
                torch.arange(self.n_contact_candidates, dtype=torch.long).unsqueeze(0).expand(self.hand_pose.shape[0], -1)
            )
        if env_mask is None:
            self.contact_point_indices = contact_point_indices.clone()
        else:
            self.contact_point_indices = torch.where(env_mask.unsqueeze(-1), contact_point_indices, self.contact_point_indices)
        pw, nw = kin.contact_candidates_world(self.spec, self.current_status, self.global_rotation, self.global_translation)
        self.all_contact_points, self._all_contact_normals = pw, nw
        idx = contact_point_indices.unsqueeze(-1).expand(-1, -1, 3)
        self.contact_points = pw.gather(1, idx)
        self.contact_normals = nw.gather(1, idx)

    # reference hand_model.py:875-987 (TORCHSDF branch)
    def cal_distance(self, x):
        B, N, _ = x.shape
        xh = (x - self.global_translation.unsqueeze(1)) @ self.global_rotation
        dis = []
        for l, fv in enumerate(self.link_faces):
            if fv.shape[0] == 0:
                continue
            T = self.current_status[:, l]
            xl = (xh - T[:, :3, 3].unsqueeze(1)) @ T[:, :3, :3]
            d2, sgn, _, _ = sdf.compute_sdf(xl.reshape(-1, 3), fv)
            dl = torch.sqrt(d2 + 1e-8) * (-sgn)
            dis.append(dl.reshape(B, N))
        return torch.max(torch.stack(dis, dim=0), dim=0)[0]

    # reference hand_model.py:1042-1071
    def get_surface_points(self):
        T = self.current_status[:, torch.as_tensor(self.surface_link, dtype=torch.long)]
        c = torch.as_tensor(self.surface_points, dtype=self.dtype)
        ph = (T[..., :3, :3] @ c[None, :, :, None]).squeeze(-1) + T[..., :3, 3]
        return ph @ self.global_rotation.transpose(1, 2) + self.global_translation.unsqueeze(1)

    # reference hand_model.py:989-1040
    def self_penetration(self):
        c = kin.sphere_centers_world(self.spec, self.current_status, self.global_rotation, self.global_translation)
        return kin.self_penetration(self.spec, c)


class OracleObject:
    """Object meshes + surface samples; ``scale`` fixed to 1 (reference object_model.py:60)."""

    def __init__(self, face_verts_list, surface_points_list, batch_size_each, dtype=torch.float64):
        self.device = "cpu"
        self.dtype = dtype
        self.batch_size_each = batch_size_each
        self.object_face_verts_list = [torch.as_tensor(f, dtype=dtype) for f in face_verts_list]
        self.object_mesh_list = list(range(len(face_verts_list)))
        sp = torch.stack([torch.as_tensor(s, dtype=dtype) for s in surface_points_list], dim=0)
        self.surface_points_tensor = sp.repeat_interleave(batch_size_each, dim=0)  # object_model.py:182-184
        self.object_scale_tensor = torch.ones(len(face_verts_list), batch_size_each, dtype=dtype)
        self._cog = None

    @property
    def cog(self):  # object_model.py:64-68
        if self._cog is None:
            self._cog = self.surface_points_tensor.mean(dim=1)
        return self._cog

    def cal_distance(self, x):  # object_model.py:186-255 (TORCHSDF branch)
        _, n_points, _ = x.shape
        x = x.reshape(-1, self.batch_size_each * n_points, 3)
        scale = self.object_scale_tensor.repeat_interleave(n_points, dim=1)
        x = x / scale.unsqueeze(2)
        distance, normals = [], []
        for i, fv in enumerate(self.object_face_verts_list):
            d2, sgn, nrm, _ = sdf.compute_sdf(x[i], fv)
            dis = torch.sqrt(d2 + 1e-8) * (-sgn)
            distance.append(dis)
            normals.append(nrm * sgn.unsqueeze(1))
        distance = torch.stack(distance) * scale
        normals = torch.stack(normals)
        return distance.reshape(-1, n_points), normals.reshape(-1, n_points, 3)
