"""ObjectModel with the reference's surface (core/object_model.py) on top of the HIP SDF kernels.

Meshes come as triangle soups (the reference loads them with trimesh from a dataset directory,
object_model.py:97-136; here ``initialize_from_meshes`` takes arrays, ``initialize`` reads OBJ files).
"""

from __future__ import annotations

import os

import numpy as np
import torch

from .. import ops
from ..hands.spec import load_obj_triangles
from ..utils import meshes as mesh_utils


class ObjectModel:
    def __init__(self, data_root_path=None, batch_size_each=1, scale=1.0, num_samples=2000, device="cuda"):
        if not str(device).startswith("cuda"):
            raise RuntimeError("graspqp_amd.ObjectModel runs on the GPU only (device='cuda')")
        self.device = torch.device(device)
        self.batch_size_each = batch_size_each
        self.data_root_path = data_root_path
        self.num_samples = num_samples
        self.scale = scale
        self.object_code_list = None
        self.object_scale_tensor = None
        self.object_mesh_list = None
        self.object_face_verts_list = None
        self.surface_points_tensor = None
        self.surface_points_each = None  # (n_obj, P, 3), un-expanded
        self._meshset = None
        self._cog = None
        self._hull = None  # (face_verts (sumF,3,3), area cdf (sumF), offsets (n_obj+1)) on the device, built on demand
        self.sdf_library = "HIP"

    @property
    def cog(self):  # object_model.py:64-68
        if self._cog is None:
            self._cog = self.surface_points_tensor.mean(dim=1)
        return self._cog

    def initialize(self, object_code_list, extension=".obj", **_):
        """object_model.py:70-184 for OBJ files under ``<root>/<code>/coacd/{remeshed,decomposed}.obj`` or ``<root>/<code>/*.obj``."""
        if not isinstance(object_code_list, list):
            object_code_list = [object_code_list]
        fvs = []
        for code in object_code_list:
            cand = [os.path.join(self.data_root_path, code, "coacd", "remeshed.obj"),
                    os.path.join(self.data_root_path, code, "coacd", "decomposed.obj")]
            cand += sorted(
                os.path.join(self.data_root_path, code, f)
                for f in (os.listdir(os.path.join(self.data_root_path, code)) if os.path.isdir(os.path.join(self.data_root_path, code)) else [])
                if f.endswith(extension)
            )
            path = next((p for p in cand if os.path.exists(p)), None)
            if path is None:
                raise ValueError(f"Object {code} not found under {self.data_root_path}")
            fv = load_obj_triangles(path) * self.scale
            if len(np.unique(fv.reshape(-1, 3), axis=0)) < 100:
                raise ValueError(f"Object {code} has too few vertices, please check the data.")  # object_model.py:118
            fvs.append(fv.astype(np.float32))
        self.initialize_from_meshes(fvs, object_code_list)

    def initialize_from_meshes(self, face_verts_list, object_code_list=None, surface_points_list=None, generator=None):
        """``surface_points_list`` None: the ``num_samples`` surface points of every object are drawn ON THE DEVICE the way
        the reference does (object_model.py:163-178: 100 x num_samples area-weighted samples, farthest-point sampling
        from sample 0; ``gq_surface_fps``), then Morton-ordered; a list injects them (tests, fixed workloads)."""
        self.object_code_list = object_code_list or [f"obj{i}" for i in range(len(face_verts_list))]
        self.object_mesh_list = [np.asarray(f, dtype=np.float32) for f in face_verts_list]
        self.object_face_verts_list = [torch.tensor(f, device=self.device) for f in self.object_mesh_list]
        self._meshset = ops.MeshSet(self.object_mesh_list)
        n_obj = len(face_verts_list)
        self.object_scale_tensor = torch.ones(n_obj, self.batch_size_each, device=self.device)  # scale_choice = [1.0]
        if self.num_samples != 0:
            if surface_points_list is None:
                sp = ops.morton_sort_points(ops.surface_fps(self.object_mesh_list, self.num_samples, 100, generator, device=self.device))
            else:
                sp = torch.tensor(np.stack(surface_points_list), dtype=torch.float32, device=self.device)
            self.surface_points_each = sp.contiguous()
            self.surface_points_tensor = sp.repeat_interleave(self.batch_size_each, dim=0)
        self._cog = None
        self._hull = None

    def convex_hulls(self):
        """Convex hull of every object (scaled like initializations.py:42-46) as device arrays for
        ``initialize_convex_hull``: triangles oriented outward, per-object cumulative area table, offsets."""
        if self._hull is None:
            fvs, cdfs, off = [], [], [0]
            for i, f in enumerate(self.object_mesh_list):
                h = mesh_utils.convex_hull_faces(f.reshape(-1, 3) * float(self.object_scale_tensor[i].max().item()))
                fvs.append(h)
                cdfs.append(mesh_utils.area_cdf(h))
                off.append(off[-1] + len(h))
            self._hull = (torch.tensor(np.concatenate(fvs), dtype=torch.float32, device=self.device).contiguous(),
                          torch.tensor(np.concatenate(cdfs), dtype=torch.float32, device=self.device).contiguous(),
                          torch.tensor(off, dtype=torch.int32, device=self.device))
            self.hull_face_verts_list = fvs
        return self._hull

    def attach(self, hand_model):
        hand_model._batch_each_hint = self.batch_size_each

    # object_model.py:186-255
    def cal_distance(self, x, with_closest_points=False):
        _, n_points, _ = x.shape
        d2, sgn, nrm, cls = ops.sdf_meshset(x.reshape(-1, 3), self._meshset, self.batch_size_each * n_points)
        dis, normals = ops.signed_distance(d2, sgn, nrm)  # sqrt(d2 + 1e-8) * (-sgn), nrm * sgn
        distance = dis.reshape(-1, n_points)
        normals = normals.reshape(-1, n_points, 3)
        if with_closest_points:
            return distance, normals, cls.reshape(-1, n_points, 3)
        return distance, normals
