"""Hand specification: everything the grasp hot path needs to know about a hand, as flat arrays.

A ``HandSpec`` is the setup-time product of what the reference does in
``HandModel.__init__`` / ``_parse_mjcf`` (reference ``graspqp/src/graspqp/core/hand_model.py:144-393,
395-696``): parse the URDF kinematic tree, bake visual/collision origins into mesh vertices,
collect contact candidates (+normals) and penetration spheres per link, read joint limits.
It is plain data (numpy, float32/int32) so that the HIP kernels, the host mirror and the test
oracle all consume the *same* numbers; it serialises to ``.npz``.

Conventions kept from the reference (they are semantics, not bugs to fix):

* frames are visited depth-first, children in URDF joint *file order* (pytorch_kinematics'
  ``build_chain_from_urdf``); the actuated-joint order is the DFS order of non-fixed joints.
* "mesh links" are the links that carry a visual or collision geometry, in DFS order
  (``self.mesh`` in the reference); collision geometry wins when present
  (``use_collision_if_possible=True``, hand_model.py:224-235).
* contact candidates / normals come from ``meshes/contact_infos.json`` -- the reference's own dump
  of ``contact_candidates`` / ``normal_candidates`` (hand_model.py:381-391), i.e. already expressed
  in the link frame *after* the last-visual-origin transform (hand_model.py:296-303).
* penetration spheres ``[x, y, z, r]`` are transformed by the origin of the *last* visual/collision
  element of their link (leaked loop variable, hand_model.py:323-325).

The tree is also stored in a *reduced* form for the kernels: fixed joints are folded into their
children, leaving one node per actuated joint (``node_*``) and one constant offset per mesh link.
"""

from __future__ import annotations

import json
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field, fields
from typing import Dict, List, Optional

import numpy as np

JOINT_FIXED, JOINT_REVOLUTE, JOINT_PRISMATIC = 0, 1, 2
_JT = {"fixed": JOINT_FIXED, "revolute": JOINT_REVOLUTE, "continuous": JOINT_REVOLUTE, "prismatic": JOINT_PRISMATIC}


def rpy_to_matrix(rpy) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw -> R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ],
        dtype=np.float64,
    )


def _origin(elem) -> np.ndarray:
    """4x4 transform of an <origin xyz rpy> child (identity if absent)."""
    T = np.eye(4)
    if elem is None:
        return T
    o = elem.find("origin")
    if o is None:
        return T
    xyz = [float(v) for v in o.get("xyz", "0 0 0").split()]
    rpy = [float(v) for v in o.get("rpy", "0 0 0").split()]
    T[:3, :3] = rpy_to_matrix(rpy)
    T[:3, 3] = xyz
    return T


def load_mesh_triangles(path: str) -> np.ndarray:
    from .mesh_io import load_mesh_triangles as _load

    return _load(path)


def load_obj_triangles(path: str) -> np.ndarray:
    """(F,3,3) float64 triangle corner positions of a Wavefront OBJ, in file order.

    Polygons are fan-triangulated about their first corner -- for quads that is the
    (0,1,2),(0,2,3) split, the same diagonal trimesh uses; the reference loads with
    ``process=False`` so nothing is merged or dropped (hand_model.py:199).
    """
    verts: List[List[float]] = []
    tris: List[List[int]] = []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    tris.append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    t = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    return v[t]


@dataclass
class HandSpec:
    name: str = ""
    # ---- full frame tree (one frame per URDF link, DFS order) ---------------------------------
    frame_names: List[str] = field(default_factory=list)
    frame_parent: np.ndarray = None  # (F,) int32, -1 for root
    frame_joint_type: np.ndarray = None  # (F,) int32
    frame_dof: np.ndarray = None  # (F,) int32 index into joint angles, -1 if fixed
    frame_origin: np.ndarray = None  # (F,4,4) float32 joint origin (parent link frame -> joint frame)
    frame_axis: np.ndarray = None  # (F,3) float32 unit axis in joint frame
    joint_names: List[str] = field(default_factory=list)  # ACTUATED joints, DFS order (reference _actuated_joints_names)
    joints_lower: np.ndarray = None  # (J,)  J = number of actuated joints = pose dimension - 9
    joints_upper: np.ndarray = None  # (J,)
    default_state: np.ndarray = None  # (J,)
    # coupled hands (reference hands/{ability_hand,panda,schunk}.py: joint_filter + joint_calc_fnc / jacobian_fnc): the
    # kinematic tree has N >= J moving joints, theta_full = coupling @ theta_actuated + coupling_offset.  For the other
    # hands N == J and coupling is the identity.
    full_joint_names: List[str] = field(default_factory=list)  # all moving joints of the tree, DFS order (N)
    coupling: np.ndarray = None  # (N,J) float32
    coupling_offset: np.ndarray = None  # (N,) float32
    # grasp-type link subsets (reference <hand>/eigengrasps.json, hand_model.py:438-451): JSON text, "" if the hand has none
    eigengrasps: str = ""
    # ---- mesh links ----------------------------------------------------------------------------
    link_names: List[str] = field(default_factory=list)  # mesh links, DFS order (reference self.mesh)
    link_frame: np.ndarray = None  # (L,) int32 frame index
    link_face_offset: np.ndarray = None  # (L+1,) int32 prefix sum into face_verts
    face_verts: np.ndarray = None  # (sumF,3,3) float32, link frame, origins baked in
    # ---- contact candidates ----------------------------------------------------------------------
    cand_pos: np.ndarray = None  # (C,3) link frame
    cand_nrm: np.ndarray = None  # (C,3) link frame
    cand_link: np.ndarray = None  # (C,) int32 mesh-link index
    # contact patches the reference samples its candidates from (contact_points.json -> [mesh file, n],
    # hand_model.py:269-296): triangles in the link frame (visual scale and offset applied); only read when a grasp type
    # asks for MORE candidates on a link than the reference's dump holds
    patch_verts: np.ndarray = None  # (sumP,3,3) float32
    patch_face_offset: np.ndarray = None  # (L+1,) int32 prefix sum into patch_verts
    # ---- penetration spheres ---------------------------------------------------------------------
    sphere: np.ndarray = None  # (S,4) xyz r, link frame
    sphere_link: np.ndarray = None  # (S,) int32 mesh-link index (non-decreasing)
    # ---- reduced tree for kernels: one node per moving joint of the tree (N) ---------------------------
    node_parent: np.ndarray = None  # (N,) int32 parent node (-1 = hand base)
    node_pre: np.ndarray = None  # (N,4,4) float32 fixed transform parent-node frame -> this joint frame
    node_axis: np.ndarray = None  # (N,3)
    node_type: np.ndarray = None  # (N,) int32 (REVOLUTE / PRISMATIC)
    link_node: np.ndarray = None  # (L,) int32 node the mesh link rides on (-1 = base)
    link_offset: np.ndarray = None  # (L,4,4) float32 constant transform node frame -> link frame
    # ---- axes used by initialisation ----------------------------------------------------------------
    forward_axis: np.ndarray = None
    up_axis: np.ndarray = None
    grasp_axis: np.ndarray = None

    # convenience -----------------------------------------------------------------------------------
    @property
    def n_dofs(self) -> int:
        return len(self.joint_names)

    @property
    def n_nodes(self) -> int:
        return len(self.full_joint_names)

    @property
    def is_coupled(self) -> bool:
        return self.n_nodes != self.n_dofs or not np.array_equal(self.coupling, np.eye(self.n_dofs, dtype=np.float32)) \
            or bool(np.any(self.coupling_offset != 0))

    def full_joint_angles(self, theta):
        """theta (..., J) actuated -> (..., N) angles of every moving joint (numpy or torch)."""
        if hasattr(theta, "detach"):
            import torch

            C = torch.as_tensor(self.coupling, dtype=theta.dtype, device=theta.device)
            return theta @ C.T + torch.as_tensor(self.coupling_offset, dtype=theta.dtype, device=theta.device)
        return np.asarray(theta) @ self.coupling.T + self.coupling_offset

    @property
    def n_links(self) -> int:
        return len(self.link_names)

    @property
    def n_contact_candidates(self) -> int:
        return int(self.cand_pos.shape[0])

    @property
    def n_spheres(self) -> int:
        return int(self.sphere.shape[0])

    def grasp_types(self) -> List[str]:
        return sorted(json.loads(self.eigengrasps)) if self.eigengrasps else []

    def with_grasp_type(self, grasp_type: str) -> "HandSpec":
        """The hand restricted to a grasp type of its eigengrasps.json (reference hand_model.py:438-451,262-263,278-279):
        contact candidates only on the listed links, ``n_points`` of them per link, and the default joint state of the
        unused fingers moved to the upper limit for "pinch" / "precision" (hand_model.py:550-589).  The reference samples
        the candidates of a link by farthest-point sampling from a fixed seed, so the n_points candidates of a link are
        the first n_points of its default set (farthest-point sequences are nested); a request for MORE points than the
        dumped set holds cannot be honoured (no trimesh here) and raises."""
        import copy

        if not self.eigengrasps:
            raise ValueError(f"{self.name}: no eigengrasps.json for this hand")
        data = json.loads(self.eigengrasps)
        if grasp_type not in data:
            raise ValueError(f"grasp type {grasp_type} not found in eigengrasps.json. Available grasp types are {list(data.keys())}")
        links = data[grasp_type]
        keep, extra = [], []
        for li, lname in enumerate(self.link_names):
            idx = np.nonzero(self.cand_link == li)[0]
            if lname not in links or len(idx) == 0:
                continue
            k = int(links[lname].get("n_points", len(idx))) if isinstance(links[lname], dict) else len(idx)
            if k > len(idx):
                extra.append((li, *self._more_candidates(li, idx, k)))
            keep.append(idx[:k])
        s = copy.copy(self)
        keep = np.concatenate(keep) if keep else np.zeros(0, dtype=np.int64)
        s.cand_pos, s.cand_nrm, s.cand_link = self.cand_pos[keep], self.cand_nrm[keep], self.cand_link[keep]
        for li, pos, nrm in extra:  # the new candidates follow the dumped ones of their link (candidates stay link-sorted)
            at = int(np.searchsorted(s.cand_link, li, side="right"))
            s.cand_pos = np.concatenate([s.cand_pos[:at], pos, s.cand_pos[at:]]).astype(np.float32)
            s.cand_nrm = np.concatenate([s.cand_nrm[:at], nrm, s.cand_nrm[at:]]).astype(np.float32)
            s.cand_link = np.concatenate([s.cand_link[:at], np.full(len(pos), li, np.int32), s.cand_link[at:]]).astype(np.int32)
        s.contact_links = links
        ds = self.default_state.copy()
        rules = {  # hand_model.py:550-589: (substrings of the fingers to fold away, joint-name parts that are left alone)
            ("allegro", "pinch"): (("middle", "ring"), ("joint_0",)), ("allegro", "precision"): (("ring",), ("joint_0",)),
            ("shadow_hand", "pinch"): (("MF", "RF", "LF"), ("J3", "LFJ4")), ("shadow_hand", "precision"): (("RF", "LF"), ("J3", "LFJ4")),
            ("ability_hand", "pinch"): (("middle", "ring", "pinky"), ()), ("ability_hand", "precision"): (("ring", "pinky"), ()),
        }
        r = rules.get((self.name, grasp_type))
        if r is not None:
            for i, jn in enumerate(self.joint_names):
                if any(f in jn for f in r[0]) and not any(x in jn for x in r[1]):
                    ds[i] = self.joints_upper[i]
        s.default_state = ds
        s.grasp_type = grasp_type
        return s

    def _more_candidates(self, li: int, have: np.ndarray, k: int):
        """k - len(have) further contact candidates on mesh link ``li`` (reference hand_model.py:269-296,333-335: 1000 even
        surface samples of the link's contact patch -> farthest-point sampling -> normal of the closest link-mesh face).
        The reference's sample stream (trimesh + numpy seed 42) cannot be reproduced here, so the candidates its dump
        holds stay the first ones -- farthest-point sequences are nested, they ARE the reference's first len(have) -- and
        the sequence is continued over a natively sampled pool (parity unpinned for the added points)."""
        from ..utils import meshes

        if self.patch_face_offset is None or self.patch_face_offset[li + 1] == self.patch_face_offset[li]:
            raise NotImplementedError(f"{self.name}: {k} contact candidates requested on {self.link_names[li]}, the "
                                      f"reference's dump holds {len(have)} and the spec has no contact patch for the link")
        patch = self.patch_verts[self.patch_face_offset[li]:self.patch_face_offset[li + 1]]
        pool = meshes.sample_surface(patch, 1000, seed=42)
        got = self.cand_pos[have].astype(np.float64)
        dist = ((pool[:, None] - got[None]) ** 2).sum(-1).min(1) if len(got) else np.full(len(pool), np.inf)
        new = []
        for _ in range(k - len(have)):
            j = int(np.argmax(dist))
            new.append(pool[j])
            dist = np.minimum(dist, ((pool - pool[j]) ** 2).sum(1))
        new = np.asarray(new)
        fv = self.link_faces(li).astype(np.float64)
        _, _, fi = meshes.closest_face(new, fv)
        n = np.cross(fv[fi, 1] - fv[fi, 0], fv[fi, 2] - fv[fi, 0])
        n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
        return new.astype(np.float32), n.astype(np.float32)

    def link_faces(self, l: int) -> np.ndarray:
        return self.face_verts[self.link_face_offset[l] : self.link_face_offset[l + 1]]

    def save(self, path: str) -> None:
        out = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, list):
                out[f.name] = np.array(v, dtype=np.str_)
            elif isinstance(v, str):
                out[f.name] = np.array(v, dtype=np.str_)
            elif v is None:
                continue
            else:
                out[f.name] = v
        np.savez_compressed(path, **out)

    @classmethod
    def load(cls, path: str) -> "HandSpec":
        z = np.load(path, allow_pickle=False)
        kw = {}
        for f in fields(cls):
            if f.name not in z.files and f.name in ("patch_verts", "patch_face_offset"):
                continue  # specs written before the contact patches were stored
            v = z[f.name]
            if f.name in ("frame_names", "joint_names", "link_names", "full_joint_names"):
                kw[f.name] = [str(s) for s in v.tolist()]
            elif f.name in ("name", "eigengrasps"):
                kw[f.name] = str(v)
            else:
                kw[f.name] = v
        return cls(**kw)


_AXES = {
    "x": [1, 0, 0],
    "y": [0, 1, 0],
    "z": [0, 0, 1],
    "-x": [-1, 0, 0],
    "-y": [0, -1, 0],
    "-z": [0, 0, -1],
}


def build_hand_spec(
    name: str,
    urdf_path: str,
    mesh_path: str,
    penetration_points_path: Optional[str],
    contact_infos_path: Optional[str] = None,
    default_state=None,
    forward_axis: str = "x",
    up_axis: str = "z",
    grasp_axis: Optional[str] = None,
    use_collision_if_possible: bool = True,
    only_use_collision: bool = False,
    joint_filter: Optional[List[str]] = None,
    coupling: Optional[Dict[str, tuple]] = None,
    eigengrasps_path: Optional[str] = None,
    contact_points_path: Optional[str] = None,
    contact_mesh_root: Optional[str] = None,
) -> HandSpec:
    """URDF + OBJ meshes + JSON side files -> HandSpec (reference hand_model.py:395-696)."""
    root = ET.parse(urdf_path).getroot()
    links: Dict[str, ET.Element] = {l.get("name"): l for l in root.findall("link")}
    joints = root.findall("joint")
    children = {j.find("child").get("link") for j in joints}
    roots = [n for n in links if n not in children]
    if len(roots) != 1:
        raise ValueError(f"URDF {urdf_path} must have exactly one root link, found {roots}")

    frame_names, frame_parent, frame_jt, frame_dof, frame_origin, frame_axis = [], [], [], [], [], []
    joint_names, lo, hi = [], [], []

    def visit(link_name: str, parent: int, joint: Optional[ET.Element]):
        idx = len(frame_names)
        frame_names.append(link_name)
        frame_parent.append(parent)
        if joint is None:
            frame_jt.append(JOINT_FIXED)
            frame_dof.append(-1)
            frame_origin.append(np.eye(4))
            frame_axis.append([1.0, 0.0, 0.0])
        else:
            jt = _JT[joint.get("type")]
            frame_jt.append(jt)
            frame_origin.append(_origin(joint))
            ax = joint.find("axis")
            a = np.array([float(v) for v in ax.get("xyz").split()]) if ax is not None else np.array([1.0, 0, 0])
            frame_axis.append((a / max(np.linalg.norm(a), 1e-12)).tolist())
            if jt != JOINT_FIXED:
                frame_dof.append(len(joint_names))
                joint_names.append(joint.get("name"))
                lim = joint.find("limit")
                if lim is not None and lim.get("lower") is not None:
                    lo.append(float(lim.get("lower")))
                    hi.append(float(lim.get("upper")))
                else:
                    lo.append(-np.inf)
                    hi.append(np.inf)
            else:
                frame_dof.append(-1)
        for j in joints:  # children in joint file order (pytorch_kinematics urdf builder)
            if j.find("parent").get("link") == link_name:
                visit(j.find("child").get("link"), idx, j)

    visit(roots[0], -1, None)
    F = len(frame_names)

    # ---- geometry per link (hand_model.py:215-257) ------------------------------------------------
    pen_pts = json.load(open(penetration_points_path)) if penetration_points_path else {}
    cinfo = json.load(open(contact_infos_path)) if contact_infos_path else {}
    cpts = json.load(open(contact_points_path)) if contact_points_path and os.path.exists(contact_points_path) else {}
    patch_chunks, patch_off = [], [0]

    link_names, link_frame, face_chunks, face_off = [], [], [], [0]
    cand_pos, cand_nrm, cand_link = [], [], []
    sphere, sphere_link = [], []

    def geoms(link_elem, tag):
        return [g for g in link_elem.findall(tag)]

    for fi, lname in enumerate(frame_names):
        le = links[lname]
        vis, col = geoms(le, "visual"), geoms(le, "collision")
        if not ((len(vis) > 0 and not only_use_collision) or len(col) > 0):
            continue
        if not only_use_collision and (len(col) == 0 or not use_collision_if_possible):
            elems = vis
        else:
            elems = col
        tris = []
        last_T, last_scale = np.eye(4), np.ones(3)
        for e in elems:
            g = e.find("geometry")
            m = g.find("mesh") if g is not None else None
            if m is None:
                raise NotImplementedError(f"{lname}: only <mesh> geometries are supported in hand specs")
            fn = m.get("filename").replace("package://", "")
            p = os.path.join(mesh_path, fn)
            if not os.path.exists(p):
                p = os.path.join(mesh_path, os.path.basename(fn))
            if not os.path.exists(p):
                p = os.path.join(os.path.dirname(mesh_path), fn)
            if not os.path.exists(p):
                raise FileNotFoundError(p)
            scale = np.array([float(v) for v in (m.get("scale") or "1 1 1").split()])
            tv = load_mesh_triangles(p) * scale  # OBJ / STL / COLLADA (hands/mesh_io.py)
            T = _origin(e)
            tv = tv @ T[:3, :3].T + T[:3, 3]
            tris.append(tv)
            last_T, last_scale = T, scale
        li = len(link_names)
        link_names.append(lname)
        link_frame.append(fi)
        tv = np.concatenate(tris, 0) if tris else np.zeros((0, 3, 3))
        face_chunks.append(tv)
        face_off.append(face_off[-1] + len(tv))
        if lname in cinfo:
            cp = np.asarray(cinfo[lname]["contact_candidates"], dtype=np.float64).reshape(-1, 3)
            cn = np.asarray(cinfo[lname]["normal_candidates"], dtype=np.float64).reshape(-1, 3)
            cand_pos.append(cp)
            cand_nrm.append(cn)
            cand_link += [li] * len(cp)
        pt, sample_n = [], []
        for cand in (cpts.get(lname) or []):  # [mesh file, n]: the patch the reference samples n candidates from
            if isinstance(cand, list) and len(cand) == 2 and isinstance(cand[0], str):
                pm = os.path.normpath(os.path.join(contact_mesh_root or mesh_path, cand[0]))
                if os.path.exists(pm):
                    pv = load_mesh_triangles(pm) * last_scale
                    pt.append(pv @ last_T[:3, :3].T + last_T[:3, 3])
                    sample_n.append(int(cand[1]))
        patch_chunks.append(np.concatenate(pt, 0) if pt else np.zeros((0, 3, 3)))
        patch_off.append(patch_off[-1] + len(patch_chunks[-1]))
        if lname not in cinfo and pt and (not cinfo):
            # a hand WITHOUT the reference's contact_infos.json dump (schunk2): the candidates are sampled here the way
            # hand_model.py:269-296,333-335 does it -- 1000 surface samples of every contact patch, farthest-point sampling
            # from the first sample, normal = face normal of the closest link-mesh triangle.  The reference draws the
            # samples with trimesh (numpy seed 42) and pytorch3d; neither is available, so the sample SET differs:
            # PARITY UNPINNED for these points (same construction, same counts, same surfaces).
            from ..utils import meshes as _meshes

            for patch, k in zip(pt, sample_n):
                pool = _meshes.sample_surface(patch, 1000, seed=42)
                sel = _meshes.farthest_point_sampling(pool, k)
                _, _, fi = _meshes.closest_face(sel, tv)
                nn = np.cross(tv[fi, 1] - tv[fi, 0], tv[fi, 2] - tv[fi, 0])
                nn /= np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-30)
                cand_pos.append(sel)
                cand_nrm.append(nn)
                cand_link += [li] * len(sel)
        if lname in pen_pts and len(pen_pts[lname]) > 0:
            pk = np.asarray(pen_pts[lname], dtype=np.float64)
            if pk.shape[-1] == 4:
                r = pk[:, 3]
                c = pk[:, :3] @ last_T[:3, :3].T + last_T[:3, 3]  # last-visual-origin quirk
            else:
                r = np.full(len(pk), 0.01)
                c = pk[:, :3]
            sphere.append(np.concatenate([c, r[:, None]], 1))
            sphere_link += [li] * len(pk)

    L = len(link_names)

    # ---- reduced tree ---------------------------------------------------------------------------------
    # frame f: world = world(parent) @ origin_f @ motion_f(theta). Fold fixed frames.
    J = len(joint_names)
    node_parent = np.full(J, -1, np.int32)
    node_pre = np.zeros((J, 4, 4))
    node_axis = np.zeros((J, 3))
    node_type = np.zeros(J, np.int32)
    frame_node = np.full(F, -1, np.int32)  # node whose frame this frame rides on
    frame_off = np.zeros((F, 4, 4))  # constant transform node frame -> this frame
    for f in range(F):
        p = frame_parent[f]
        if p < 0:
            pn, poff = -1, np.eye(4)
        else:
            pn, poff = frame_node[p], frame_off[p]
        pre = poff @ frame_origin[f]
        if frame_jt[f] == JOINT_FIXED:
            frame_node[f] = pn
            frame_off[f] = pre
        else:
            d = frame_dof[f]
            node_parent[d] = pn
            node_pre[d] = pre
            node_axis[d] = frame_axis[f]
            node_type[d] = frame_jt[f]
            frame_node[f] = d
            frame_off[f] = np.eye(4)

    # ---- actuated joints and coupling (reference joint_filter / _joint_mask, hand_model.py:472-478,546-548;
    # joint_calc_fnc of hands/{ability_hand,panda,schunk}.py): theta_full = C theta_act + c0
    full_joint_names = list(joint_names)
    act = [n for n in full_joint_names if joint_filter is None or n in joint_filter]
    C = np.zeros((J, len(act)))
    c0 = np.zeros(J)
    for jn_i, jn in enumerate(full_joint_names):
        if coupling is not None and jn in coupling:
            src, mult, off = coupling[jn]
            C[jn_i, act.index(src)] = mult
            c0[jn_i] = off
        elif jn in act:
            C[jn_i, act.index(jn)] = 1.0
        else:
            raise ValueError(f"{name}: joint {jn} is neither actuated nor coupled to an actuated joint")
    mask = [full_joint_names.index(n) for n in act]
    lo, hi = [lo[i] for i in mask], [hi[i] for i in mask]
    joint_names = act
    ds = np.zeros(len(act)) if default_state is None else np.asarray(default_state, dtype=np.float64)
    eig = open(eigengrasps_path).read() if eigengrasps_path and os.path.exists(eigengrasps_path) else ""
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    i32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int32))
    spec = HandSpec(
        name=name,
        frame_names=frame_names,
        frame_parent=i32(frame_parent),
        frame_joint_type=i32(frame_jt),
        frame_dof=i32(frame_dof),
        frame_origin=f32(np.stack(frame_origin)),
        frame_axis=f32(frame_axis),
        joint_names=joint_names,
        joints_lower=f32(lo),
        joints_upper=f32(hi),
        default_state=f32(ds),
        full_joint_names=full_joint_names,
        coupling=f32(C),
        coupling_offset=f32(c0),
        eigengrasps=eig,
        link_names=link_names,
        link_frame=i32(link_frame),
        link_face_offset=i32(face_off),
        face_verts=f32(np.concatenate(face_chunks, 0).reshape(-1, 3, 3)),
        cand_pos=f32(np.concatenate(cand_pos, 0) if cand_pos else np.zeros((0, 3))),
        cand_nrm=f32(np.concatenate(cand_nrm, 0) if cand_nrm else np.zeros((0, 3))),
        cand_link=i32(cand_link),
        patch_verts=f32(np.concatenate(patch_chunks, 0).reshape(-1, 3, 3) if patch_chunks else np.zeros((0, 3, 3))),
        patch_face_offset=i32(patch_off),
        sphere=f32(np.concatenate(sphere, 0) if sphere else np.zeros((0, 4))),
        sphere_link=i32(sphere_link),
        node_parent=i32(node_parent),
        node_pre=f32(node_pre),
        node_axis=f32(node_axis),
        node_type=i32(node_type),
        link_node=i32([frame_node[f] for f in link_frame]),
        link_offset=f32(np.stack([frame_off[f] for f in link_frame]) if L else np.zeros((0, 4, 4))),
        forward_axis=f32(_AXES[forward_axis]),
        up_axis=f32(_AXES[up_axis]),
        grasp_axis=f32(_AXES[grasp_axis if grasp_axis is not None else forward_axis]),
    )
    return spec
