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
    joint_names: List[str] = field(default_factory=list)  # actuated joints, DFS order
    joints_lower: np.ndarray = None  # (J,)
    joints_upper: np.ndarray = None  # (J,)
    default_state: np.ndarray = None  # (J,)
    # ---- mesh links ----------------------------------------------------------------------------
    link_names: List[str] = field(default_factory=list)  # mesh links, DFS order (reference self.mesh)
    link_frame: np.ndarray = None  # (L,) int32 frame index
    link_face_offset: np.ndarray = None  # (L+1,) int32 prefix sum into face_verts
    face_verts: np.ndarray = None  # (sumF,3,3) float32, link frame, origins baked in
    # ---- contact candidates ----------------------------------------------------------------------
    cand_pos: np.ndarray = None  # (C,3) link frame
    cand_nrm: np.ndarray = None  # (C,3) link frame
    cand_link: np.ndarray = None  # (C,) int32 mesh-link index
    # ---- penetration spheres ---------------------------------------------------------------------
    sphere: np.ndarray = None  # (S,4) xyz r, link frame
    sphere_link: np.ndarray = None  # (S,) int32 mesh-link index (non-decreasing)
    # ---- reduced tree for kernels: one node per actuated joint -------------------------------------
    node_parent: np.ndarray = None  # (J,) int32 parent node (-1 = hand base)
    node_pre: np.ndarray = None  # (J,4,4) float32 fixed transform parent-node frame -> this joint frame
    node_axis: np.ndarray = None  # (J,3)
    node_type: np.ndarray = None  # (J,) int32 (REVOLUTE / PRISMATIC)
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
    def n_links(self) -> int:
        return len(self.link_names)

    @property
    def n_contact_candidates(self) -> int:
        return int(self.cand_pos.shape[0])

    @property
    def n_spheres(self) -> int:
        return int(self.sphere.shape[0])

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
            else:
                out[f.name] = v
        np.savez_compressed(path, **out)

    @classmethod
    def load(cls, path: str) -> "HandSpec":
        z = np.load(path, allow_pickle=False)
        kw = {}
        for f in fields(cls):
            v = z[f.name]
            if f.name in ("frame_names", "joint_names", "link_names"):
                kw[f.name] = [str(s) for s in v.tolist()]
            elif f.name == "name":
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
        last_T = np.eye(4)
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
            tv = load_obj_triangles(p) * scale
            T = _origin(e)
            tv = tv @ T[:3, :3].T + T[:3, 3]
            tris.append(tv)
            last_T = T
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

    ds = np.zeros(J) if default_state is None else np.asarray(default_state, dtype=np.float64)
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
        link_names=link_names,
        link_frame=i32(link_frame),
        link_face_offset=i32(face_off),
        face_verts=f32(np.concatenate(face_chunks, 0).reshape(-1, 3, 3)),
        cand_pos=f32(np.concatenate(cand_pos, 0) if cand_pos else np.zeros((0, 3))),
        cand_nrm=f32(np.concatenate(cand_nrm, 0) if cand_nrm else np.zeros((0, 3))),
        cand_link=i32(cand_link),
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
