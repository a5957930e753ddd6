"""Triangle-soup readers for the mesh formats the reference's hand assets use (set-up time, numpy + stdlib only).

The reference loads every link mesh with ``trimesh.load_mesh(path, process=False)`` (core/hand_model.py:199) and uses
``.vertices`` / ``.faces`` of the result: nothing is merged or dropped, and for a scene-type file (COLLADA) the geometry
arrives with the scene graph applied -- node transforms concatenated from the root to every ``<instance_geometry>`` -- and
all instances concatenated.  trimesh is not installable here, so these readers restate that contract:

* Wavefront OBJ   ``load_obj_triangles`` (hands/spec.py; fan triangulation)
* STL             binary (80-byte header, uint32 count, 50-byte records) and ASCII (``facet normal`` / ``vertex``)
* COLLADA 1.4     ``<library_geometries>`` meshes with ``<triangles>`` / ``<polylist>`` / ``<polygons>`` primitives,
                  ``<library_visual_scenes>`` node trees with ``<matrix>`` / ``<translate>`` / ``<rotate>`` / ``<scale>`` in
                  document order, ``<instance_geometry>`` and ``<instance_node>``; ``<asset><unit meter=..>`` is NOT
                  applied (trimesh records it as metadata only) and ``<up_axis>`` does not re-orient the data

PARITY UNPINNED against trimesh / pycollada (neither can run here); the geometry is cross-checked in
tests/test_fixtures_and_assets.py against the STL twins the Schunk asset directory ships for the same links.
"""

from __future__ import annotations

import os
import struct
import xml.etree.ElementTree as ET
from typing import Dict, List

import numpy as np


def load_stl_triangles(path: str) -> np.ndarray:
    """(F,3,3) float64 corner positions of an STL file, in file order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) >= 84:
        n = struct.unpack("<I", data[80:84])[0]
        if 84 + 50 * n == len(data):  # binary: the size matches the record count
            rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
            return rec["v"].astype(np.float64)
    tris: List[List[float]] = []
    for line in data.decode("ascii", errors="ignore").splitlines():
        t = line.split()
        if len(t) == 4 and t[0] == "vertex":
            tris.append([float(t[1]), float(t[2]), float(t[3])])
    if len(tris) == 0 or len(tris) % 3:
        raise ValueError(f"{path}: neither a binary nor an ASCII STL")
    return np.asarray(tris, dtype=np.float64).reshape(-1, 3, 3)


def _strip(tag: str) -> str:
    return tag.split("}", 1)[1] if "}" in tag else tag


def _children(e, name):
    return [c for c in e if _strip(c.tag) == name]


def _child(e, name):
    c = _children(e, name)
    return c[0] if c else None


def _floats(text) -> np.ndarray:
    return np.asarray((text or "").split(), dtype=np.float64)


def _rotation(axis, deg) -> np.ndarray:
    a = np.asarray(axis, dtype=np.float64)
    a = a / max(np.linalg.norm(a), 1e-30)
    t = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(4)
    R[:3, :3] = np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * (K @ K)
    return R


def _node_matrix(node) -> np.ndarray:
    """Product of the transform elements of a <node>, in document order (COLLADA 1.4 spec, 'node')."""
    M = np.eye(4)
    for c in node:
        tag = _strip(c.tag)
        if tag == "matrix":
            M = M @ _floats(c.text).reshape(4, 4)
        elif tag == "translate":
            T = np.eye(4)
            T[:3, 3] = _floats(c.text)
            M = M @ T
        elif tag == "rotate":
            v = _floats(c.text)
            M = M @ _rotation(v[:3], v[3])
        elif tag == "scale":
            M = M @ np.diag(np.append(_floats(c.text), 1.0))
    return M


def _geometry_triangles(geom) -> np.ndarray:
    """(F,3,3) corner positions of one <geometry><mesh>, primitives in document order."""
    mesh = _child(geom, "mesh")
    if mesh is None:
        return np.zeros((0, 3, 3))
    sources: Dict[str, np.ndarray] = {}
    for s in _children(mesh, "source"):
        fa = _child(s, "float_array")
        if fa is None:
            continue
        stride = 3
        tc = _child(s, "technique_common")
        acc = _child(tc, "accessor") if tc is not None else None
        if acc is not None and acc.get("stride"):
            stride = int(acc.get("stride"))
        sources[s.get("id")] = _floats(fa.text).reshape(-1, stride)
    vert_src: Dict[str, str] = {}
    for v in _children(mesh, "vertices"):
        for i in _children(v, "input"):
            if i.get("semantic") == "POSITION":
                vert_src[v.get("id")] = i.get("source").lstrip("#")
    out = []
    for prim in mesh:
        kind = _strip(prim.tag)
        if kind not in ("triangles", "polylist", "polygons"):
            continue
        inputs = _children(prim, "input")
        stride = 1 + max(int(i.get("offset", "0")) for i in inputs)
        vin = next((i for i in inputs if i.get("semantic") == "VERTEX"), None)
        if vin is None:
            continue
        pos = sources[vert_src[vin.get("source").lstrip("#")]][:, :3]
        voff = int(vin.get("offset", "0"))
        if kind == "polygons":
            polys = [np.asarray(p.text.split(), dtype=np.int64).reshape(-1, stride)[:, voff] for p in _children(prim, "p")]
        else:
            idx = np.asarray((_child(prim, "p").text or "").split(), dtype=np.int64).reshape(-1, stride)[:, voff]
            if kind == "triangles":
                out.append(pos[idx.reshape(-1, 3)])
                continue
            counts = np.asarray(_child(prim, "vcount").text.split(), dtype=np.int64)
            ends = np.cumsum(counts)
            polys = [idx[e - c:e] for c, e in zip(counts, ends)]
        tri = [[p[0], p[k], p[k + 1]] for p in polys for k in range(1, len(p) - 1)]  # fan, like the OBJ reader
        if tri:
            out.append(pos[np.asarray(tri)])
    return np.concatenate(out, 0) if out else np.zeros((0, 3, 3))


def load_dae_triangles(path: str) -> np.ndarray:
    """(F,3,3) float64: every instantiated geometry of the default visual scene with its node transforms applied,
    concatenated in scene-graph (document, depth-first) order."""
    root = ET.parse(path).getroot()
    geoms = {}
    lg = _child(root, "library_geometries")
    for g in (_children(lg, "geometry") if lg is not None else []):
        geoms[g.get("id")] = g
    lib_nodes = {}
    ln = _child(root, "library_nodes")
    for n in (_children(ln, "node") if ln is not None else []):
        lib_nodes[n.get("id")] = n
    scenes = _child(root, "library_visual_scenes")
    vs = _children(scenes, "visual_scene") if scenes is not None else []
    sc = _child(root, "scene")
    inst = _child(sc, "instance_visual_scene") if sc is not None else None
    if inst is not None:
        want = inst.get("url", "").lstrip("#")
        vs = [v for v in vs if v.get("id") == want] or vs
    cache: Dict[str, np.ndarray] = {}
    out = []

    def walk(node, M):
        M = M @ _node_matrix(node)
        for c in node:
            tag = _strip(c.tag)
            if tag == "instance_geometry":
                gid = c.get("url", "").lstrip("#")
                if gid in geoms:
                    if gid not in cache:
                        cache[gid] = _geometry_triangles(geoms[gid])
                    t = cache[gid]
                    out.append(t @ M[:3, :3].T + M[:3, 3])
            elif tag == "instance_node":
                nid = c.get("url", "").lstrip("#")
                if nid in lib_nodes:
                    walk(lib_nodes[nid], M)
            elif tag == "node":
                walk(c, M)

    for v in vs[:1]:
        for n in _children(v, "node"):
            walk(n, np.eye(4))
    if not out:  # a file without a scene: the bare geometries
        out = [_geometry_triangles(g) for g in geoms.values()]
    return np.concatenate(out, 0) if out else np.zeros((0, 3, 3))


def load_mesh_triangles(path: str) -> np.ndarray:
    """(F,3,3) float64 triangles of an OBJ / STL / DAE file (reference hand_model.py:187-199 accepts exactly these)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        from .spec import load_obj_triangles

        return load_obj_triangles(path)
    if ext == ".stl":
        return load_stl_triangles(path)
    if ext == ".dae":
        return load_dae_triangles(path)
    raise NotImplementedError(f"{path}: unsupported mesh format {ext}")
