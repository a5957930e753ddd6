"""Procedural object meshes and surface sampling (setup-time, numpy).

The reference loads object meshes from a dataset directory with trimesh and samples 2500 surface
points with pytorch3d (``object_model.py:117,163-178``); neither the dataset nor those libraries
exist here, so the benchmark/test objects are generated procedurally: an icosphere
(config 1) and seeded watertight "YCB-style" superquadrics / noisy ellipsoids (configs 2-5).
All meshes are closed, consistently outward-oriented triangle soups ``(F,3,3) float32``.
"""

from __future__ import annotations

import numpy as np


def icosphere(subdiv: int = 3, radius: float = 0.05) -> np.ndarray:
    """Unit icosahedron subdivided ``subdiv`` times (3 -> 642 vertices / 1280 faces), scaled."""
    t = (1.0 + 5.0**0.5) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
         [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
         [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
         [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(subdiv):
        cache = {}
        verts = list(v)

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (verts[a] + verts[b]) * 0.5
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v = np.array(verts)
        f = np.array(nf, dtype=np.int64)
    return (v[f] * radius).astype(np.float32)


def superquadric(seed: int, n_u: int = 96, n_v: int = 48) -> np.ndarray:
    """Seeded watertight superquadric with smooth radial noise; extents 0.05-0.15 m.

    ~2*n_u*(n_v-1) faces (default ~9k).  Outward orientation.
    """
    rng = np.random.default_rng(seed)
    half = rng.uniform(0.025, 0.075, size=3)  # half extents
    e1, e2 = rng.uniform(0.3, 1.4, size=2)
    amp = rng.uniform(0.0, 0.08)
    ph = rng.uniform(0, 2 * np.pi, size=4)
    fr = rng.integers(2, 6, size=2)

    def spow(x, e):
        return np.sign(x) * np.abs(x) ** e

    def point(u, v):  # u in [0,2pi), v in [-pi/2, pi/2]
        u, v = np.broadcast_arrays(np.asarray(u, dtype=np.float64), np.asarray(v, dtype=np.float64))
        cu, su, cv, sv = np.cos(u), np.sin(u), np.cos(v), np.sin(v)
        p = np.stack([spow(cv, e1) * spow(cu, e2), spow(cv, e1) * spow(su, e2), spow(sv, e1)], -1)
        bump = 1.0 + amp * (np.sin(fr[0] * u + ph[0]) * np.cos(fr[1] * v + ph[1]) * np.cos(v))
        return p * half * bump[..., None]

    us = np.linspace(0, 2 * np.pi, n_u, endpoint=False)
    vs = np.linspace(-np.pi / 2, np.pi / 2, n_v + 1)[1:-1]
    grid = point(us[None, :], vs[:, None])  # (n_v-1, n_u, 3)
    south = point(np.array(0.0), np.array(-np.pi / 2))
    north = point(np.array(0.0), np.array(np.pi / 2))
    tris = []
    R = n_v - 1
    for j in range(n_u):
        j2 = (j + 1) % n_u
        tris.append([south, grid[0, j2], grid[0, j]])
        tris.append([north, grid[R - 1, j], grid[R - 1, j2]])
        for i in range(R - 1):
            a, b, c, d = grid[i, j], grid[i, j2], grid[i + 1, j2], grid[i + 1, j]
            tris.append([a, b, c])
            tris.append([a, c, d])
    fv = np.array(tris, dtype=np.float64)
    # make sure orientation is outward (signed volume > 0)
    vol = np.einsum("ij,ij->i", fv[:, 0], np.cross(fv[:, 1], fv[:, 2])).sum()
    if vol < 0:
        fv = fv[:, [0, 2, 1]]
    return fv.astype(np.float32)


def box(half=(0.03, 0.04, 0.05)) -> np.ndarray:
    """Axis-aligned box, 12 outward-oriented triangles."""
    hx, hy, hz = half
    c = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    q = [[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1], [2, 3, 7, 6], [0, 2, 6, 4], [1, 5, 7, 3]]
    tris = []
    for a, b, cc, d in q:
        tris += [[c[a], c[b], c[cc]], [c[a], c[cc], c[d]]]
    fv = np.array(tris)
    vol = np.einsum("ij,ij->i", fv[:, 0], np.cross(fv[:, 1], fv[:, 2])).sum()
    if vol < 0:
        fv = fv[:, [0, 2, 1]]
    return fv.astype(np.float32)


def sample_surface(face_verts: np.ndarray, n: int, seed: int = 42) -> np.ndarray:
    """Area-weighted uniform samples on a triangle soup -> (n,3) float64."""
    rng = np.random.default_rng(seed)
    fv = face_verts.astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0]), axis=1)
    fi = rng.choice(len(fv), size=n, p=area / area.sum())
    r1, r2 = rng.random(n), rng.random(n)
    s = np.sqrt(r1)
    w0, w1, w2 = 1 - s, s * (1 - r2), s * r2
    return fv[fi, 0] * w0[:, None] + fv[fi, 1] * w1[:, None] + fv[fi, 2] * w2[:, None]


def farthest_point_sampling(points: np.ndarray, k: int, start: int = 0) -> np.ndarray:
    """Classic FPS (first point = index ``start``), returns (k,3)."""
    n = len(points)
    sel = np.empty(k, dtype=np.int64)
    dist = np.full(n, np.inf)
    cur = start
    for i in range(k):
        sel[i] = cur
        d = ((points - points[cur]) ** 2).sum(1)
        dist = np.minimum(dist, d)
        cur = int(np.argmax(dist))
    return points[sel]


def morton_order(points: np.ndarray, bits: int = 10) -> np.ndarray:
    """Permutation that sorts points along a 3-D Morton (Z-order) curve: neighbouring indices are spatial
    neighbours, so the 64 points of a wavefront hit the same hand links (coherent culling in the E_pen kernel)."""
    p = np.asarray(points, dtype=np.float64)
    lo, hi = p.min(0), p.max(0)
    q = ((p - lo) / np.maximum(hi - lo, 1e-12) * ((1 << bits) - 1)).astype(np.uint64)

    def spread(v):
        out = np.zeros_like(v)
        for b in range(bits):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out

    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")


def surface_points(face_verts: np.ndarray, num_samples: int = 2500, oversample: int = 20, seed: int = 42,
                   sort: bool = True) -> np.ndarray:
    """Dense area-weighted cloud -> FPS down to ``num_samples`` (object_model.py:163-178), float32, stored in
    Morton order (the set of points is what matters to E_pen / cog, not their order)."""
    dense = sample_surface(face_verts, oversample * num_samples, seed)
    pts = farthest_point_sampling(dense, num_samples)
    if sort:
        pts = pts[morton_order(pts)]
    return pts.astype(np.float32)


def convex_hull_faces(verts: np.ndarray) -> np.ndarray:
    """Outward-oriented triangles (F,3,3) float64 of the convex hull of ``verts`` (qhull through scipy; the reference
    takes ``trimesh.Trimesh.convex_hull``, initializations.py:42).  Degenerate faces are dropped (:47)."""
    from scipy.spatial import ConvexHull

    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    hull = ConvexHull(v)
    fv = v[hull.simplices]
    n = np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0])
    flip = (n * hull.equations[:, :3]).sum(1) < 0
    fv[flip] = fv[flip][:, [0, 2, 1]]
    area = 0.5 * np.linalg.norm(np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0]), axis=1)
    return fv[area > 1e-14]


def area_cdf(face_verts: np.ndarray) -> np.ndarray:
    """Cumulative face area / total area (the table trimesh.sample.sample_surface searches)."""
    fv = np.asarray(face_verts, dtype=np.float64)
    area = 0.5 * np.linalg.norm(np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0]), axis=1)
    c = np.cumsum(area)
    return c / c[-1]


def closest_face(points: np.ndarray, face_verts: np.ndarray):
    """Closest point of a triangle soup to every query: (closest (N,3), squared distance (N,), face index (N,)), float64.
    Asset set-up only (the contact normals of re-sampled candidates, reference hand_model.py:333-335 ->
    ``trimesh.proximity.closest_point``); the per-iteration queries are the HIP kernels.  Region walk of Ericson,
    Real-Time Collision Detection 5.1.5, vectorised over (N, F); ties go to the smallest face index."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 1, 3)
    fv = np.asarray(face_verts, dtype=np.float64)
    a, b, c = fv[None, :, 0], fv[None, :, 1], fv[None, :, 2]
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        den = va + vb + vc
        v_in, w_in = vb / den, vc / den
        t_ab, t_ac = d1 / (d1 - d3), d2 / (d2 - d6)
        t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    # barycentric (v, w) of the closest point: a + v ab + w ac; later assignments have lower priority
    v, w = v_in, w_in
    bc = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)
    v, w = np.where(bc, 1 - t_bc, v), np.where(bc, t_bc, w)
    e_ac = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
    v, w = np.where(e_ac, 0.0, v), np.where(e_ac, t_ac, w)
    vert_c = (d6 >= 0) & (d5 <= d6)
    v, w = np.where(vert_c, 0.0, v), np.where(vert_c, 1.0, w)
    e_ab = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
    v, w = np.where(e_ab, t_ab, v), np.where(e_ab, 0.0, w)
    vert_b = (d3 >= 0) & (d4 <= d3)
    v, w = np.where(vert_b, 1.0, v), np.where(vert_b, 0.0, w)
    vert_a = (d1 <= 0) & (d2 <= 0)
    v, w = np.where(vert_a, 0.0, v), np.where(vert_a, 0.0, w)
    v, w = np.nan_to_num(v), np.nan_to_num(w)  # degenerate faces collapse onto corner a
    q = a + v[..., None] * ab + w[..., None] * ac
    d2q = ((p - q) ** 2).sum(-1)
    fi = np.argmin(d2q, axis=1)
    n = np.arange(len(fi))
    return q[n, fi], d2q[n, fi], fi
