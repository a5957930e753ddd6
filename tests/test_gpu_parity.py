"""GPU parity tests: every HIP kernel, called through the C ABI, against the CPU oracle (oracle/ref_cpu, fp64) on the
same seeded inputs and against the committed golden fixtures.  Floating-point tolerances follow the stated contract
(SURVEY 8c / BASELINE.json north_star): total energy <= 1e-4 rel, E_fc <= 1e-4 rel at the median (the PDIPM iterate has
an fp32 noise tail), gradients <= 1e-3 rel norm-wise."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import ref_cpu  # noqa: E402
from ref_cpu import kin as okin  # noqa: E402
from ref_cpu import mala as omala  # noqa: E402
from ref_cpu import models as omodels  # noqa: E402
from ref_cpu import qp as oqp  # noqa: E402
from ref_cpu import sdf as osdf  # noqa: E402
from ref_cpu import span as ospan  # noqa: E402

from _parity import assert_tail_within_fp32_noise, oracle_fc_fp32_noise, rel_err  # noqa: E402
from graspqp_amd.hands import get_hand_spec  # noqa: E402
from graspqp_amd.utils import meshes  # noqa: E402


@pytest.fixture(scope="module")
def gq():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from graspqp_amd import _C, ops, stepper

    _C.lib()
    return type("gq", (), {"ops": ops, "C": _C, "stepper": stepper})


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _rel(a, b, floor=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


# ---------------------------------------------------------------------------------------------------------------
# SDF
# ---------------------------------------------------------------------------------------------------------------
def _check_sdf(gq, fv, pts, tag):
    d2, sg, nrm, cls = gq.ops.compute_sdf(torch.tensor(pts, device="cuda"), torch.tensor(fv, device="cuda"))
    torch.cuda.synchronize()
    od2, osg, onrm, ocls = osdf.compute_sdf(torch.tensor(pts, dtype=torch.float64), torch.tensor(fv, dtype=torch.float64))
    d2, sg, nrm, cls = d2.cpu().numpy(), sg.cpu().numpy(), nrm.cpu().numpy(), cls.cpu().numpy()
    od2, osg, onrm, ocls = od2.numpy(), osg.numpy(), onrm.numpy(), ocls.numpy()
    # distances: fp32 round-off on coordinates of size ~0.1 -> absolute 1e-7 on the distance
    np.testing.assert_allclose(np.sqrt(d2), np.sqrt(od2), rtol=1e-4, atol=2e-7, err_msg=tag)
    far = np.sqrt(od2) > 1e-5
    mism = (sg != osg) & far
    assert mism.mean() <= 2e-3, f"{tag}: sign mismatch rate {mism.mean()}"
    # closest points may legitimately differ when two features tie; the distance to the query must agree
    dd = np.linalg.norm(pts - cls, axis=1)
    np.testing.assert_allclose(dd, np.sqrt(od2), rtol=1e-4, atol=2e-7, err_msg=tag)
    agree = np.linalg.norm(cls - ocls, axis=1) < 1e-5
    assert agree.mean() > 0.99, f"{tag}: closest-point agreement {agree.mean()}"
    ok = agree & far
    np.testing.assert_allclose(nrm[ok], onrm[ok], atol=5e-3, err_msg=tag)


@pytest.mark.parametrize("mesh", ["sphere", "box", "superquadric", "allegro_link"])
def test_sdf_matches_oracle_wave_kernel(gq, mesh):
    rng = np.random.default_rng(0)
    if mesh == "sphere":
        fv = meshes.icosphere(2, 0.05)
    elif mesh == "box":
        fv = meshes.box()
    elif mesh == "superquadric":
        fv = meshes.superquadric(3, 32, 16)
    else:
        fv = get_hand_spec("allegro").link_faces(3)
    ext = np.abs(fv).max()
    pts = (rng.normal(size=(3000, 3)) * ext * 0.8).astype(np.float32)
    pts[:50] = fv.reshape(-1, 3)[rng.integers(0, fv.shape[0] * 3, 50)]  # queries exactly on vertices
    _check_sdf(gq, fv, pts, mesh)


def test_sdf_points_kernel_and_analytic_sphere(gq):
    """>= 131072 queries take the point-per-lane kernel; also checked against the closed-form sphere distance."""
    rng = np.random.default_rng(1)
    fv = meshes.icosphere(3, 0.05)
    pts = (rng.normal(size=(140000, 3)) * 0.05).astype(np.float32)
    d2, sg, nrm, cls = gq.ops.compute_sdf(torch.tensor(pts, device="cuda"), torch.tensor(fv, device="cuda"))
    r = np.linalg.norm(pts, axis=1)
    sd = np.sqrt(d2.cpu().numpy()) * sg.cpu().numpy()
    assert np.abs(sd - (r - 0.05)).max() < 3e-4  # faceting error of the 1280-face icosphere
    sub = slice(0, 4000)
    od2, osg, _, _ = osdf.compute_sdf(torch.tensor(pts[sub], dtype=torch.float64), torch.tensor(fv, dtype=torch.float64))
    np.testing.assert_allclose(np.sqrt(d2.cpu().numpy()[sub]), np.sqrt(od2.numpy()), rtol=1e-4, atol=2e-7)
    # the one-wavefront-per-query search (5000 queries) and the one-query-per-lane hierarchy (140000) agree on the distance
    # of the same queries (both finish the winner exactly; faces tied within the ranking noise may swap: a few ulp)
    d2b, _, _, _ = gq.ops.compute_sdf(torch.tensor(pts[:5000], device="cuda"), torch.tensor(fv, device="cuda"))
    assert (d2b.cpu() == d2[:5000].cpu()).float().mean() > 0.999
    np.testing.assert_allclose(d2b.cpu().numpy(), d2[:5000].cpu().numpy(), rtol=1e-5, atol=1e-10)


@pytest.mark.parametrize("mesh", ["allegro_link", "allegro_palm", "superquadric", "tiny", "shadow_link"])
def test_sdf_box_hierarchy_equals_the_face_loop(gq, mesh):
    """compute_sdf with >= 32768 queries goes one query per lane through the mesh's implicit 4-ary box hierarchy
    (csrc/bvh.hip; LDS-resident for the hand-link meshes, global memory for the 9024-face object).  It must return what the
    loop over ALL faces returns -- same winner rule -- for points far from the mesh (the reference's per-link calls: object
    surface points in the link frame, mostly centimetres away), near it, inside it and exactly on it."""
    rng = np.random.default_rng(11)
    if mesh == "allegro_link":
        fv = get_hand_spec("allegro").link_faces(3)      # 342 faces: 4 levels, LDS
    elif mesh == "allegro_palm":
        fv = get_hand_spec("allegro").link_faces(0)      # 324 faces
    elif mesh == "superquadric":
        fv = meshes.superquadric(0)                      # 9024 faces: 7 levels, global memory
    elif mesh == "tiny":
        fv = meshes.icosphere(1, 0.03)[:36]              # 36 faces: 2 levels, an OPEN surface
    else:
        fv = get_hand_spec("shadow_hand").link_faces(5)
    fv = np.ascontiguousarray(fv, dtype=np.float32)
    ext = float(np.abs(fv).max())
    N = 70001  # not a multiple of the 512-point chunks
    pts = np.concatenate([
        rng.normal(size=(30000, 3)) * ext * 4.0,                                   # far field
        rng.normal(size=(30000, 3)) * ext * 0.7,                                   # near / inside
        fv.reshape(-1, 3)[rng.integers(0, fv.shape[0] * 3, 5000)],                 # exactly on vertices
        fv.mean(1)[rng.integers(0, fv.shape[0], 5001)] + rng.normal(size=(5001, 3)) * 1e-4,  # a hair off face centres
    ]).astype(np.float32)
    assert pts.shape[0] == N
    p, f = torch.tensor(pts, device="cuda"), torch.tensor(fv, device="cuda")
    d2, sg, nrm, cls = gq.ops.compute_sdf(p, f)
    assert (id(f), "bvh") in gq.ops._MESH_CACHE
    d2l, sgl, nrml, clsl = torch.ops.graspqp_amd.compute_sdf(p, f)  # the face loop on the raw tensors
    torch.cuda.synchronize()
    same = (d2 == d2l)  # (queries ON a vertex tie several faces at distance ~0: another winner, a distance of 0 vs 1e-17)
    assert same.float().mean() > 0.99, float(same.float().mean())
    np.testing.assert_allclose(d2.cpu().numpy(), d2l.cpu().numpy(), rtol=2e-5, atol=1e-10)
    # where the same face won, everything is bit-identical; elsewhere (faces tied within the ranking noise) the closest
    # point may jump to the other face, but it is as close
    # (a query ON a vertex or an edge ties several faces at distance 0 +- ranking noise: the winner may differ, the closest
    # point is the same point computed through another face -- equal to round-off, not bit for bit)
    eq = (cls == clsl).all(1)
    assert eq.float().mean() > 0.99 and (sg[eq] == sgl[eq]).float().mean() > 0.999
    off = eq & (d2l > 1e-10)  # (at distance 0 the "normal" is the winning face's normal)
    np.testing.assert_allclose(nrm[off].cpu().numpy(), nrml[off].cpu().numpy(), atol=2e-4)
    dd = (p - cls).norm(dim=1)
    np.testing.assert_allclose(dd.cpu().numpy(), np.sqrt(d2l.cpu().numpy()), rtol=1e-4, atol=3e-7)
    sub = rng.choice(N, 1500, replace=False)
    od2, osg, _, _ = osdf.compute_sdf(torch.tensor(pts[sub], dtype=torch.float64), torch.tensor(fv, dtype=torch.float64))
    # (the Allegro palm has SLIVER faces -- fillet strips with sin^2 of the smallest angle down to 3e-6; queries that project
    # into one are classified by the edge functions of the face's own frame, tri.h, and agree with the fp64 oracle like any
    # other query.  With the barycentric constants of rounds 1-2 they could be up to half a sliver's width, 2e-4 m, off.)
    np.testing.assert_allclose(np.sqrt(d2.cpu().numpy()[sub]), np.sqrt(od2.numpy()), rtol=1e-5, atol=2e-7)
    assert (sg.cpu().numpy()[sub] == osg.numpy())[od2.numpy() > 1e-12].mean() > 0.999
    # the direction-sorted variant of the kernel (A/B switch): same answers, query for query
    gq.C.call("gq_debug_set_bvh_sorted", 1)
    try:
        d2s, sgs, _, clss = gq.ops.compute_sdf(p, f)
        torch.cuda.synchronize()
    finally:
        gq.C.call("gq_debug_set_bvh_sorted", 0)
    assert torch.equal(d2s, d2) and torch.equal(sgs, sg) and torch.equal(clss, cls)
    # gradient route (only dist_sq w.r.t. points)
    pg = p[:40000].clone().requires_grad_()
    d2g, _, _, clg = gq.ops.compute_sdf(pg, f)
    d2g.sum().backward()
    np.testing.assert_allclose(pg.grad.cpu().numpy(), 2 * (p[:40000] - clg).cpu().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("width", [4e-4, 5e-5, 2e-6])
def test_sdf_sliver_faces_are_classified_like_any_other_face(gq, width):
    """A ribbon sheet of 64 needle triangles (5 cm long, `width` wide: sin^2 of the smallest angle 6e-5 / 1e-6 / 1.6e-9),
    in a general orientation.  A query above the sheet's interior must get its height above the sheet, whichever sliver it
    projects into -- the inside test of tri.h works on the edge functions of the face's own 2-D frame; with barycentric
    constants (rounds 1-2) such queries were pushed to a sliver's edge.  Both routes of compute_sdf (face loop, box
    hierarchy), against the closed form and the fp64 oracle."""
    rng = np.random.default_rng(5)
    Lx, n = 0.05, 32
    tris = []
    for i in range(n):
        y0, y1 = i * width, (i + 1) * width
        tris += [[(0, y0, 0), (Lx, y0, 0), (Lx, y1, 0)], [(0, y0, 0), (Lx, y1, 0), (0, y1, 0)]]
    fv = np.array(tris, dtype=np.float64)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q *= np.sign(np.linalg.det(q))
    shift = np.array([0.013, -0.021, 0.008])
    fvw = (fv @ q.T + shift).astype(np.float32)
    N = 40000
    uv = np.stack([rng.uniform(0.02, 0.98, N) * Lx, rng.uniform(0.02, 0.98, N) * n * width], 1)
    h = np.exp(rng.uniform(np.log(1e-6), np.log(1e-2), N)) * rng.choice([-1.0, 1.0], N)
    local = np.concatenate([uv, h[:, None]], 1)
    local[-4000:, :2] += rng.normal(size=(4000, 2)) * 0.02  # some queries beyond the border of the sheet
    pts = (local @ q.T + shift).astype(np.float32)
    p, f = torch.tensor(pts, device="cuda"), torch.tensor(fvw, device="cuda")
    d2, sg, nrm, cls = gq.ops.compute_sdf(p, f)                       # box hierarchy (>= 32768 queries, >= 32 faces)
    assert (id(f), "bvh") in gq.ops._MESH_CACHE
    d2l, sgl, _, _ = torch.ops.graspqp_amd.compute_sdf(p, f)          # the loop over all faces
    torch.cuda.synchronize()
    # the mesh and the queries were rounded to fp32 after the rotation: heights are exact to ~|coordinate| * 6e-8
    inner = np.arange(N) < N - 4000
    for got in (d2, d2l):
        np.testing.assert_allclose(np.sqrt(got.cpu().numpy()[inner]), np.abs(h[inner]), rtol=2e-5, atol=3e-8)
    assert (sg.cpu().numpy()[inner] == np.sign(h[inner])).all() and (sgl.cpu().numpy()[inner] == np.sign(h[inner])).all()
    # the normal of an interior query is the sheet's normal.  Within ~sqrt(2 h dh) of an edge (dh ~ 3e-9: the fp32 error of
    # the height n.(p - a), |p - a| up to 5 cm) the neighbouring face may win by its edge -- an equal distance to fp32, a
    # normal tilted by up to sqrt(2 dh / h) ~ 3e-3: inherent to fp32 ranking, the same for every face shape
    nz = q[:, 2]
    big = inner & (np.abs(h) > 1e-4)
    nerr = np.abs(nrm.cpu().numpy()[big] - np.sign(h[big])[:, None] * nz[None]).max(1)
    # (rounding the rotated vertices to fp32, ~4e-9, also tilts every needle about its long axis by ~4e-9 / width)
    assert nerr.max() < 8e-3 + 1e-8 / width, nerr.max()
    assert width < 1e-4 or (nerr < 2e-4).mean() > 0.99, (nerr < 2e-4).mean()
    sub = rng.choice(N, 1500, replace=False)
    od2, _, _, _ = osdf.compute_sdf(torch.tensor(pts[sub], dtype=torch.float64), torch.tensor(fvw, dtype=torch.float64))
    for got in (d2, d2l):
        np.testing.assert_allclose(np.sqrt(got.cpu().numpy()[sub]), np.sqrt(od2.numpy()), rtol=2e-5, atol=2e-8)
    np.testing.assert_allclose((p - cls).norm(dim=1).cpu().numpy(), np.sqrt(d2.cpu().numpy()), rtol=1e-4, atol=3e-8)


def test_sdf_degenerate_faces_count_as_their_segments(gq):
    """Zero-area faces (two or three coinciding corners, three collinear corners) in an otherwise regular mesh: the face
    record of tri.h describes the segment such a face collapses to (its `gate` sends every query down the edge path), so the
    distance is the distance to that segment, the sign is +1 (dot with a zero normal >= 0) and the normal is the direction
    to the closest point -- for the face loop and for the box hierarchy, against segment distances computed in fp64."""
    rng = np.random.default_rng(9)
    reg = meshes.icosphere(1, 0.03).astype(np.float64)
    P = lambda *v: np.array(v, dtype=np.float64)
    a, b, c = P(0.06, 0.0, 0.0), P(0.09, 0.01, 0.0), P(0.06, 0.03, 0.02)
    deg = np.stack([
        np.stack([a, a, c]),                                  # a == b: the segment a-c
        np.stack([b, b, b]),                                  # a point
        np.stack([P(-0.06, 0, 0), P(-0.10, 0, 0), P(-0.08, 0, 0)]),      # collinear, c between a and b
        np.stack([P(0, 0.06, 0), P(0, 0.08, 0), P(0, 0.11, 0)]),         # collinear, c beyond b
        np.stack([P(0, -0.06, 0.01), P(0.02, -0.08, 0), P(0.02, -0.08, 0)]),   # b == c
        np.stack([P(0, 0, 0.07), P(0.01, 0.02, 0.09), P(0, 0, 0.07)]),         # a == c
    ])
    fv = np.concatenate([reg, deg]).astype(np.float32)
    nreg = reg.shape[0]
    N = 40000
    centres = fv[nreg:].astype(np.float64).mean(1)
    pts = np.concatenate([centres[rng.integers(0, len(deg), N // 2)] + rng.normal(size=(N // 2, 3)) * 0.01,
                          rng.normal(size=(N - N // 2, 3)) * 0.06]).astype(np.float32)

    def seg_d2(p, u, v):
        d = v - u
        L2 = (d * d).sum()
        t = np.clip(((p - u) @ d) / L2, 0.0, 1.0) if L2 > 0 else np.zeros(len(p))
        q = u + t[:, None] * d
        return ((p - q) ** 2).sum(1), q

    p64, f64 = pts.astype(np.float64), fv.astype(np.float64)
    od2, _, _, _ = osdf.compute_sdf(torch.tensor(p64[:3000]), torch.tensor(f64[:nreg]))
    best = od2.numpy().copy()
    from_deg = np.zeros(3000, dtype=bool)
    for t in f64[nreg:]:
        for u, v in ((t[0], t[1]), (t[0], t[2]), (t[1], t[2])):
            d2s, _ = seg_d2(p64[:3000], u, v)
            from_deg |= d2s < best
            best = np.minimum(best, d2s)
    assert from_deg.mean() > 0.2  # the degenerate faces do decide a good share of the queries
    pd, fd = torch.tensor(pts, device="cuda"), torch.tensor(fv, device="cuda")
    for route in ("hierarchy", "loop"):
        d2, sg, nrm, cls = gq.ops.compute_sdf(pd, fd) if route == "hierarchy" else torch.ops.graspqp_amd.compute_sdf(pd, fd)
        torch.cuda.synchronize()
        np.testing.assert_allclose(np.sqrt(d2.cpu().numpy()[:3000]), np.sqrt(best), rtol=2e-5, atol=2e-7, err_msg=route)
        far = from_deg & (best > 1e-8)
        assert (sg.cpu().numpy()[:3000][far] == 1).all(), route
        diff = (pd - cls)[:3000].cpu().numpy()
        np.testing.assert_allclose(nrm.cpu().numpy()[:3000][far], diff[far] / np.linalg.norm(diff[far], axis=1, keepdims=True),
                                   atol=1e-4, err_msg=route)
        assert torch.isfinite(d2).all() and torch.isfinite(nrm).all()
    assert (id(fd), "bvh") in gq.ops._MESH_CACHE


def test_compute_sdf_mesh_cache_follows_the_face_verts_tensor(gq):
    """compute_sdf keeps the acceleration data of a mesh (>= 1024 faces) while the caller's face_verts tensor is alive and
    unmodified: a second call reuses it, an in-place change of the tensor rebuilds it, a dead tensor drops its entry; the
    cached route (cluster search) and the uncached one (face loop) return the same distances."""
    import gc

    rng = np.random.default_rng(4)
    fv_np = meshes.superquadric(5, 48, 24)
    assert fv_np.shape[0] >= 1024
    fv = torch.tensor(fv_np, device="cuda")
    pts_np = (rng.normal(size=(4096, 3)) * 0.08).astype(np.float32)
    pts = torch.tensor(pts_np, device="cuda")
    n0 = len(gq.ops._MESH_CACHE)
    d2a, sga, _, cla = gq.ops.compute_sdf(pts, fv)
    assert len(gq.ops._MESH_CACHE) == n0 + 1
    ms = gq.ops._MESH_CACHE[(id(fv), "clusters")][3]
    d2b, _, _, _ = gq.ops.compute_sdf(pts, fv)
    assert gq.ops._MESH_CACHE[(id(fv), "clusters")][3] is ms and torch.equal(d2a, d2b)
    # the uncached face loop (registered op on the raw tensors) agrees on the distance bit for bit
    d2c, sgc, _, _ = torch.ops.graspqp_amd.compute_sdf(pts, fv)
    # (both rank the faces with gq_tri_rank and finish the winner exactly; faces that tie within the ranking noise of
    # ~1e-10 m^2 may swap, which moves a distance by a few ulp at most)
    assert (d2a == d2c).float().mean() > 0.999 and (sga == sgc).float().mean() > 0.999
    np.testing.assert_allclose(d2a.cpu().numpy(), d2c.cpu().numpy(), rtol=1e-5, atol=1e-10)
    od2, _, _, _ = osdf.compute_sdf(torch.tensor(pts_np[:512], dtype=torch.float64), torch.tensor(fv_np, dtype=torch.float64))
    np.testing.assert_allclose(np.sqrt(d2a[:512].cpu().numpy()), np.sqrt(od2.numpy()), rtol=1e-4, atol=2e-7)
    # in-place change of the mesh: rebuilt, and the result follows the new mesh
    fv.mul_(0.5)
    d2s, _, _, _ = gq.ops.compute_sdf(pts, fv)
    assert gq.ops._MESH_CACHE[(id(fv), "clusters")][3] is not ms
    od2s, _, _, _ = osdf.compute_sdf(torch.tensor(pts_np[:512], dtype=torch.float64), torch.tensor(fv_np, dtype=torch.float64) * 0.5)
    np.testing.assert_allclose(np.sqrt(d2s[:512].cpu().numpy()), np.sqrt(od2s.numpy()), rtol=1e-4, atol=2e-7)
    # gradient route unchanged (only dist_sq w.r.t. points)
    pg = pts.clone().requires_grad_()
    d2g, _, _, clg = gq.ops.compute_sdf(pg, fv)
    d2g.sum().backward()
    np.testing.assert_allclose(pg.grad.cpu().numpy(), 2 * (pts - clg).cpu().numpy(), rtol=1e-5, atol=1e-7)
    del fv, ms
    gc.collect()
    assert len(gq.ops._MESH_CACHE) == n0


def test_sdf_backward_and_edge_cases(gq):
    fv = torch.tensor(meshes.box(), device="cuda")
    p = torch.randn(257, 3, device="cuda").mul(0.06).requires_grad_()
    d2, sg, nrm, cls = gq.ops.compute_sdf(p, fv)
    w = torch.randn(257, device="cuda")
    (d2 * w).sum().backward()
    np.testing.assert_allclose(p.grad.cpu().numpy(), (2 * (p.detach() - cls) * w[:, None]).cpu().numpy(), rtol=1e-6, atol=1e-9)
    # empty query set and single query
    e = gq.ops.compute_sdf(torch.zeros(0, 3, device="cuda"), fv)
    assert e[0].shape == (0,) and e[3].shape == (0, 3)
    one = gq.ops.compute_sdf(torch.tensor([[0.0, 0.0, 0.2]], device="cuda"), fv)
    assert abs(float(one[0][0]) - 0.15**2) < 1e-6 and int(one[1][0]) == 1
    with pytest.raises(ValueError):
        gq.ops.compute_sdf(torch.zeros(4, 2, device="cuda"), fv)


# ---------------------------------------------------------------------------------------------------------------
# QP
# ---------------------------------------------------------------------------------------------------------------
def test_qp_reference_known_answer(gq, golden_dir):
    """reference tests/metrics/test_solver.py:5-27 through the SQPLsqSolver-compatible class."""
    from graspqp_amd.metrics import SQPLsqSolver

    g = _load(golden_dir, "kat_solver.npz")
    A, b = torch.tensor(g["A"], device="cuda"), torch.tensor(g["b"], device="cuda")
    solver = SQPLsqSolver.from_mat(A, b)
    sol = solver(A, b, min_bound=-10.0, max_bound=1e3, init=0.1)
    assert torch.allclose(sol, torch.zeros_like(sol), atol=1e-4)


@pytest.mark.parametrize("n,k", [(4, 4), (12, 4), (16, 4), (12, 8)])
def test_qp_iterate_matches_oracle(gq, golden_dir, n, k):
    g = _load(golden_dir, f"span_n{n}_k{k}.npz")
    F = torch.tensor(g["F"], dtype=torch.float64)
    B, _, nz = F.shape
    b0 = torch.zeros(B, 6, dtype=torch.float64)
    val_o, x_o = oqp.lsq_box_qp(F, b0, 1.0, 21.0, box_form=True)
    Fg = F.float().cuda().requires_grad_()
    x = gq.ops.lsq_box_qp(Fg, None, 1.0, 21.0)
    val = 0.5 * ((Fg @ x.unsqueeze(-1)).squeeze(-1) ** 2).sum(-1)
    rel = _rel(2 * (val.detach().cpu().numpy() + 0.01), 2 * (val_o.numpy() + 0.01))
    val_32, _ = oqp.lsq_box_qp(F.float(), b0.float(), 1.0, 21.0, box_form=True)  # the oracle's own fp32 noise on these rows
    assert_tail_within_fp32_noise(rel, _rel(2 * (val_32.double().numpy() + 0.01), 2 * (val_o.numpy() + 0.01)), f"QP value n={n} k={k}")
    assert np.abs(x.detach().cpu().numpy() - x_o.numpy()).max() < 5e-2
    assert (x.min() >= 1.0 - 1e-4) and (x.max() <= 21.0 + 1e-3)
    # gradient of the value w.r.t. F (direct + implicit KKT part)
    val.sum().backward()
    Fo = F.clone().requires_grad_()
    vo, _ = oqp.lsq_box_qp(Fo, b0, 1.0, 21.0, box_form=True)
    vo.sum().backward()
    ge = np.linalg.norm(Fg.grad.cpu().numpy() - Fo.grad.numpy()) / np.linalg.norm(Fo.grad.numpy())
    assert ge < 2e-2, ge


@pytest.mark.parametrize("B", [300, 1500, 3000, 5000])
def test_qp_batch_global_stop_large_batches(gq, golden_dir, B):
    """qpth's batch-global stop rule over large batches (the single-wavefront, 1024-thread and generic stop kernels):
    the fixture's problems tiled to B rows must give every copy the solution of the un-tiled batch, because the rule
    only looks at max / any / min over the rows."""
    g = _load(golden_dir, "span_n4_k4.npz")
    F = torch.tensor(g["F"], dtype=torch.float32).cuda()
    b0 = F.shape[0]
    x0 = gq.ops.lsq_box_qp(F, None, 1.0, 21.0)
    rep = (B + b0 - 1) // b0
    Fb = F.repeat(rep, 1, 1)[:B].contiguous()
    xb = gq.ops.lsq_box_qp(Fb, None, 1.0, 21.0)
    assert torch.equal(xb, x0.repeat(rep, 1)[:B])


def test_qpfunction_level_boundary(gq):
    """QPFunction(Q, p, G, h) with G = [I; -I]; wrong G is refused loudly."""
    from graspqp_amd.metrics import QPFunction

    torch.manual_seed(0)
    B, nz = 8, 16
    M = torch.randn(B, 6, nz, dtype=torch.float64)
    Q = M.transpose(1, 2) @ M + 1e-2 * torch.eye(nz, dtype=torch.float64)
    p = torch.randn(B, nz, dtype=torch.float64)
    G = torch.cat([torch.eye(nz), -torch.eye(nz)]).double()
    h = torch.cat([2 * torch.ones(B, nz), torch.ones(B, nz)], 1).double()
    xo, _, _, nit = oqp.pdipm_forward(Q, p, G, h)
    x = QPFunction(maxIter=12, eps=5e-2)(Q.float().cuda(), p.float().cuda(), G.float().cuda(), h.float().cuda())
    assert np.abs(x.cpu().numpy() - xo.numpy()).max() < 2e-2
    with pytest.raises(NotImplementedError):
        QPFunction()(Q.float().cuda(), p.float().cuda(), (2 * G).float().cuda(), h.float().cuda())


@pytest.mark.parametrize("n,k", [(12, 8), (16, 8), (9, 8)])
def test_qpfunction_dense_hessian_above_64_variables(gq, golden_dir, n, k):
    """qpth.qp.QPFunction(Q, p, G = [I; -I], h) with a dense 72 / 96 / 128-variable Hessian -- what an unchanged
    metrics/solver/qp_solver.py:101-125 hands over with 8-edge friction cones (--n_friction_cone 8).  The matrix lives in LDS
    there (csrc/qp_dense.hip); values and the KKT-implicit gradient against the fp64 oracle, and against the low-rank route
    (SQPLsqSolver) on the same problems."""
    from _scenes import hetero_contacts
    from graspqp_amd.metrics import QPFunction

    nz = n * k
    if n == 12:
        F = torch.tensor(_load(golden_dir, "span_n12_k8.npz")["F"], dtype=torch.float64)
    else:
        pts, nrm, cog = hetero_contacts(24, n, 5)
        F = ospan.grasp_matrix(pts, nrm, cog, 0.2, k)
    B = F.shape[0]
    assert F.shape[2] == nz and nz > 64
    b0 = torch.zeros(B, 6, dtype=torch.float64)
    val_o, x_o = oqp.lsq_box_qp(F, b0, 1.0, 21.0, box_form=True)
    Q = (F.transpose(1, 2) @ F + 1e-4 * torch.eye(nz, dtype=torch.float64)).float().cuda()
    p = torch.zeros(B, nz, device="cuda")
    G = torch.cat([torch.eye(nz), -torch.eye(nz)]).cuda()
    h = torch.cat([21.0 * torch.ones(B, nz), -torch.ones(B, nz)], 1).cuda()
    Qg = Q.clone().requires_grad_()
    x = QPFunction(maxIter=12, eps=5e-2)(Qg, p, G, h)
    assert x.shape == (B, nz) and torch.isfinite(x).all()
    assert (x.min() >= 1.0 - 1e-3) and (x.max() <= 21.0 + 1e-2)
    Fc = F.float().cuda()
    val = 0.5 * ((Fc @ x.detach().unsqueeze(-1)).squeeze(-1) ** 2).sum(-1)
    rel = _rel(2 * (val.cpu().numpy() + 0.01), 2 * (val_o.numpy() + 0.01))
    # an all-fp32 dense factorisation of Q + diag(d), d spanning 1e-4 .. 1e8 (the reference's qpth does the same in fp32 on
    # the 2nz x 2nz form; oracle in that form and precision: p99 7e-4 .. 8e-4, max up to 2e-2, profiles/r03_parity_report.json)
    assert np.median(rel) < 2e-4 and rel.max() < 3e-2, (np.median(rel), rel.max())
    x_lr = gq.ops.lsq_box_qp(Fc, None, 1.0, 21.0)
    v_lr = 0.5 * ((Fc @ x_lr.unsqueeze(-1)).squeeze(-1) ** 2).sum(-1)
    assert np.median(_rel(val.cpu().numpy() + 0.01, v_lr.cpu().numpy() + 0.01)) < 2e-4
    # backward: d (c'x) / dQ = 1/2 (dx x' + x dx') with dx from the KKT system at the solution
    c = torch.linspace(-1.0, 1.0, nz, device="cuda")
    (x * c).sum().backward()
    Qo = (F.transpose(1, 2) @ F + 1e-4 * torch.eye(nz, dtype=torch.float64)).requires_grad_()
    Go = torch.cat([torch.eye(nz, dtype=torch.float64), -torch.eye(nz, dtype=torch.float64)])
    ho = torch.cat([21.0 * torch.ones(B, nz, dtype=torch.float64), -torch.ones(B, nz, dtype=torch.float64)], 1)
    xq = oqp.QPFunction(box_form=True)(Qo, torch.zeros(B, nz, dtype=torch.float64), Go, ho)
    (xq * c.cpu().double()).sum().backward()
    gerr = (Qg.grad.cpu().double() - Qo.grad).norm() / Qo.grad.norm()
    assert gerr < 0.1, gerr  # the gradient goes through the ill-conditioned KKT solve at an fp32 iterate
    assert torch.isfinite(Qg.grad).all() and float(Qg.grad.abs().max()) > 0


@pytest.mark.parametrize("n,k", [(4, 4), (12, 4), (16, 4), (12, 8)])
def test_fc_energy_and_gradient(gq, golden_dir, n, k):
    g = _load(golden_dir, f"span_n{n}_k{k}.npz")
    pts = torch.tensor(g["contact_pts"], dtype=torch.float64)
    nrm = torch.tensor(g["contact_normals"], dtype=torch.float64)
    cog = torch.tensor(g["cog"], dtype=torch.float64)
    po = pts.clone().requires_grad_()
    eo, xso = ospan.e_fc(po, nrm, cog, k=k, box_form=True)
    eo.sum().backward()
    pg = pts.float().cuda().requires_grad_()
    e, xs = gq.ops.fc_energy(pg, nrm.float().cuda(), cog.float().cuda(), n_cone_vecs=k)
    e.sum().backward()
    rel = _rel(e.detach().cpu().numpy(), eo.detach().numpy())
    assert_tail_within_fp32_noise(rel, oracle_fc_fp32_noise(ospan, pts, nrm, cog, k, eo), f"E_fc n={n} k={k}")
    ge = np.linalg.norm(pg.grad.cpu().numpy() - po.grad.numpy()) / np.linalg.norm(po.grad.numpy())
    assert ge < 2e-2, ge
    # the grasp matrix itself is pinned by the reference's own span.py (fixture F)
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF

    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": k})
    e2, _ = fn(contact_pts=pts.float().cuda(), contact_normals=nrm.float().cuda(), sdf=None, cog=cog.float().cuda(),
               with_solution=True, svd_gain=0.1)
    assert torch.allclose(e2, e.detach(), rtol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# kinematics, penetration
# ---------------------------------------------------------------------------------------------------------------
def _rand_pose(spec, B, seed, spread=0.12):
    g = torch.Generator().manual_seed(seed)
    t = torch.nn.functional.normalize(torch.randn(B, 3, generator=g, dtype=torch.float64), dim=-1) * spread
    six = torch.randn(B, 6, generator=g, dtype=torch.float64)
    th = torch.tensor(spec.default_state, dtype=torch.float64)[None] + 0.3 * torch.randn(B, spec.n_dofs, generator=g, dtype=torch.float64)
    return torch.cat([t, six, th], 1)


@pytest.mark.parametrize("hand_name", ["allegro", "shadow_hand", "robotiq3", "ability_hand", "panda"])
def test_fk_contacts_forward_backward(gq, hand_name):
    spec = get_hand_spec(hand_name)
    B, n = 6, 12
    hp = _rand_pose(spec, B, 3)
    idx = torch.randint(spec.n_contact_candidates, (B, n), generator=torch.Generator().manual_seed(4))
    oh = omodels.OracleHand(spec, torch.float64)
    hpo = hp.clone().requires_grad_()
    oh.set_parameters(hpo, idx)
    sph = okin.sphere_centers_world(spec, oh.current_status, oh.global_rotation, oh.global_translation)
    hand = gq.ops.HandHandle(spec)
    hpg = hp.float().cuda().requires_grad_()
    Rg, LT, cp, cn, sc, _ = gq.ops.fk_contacts(hpg, idx.cuda(), hand)
    np.testing.assert_allclose(Rg.detach().cpu().numpy(), oh.global_rotation.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(LT.detach().cpu().numpy(), oh.current_status.detach().numpy()[:, :, :3, :], atol=3e-6)
    np.testing.assert_allclose(cp.detach().cpu().numpy(), oh.contact_points.detach().numpy(), atol=3e-6)
    np.testing.assert_allclose(cn.detach().cpu().numpy(), oh.contact_normals.detach().numpy(), atol=3e-6)
    np.testing.assert_allclose(sc.detach().cpu().numpy(), sph.detach().numpy(), atol=3e-6)
    g = torch.Generator().manual_seed(9)
    w1, w2, w3 = (torch.randn(*s, generator=g, dtype=torch.float64) for s in ((B, n, 3), (B, n, 3), tuple(sph.shape)))
    w4 = torch.randn(B, 3, 3, generator=g, dtype=torch.float64)
    ((oh.contact_points * w1).sum() + (oh.contact_normals * w2).sum() + (sph * w3).sum() + (oh.global_rotation * w4).sum()).backward()
    ((cp * w1.float().cuda()).sum() + (cn * w2.float().cuda()).sum() + (sc * w3.float().cuda()).sum() + (Rg * w4.float().cuda()).sum()).backward()
    go, gg = oh.hand_pose.grad.numpy(), hpg.grad.cpu().numpy()
    assert np.linalg.norm(gg - go) / np.linalg.norm(go) < 1e-4
    np.testing.assert_allclose(gg, go, rtol=2e-3, atol=2e-4)


def test_hand_penetration_and_self_penetration(gq):
    spec = get_hand_spec("allegro")
    n_obj, be, P = 2, 3, 300
    B = n_obj * be
    fvs = [meshes.icosphere(2, 0.05), meshes.superquadric(5, 24, 12)]
    sps = [meshes.surface_points(f, P, oversample=4) for f in fvs]
    hp = _rand_pose(spec, B, 11, spread=0.03)  # hand inside / around the object -> many penetrating points
    hp[:, 9:] += 0.4  # curl the fingers -> self penetration
    idx = torch.randint(spec.n_contact_candidates, (B, 4), generator=torch.Generator().manual_seed(2))
    oh = omodels.OracleHand(spec, torch.float64)
    oo = omodels.OracleObject(fvs, sps, be, torch.float64)
    hpo = hp.clone().requires_grad_()
    oh.set_parameters(hpo, idx)
    dis_o = oh.cal_distance(oo.surface_points_tensor)
    spen_o = oh.self_penetration()
    e_o = torch.where(dis_o <= 0, torch.zeros_like(dis_o), dis_o).sum(-1)
    (e_o.sum() + spen_o.sum()).backward()

    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel

    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=be, num_samples=P)
    om.initialize_from_meshes(fvs, surface_points_list=sps)
    hpg = hp.float().cuda().requires_grad_()
    hm.set_parameters(hpg, idx.cuda())
    dis = hm.cal_distance(om.surface_points_each)
    spen = hm.self_penetration()
    e = torch.where(dis <= 0, torch.zeros_like(dis), dis).sum(-1)
    (e.sum() + spen.sum()).backward()
    d, do = dis.detach().cpu().numpy(), dis_o.detach().numpy()
    assert (do > 1e-4).sum() > 20, "test scene must contain penetrating points"
    big = np.abs(d - do) > 2e-6
    assert big.mean() < 2e-3, f"distance disagreement rate {big.mean()}"
    np.testing.assert_allclose(e.detach().cpu().numpy(), e_o.detach().numpy(), rtol=1e-4, atol=1e-6)
    assert float(spen_o.detach().sum()) > 0, "test scene must self-penetrate"
    np.testing.assert_allclose(spen.detach().cpu().numpy(), spen_o.detach().numpy(), rtol=1e-4, atol=1e-7)
    go, gg = oh.hand_pose.grad.numpy(), hm.hand_pose.grad.cpu().numpy()
    assert np.linalg.norm(gg - go) / np.linalg.norm(go) < 2e-3
    # the reference-style call with the row-expanded (B,P,3) tensor gives the same distances
    om.attach(hm)
    dis2 = hm.cal_distance(om.surface_points_tensor)
    assert torch.equal(dis2, dis)
    # penetration-only mode (what E_pen uses): same values where dis > 0 (two template instantiations of one
    # kernel: FMA contraction may differ in the last bit), non-positive elsewhere
    pos = dis > 1e-6
    for mode in (1, 3, 2):  # 1 = voxel candidate lists, 3 = occupancy grid + queues, 2 = AABB culling only
        dis3 = hm.cal_distance(om.surface_points_each, penetration_only=mode)
        # ranking distances carry ~1e-10 m^2 of round-off -> near-tied faces may swap: 3e-6 m on the distance
        torch.testing.assert_close(dis3[pos], dis[pos], rtol=2e-4, atol=3e-6)
        assert (dis3[dis <= -1e-6] <= 0).all()
    # gradients through the compacted kernel == through the exact kernel
    grads = []
    for mode in (1, 0):
        hm.set_parameters(hp.float().cuda().requires_grad_(), idx.cuda())
        torch.relu(hm.cal_distance(om.surface_points_each, penetration_only=mode)).sum().backward()
        grads.append(hm.hand_pose.grad.clone())
    assert (grads[0] - grads[1]).norm() <= 5e-3 * grads[1].norm()


# ---------------------------------------------------------------------------------------------------------------
# whole energy / whole iteration against the golden fixtures (reference energy.py / optimizer.py outputs)
# ---------------------------------------------------------------------------------------------------------------
def _stepper_from_fixture(gq, g, n_contact, **kw):
    spec = get_hand_spec("allegro")
    n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
    fvs = [g[f"obj{i}_face_verts"] for i in range(n_obj)]
    sps = np.stack([g[f"obj{i}_surface_points"] for i in range(n_obj)])
    hand = gq.ops.HandHandle(spec)
    return gq.stepper.GraspStepper(hand, gq.ops.MeshSet(fvs), torch.tensor(sps), be, n_contact, **kw)


@pytest.mark.parametrize("tag,n", [("allegro_sphere_b4_n4", 4), ("allegro_sq_b6_n12", 12)])
def test_energy_and_gradient_match_golden(gq, golden_dir, tag, n):
    g = _load(golden_dir, f"energy_{tag}.npz")
    st = _stepper_from_fixture(gq, g, n)
    terms, total, grad = st.evaluate(torch.tensor(g["hand_pose"], dtype=torch.float32).cuda(), torch.tensor(g["contact_idx"]).cuda())
    torch.cuda.synchronize()
    for k in ("E_dis", "E_joints", "E_pen", "E_spen"):
        np.testing.assert_allclose(terms[k].cpu().numpy(), g[k], rtol=2e-4, atol=2e-6, err_msg=k)
    # E_fc and the total: within 2x the error the oracle itself makes in fp32 on this very scene (fixture values are fp64)
    n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
    oh32 = omodels.OracleHand(get_hand_spec("allegro"), torch.float32)
    oo32 = omodels.OracleObject([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                                [g[f"obj{i}_surface_points"] for i in range(n_obj)], be, torch.float32)
    oh32.set_parameters(torch.tensor(g["hand_pose"], dtype=torch.float32), torch.tensor(g["contact_idx"]))
    lo32 = ref_cpu.calculate_energy(oh32, oo32, box_form=True)
    assert_tail_within_fp32_noise(_rel(terms["E_fc"].cpu().numpy(), g["E_fc"]), _rel(lo32["E_fc"].numpy(), g["E_fc"]), "E_fc")
    rel = _rel(total.cpu().numpy(), g["total"])
    assert_tail_within_fp32_noise(rel, _rel(ref_cpu.total_energy(lo32).numpy(), g["total"]), "total energy")
    gg, go = grad.cpu().numpy(), g["grad"]
    assert np.linalg.norm(gg - go) / np.linalg.norm(go) < 5e-3
    # autograd route == fused stepper (same kernels)
    from graspqp_amd.core.energy import calculate_energy
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF

    spec = get_hand_spec("allegro")
    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=int(g["batch_size_each"]), num_samples=g["obj0_surface_points"].shape[0])
    n_obj = int(g["n_obj"])
    om.initialize_from_meshes([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                              surface_points_list=[g[f"obj{i}_surface_points"] for i in range(n_obj)])
    hp = torch.tensor(g["hand_pose"], dtype=torch.float32).cuda().requires_grad_()
    hm.set_parameters(hp, torch.tensor(g["contact_idx"]).cuda())
    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
    losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=["E_dis", "E_fc", "E_pen", "E_spen", "E_joints"], svd_gain=0.1)
    w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    tot = sum(w[k] * v for k, v in losses.items())
    tot.sum().backward()
    np.testing.assert_allclose(tot.detach().cpu().numpy(), total.cpu().numpy(), rtol=2e-5)
    ga = hm.hand_pose.grad.cpu().numpy()
    assert np.linalg.norm(ga - gg) / np.linalg.norm(gg) < 1e-4


def test_eager_route_equals_the_registered_ops(gq, golden_dir):
    """The class surface calls the forward bodies / backwards of the registered ops directly (ops._Eager: no dispatcher
    round trip per call); ``ops.use_dispatcher(True)`` sends the same calls through ``torch.ops.graspqp_amd.*``.  One
    energy + gradient evaluation of the fixture scene on both routes, plus compute_sdf and QPFunction with gradients: the
    same kernels on the same inputs, bit for bit."""
    from graspqp_amd.core.energy import calculate_energy
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF
    from graspqp_amd.metrics import QPFunction

    g = _load(golden_dir, "energy_allegro_sq_b6_n12.npz")
    pose_key, idx_key = "hand_pose", "contact_idx"
    n_obj = int(g["n_obj"])
    spec = get_hand_spec("allegro")
    names = ["E_dis", "E_fc", "E_pen", "E_spen", "E_joints"]
    w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    rng = np.random.default_rng(2)
    fv = torch.tensor(meshes.icosphere(2, 0.05), dtype=torch.float32).cuda()
    pts0 = torch.tensor(rng.normal(size=(500, 3)) * 0.06, dtype=torch.float32).cuda()
    A = torch.tensor(rng.normal(size=(16, 6, 24)), dtype=torch.float32).cuda()
    res = {}
    for route in (False, True):
        old = gq.ops.use_dispatcher(route)
        try:
            hm = HandModel(spec, "cuda")
            om = ObjectModel(batch_size_each=int(g["batch_size_each"]), num_samples=g["obj0_surface_points"].shape[0])
            om.initialize_from_meshes([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                                      surface_points_list=[g[f"obj{i}_surface_points"] for i in range(n_obj)])
            hp = torch.tensor(g[pose_key], dtype=torch.float32).cuda().requires_grad_()
            hm.set_parameters(hp, torch.tensor(g[idx_key]).cuda())
            fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
            losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=names, svd_gain=0.1)
            tot = sum(w[k] * v for k, v in losses.items())
            tot.sum().backward()
            pts = pts0.clone().requires_grad_()
            d2, sg, nrm, cls = gq.ops.compute_sdf(pts, fv)
            d2.sum().backward()
            Aq = A.clone().requires_grad_()
            Q = Aq.transpose(1, 2) @ Aq + 1e-4 * torch.eye(24, device="cuda")
            pq = -(Aq.transpose(1, 2) @ torch.ones(16, 6, 1, device="cuda")).squeeze(-1)
            G = torch.cat([torch.eye(24), -torch.eye(24)]).cuda().expand(16, -1, -1)
            hq = torch.cat([torch.full((16, 24), 2.0), torch.zeros(16, 24)], 1).cuda()
            x = QPFunction(maxIter=12, eps=5e-2)(Q, pq, G, hq)
            x.square().sum().backward()
            torch.cuda.synchronize()
            res[route] = [tot.detach(), hm.hand_pose.grad.clone(), *[losses[k].detach() for k in names], d2.detach(), sg, nrm,
                          cls, pts.grad.clone(), x.detach(), Aq.grad.clone()]
        finally:
            gq.ops.use_dispatcher(old)
    for i, (a, b) in enumerate(zip(res[False], res[True])):
        if i == 1:  # d total / d hand_pose: the eager route sums the terms' contributions inside ONE autograd node
            assert float((a - b).norm() / b.norm()) < 2e-6  # (core/energy.py:_FusedTerms), autograd in another order
        else:
            assert torch.equal(a, b), i
    # the fused node against the term-by-term composition on the eager route: same terms, bit for bit
    from graspqp_amd.core import energy as energy_mod

    outs = []
    for fused in (True, False):
        old, energy_mod.FUSED = energy_mod.FUSED, fused
        try:
            hm = HandModel(spec, "cuda")
            om = ObjectModel(batch_size_each=int(g["batch_size_each"]), num_samples=g["obj0_surface_points"].shape[0])
            om.initialize_from_meshes([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                                      surface_points_list=[g[f"obj{i}_surface_points"] for i in range(n_obj)])
            hm.set_parameters(torch.tensor(g[pose_key], dtype=torch.float32).cuda().requires_grad_(), torch.tensor(g[idx_key]).cuda())
            fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
            up = torch.linspace(0.5, 1.5, hm.hand_pose.shape[0], device="cuda")  # a non-uniform upstream per row and term
            losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=names, svd_gain=0.1)
            sum((j + 1.0) * (losses[k] * up).sum() for j, k in enumerate(names)).backward()
            outs.append([losses[k].detach() for k in names] + [hm.hand_pose.grad.clone()])
        finally:
            energy_mod.FUSED = old
    for a, b in zip(outs[0][:-1], outs[1][:-1]):
        assert torch.equal(a, b)
    assert float((outs[0][-1] - outs[1][-1]).norm() / outs[1][-1].norm()) < 2e-6


def test_mala_iterations_match_reference_optimizer(gq, golden_dir):
    """fit.py loop order + MalaStar semantics against the fixture produced by the reference's own optimizer.py.

    Teacher-forced: before every iteration the stepper state is set to the fixture's accepted state, then ONE
    iteration runs on the GPU with the recorded random draws.  (A free-running comparison is dominated by the
    chaotic sensitivity of the trajectory: d E / d pose ~ 3e3, so fp32 round-off in one proposal moves later
    energies by O(1); that is covered, with matching tolerances, by test_mala_free_running.)"""
    g = _load(golden_dir, "mala_allegro_sphere_b8_n4.npz")
    st = _stepper_from_fixture(gq, g, 4)
    f32 = lambda k: torch.tensor(g[k], dtype=torch.float32).cuda()
    st.reset(f32("hand_pose0"), torch.tensor(g["contact_idx0"]).cuda())
    np.testing.assert_allclose(st.energy.cpu().numpy(), g["energy0"], rtol=2e-4)
    for s in range(1, int(g["n_steps"]) + 1):
        if s > 1:
            st.hand_pose.copy_(f32(f"s{s-1}_hand_pose"))
            st.contact_idx.copy_(torch.tensor(g[f"s{s-1}_contact_idx"]).cuda())
            st.grad.copy_(f32(f"s{s-1}_grad"))
            st.energy.copy_(f32(f"s{s-1}_energy"))
            st.ema.copy_(f32(f"s{s-1}_ema"))
            st.step_count.fill_(s - 1)
        st.step(draws=(f32(f"s{s}_u_switch"), torch.tensor(g[f"s{s}_new_idx"]).cuda(), f32(f"s{s}_u_accept")))
        torch.cuda.synchronize()
        np.testing.assert_allclose(st.s_out.cpu().numpy(), g[f"s{s}_step_size"], rtol=1e-5)
        np.testing.assert_allclose(st.z.cpu().numpy(), g[f"s{s}_z"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(st.pose_new.cpu().numpy(), g[f"s{s}_prop_pose"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(st.ema.cpu().numpy(), g[f"s{s}_ema"], rtol=1e-4, atol=1e-7)
        rel = _rel(st.total_new.cpu().numpy(), g[f"s{s}_new_energy"])
        assert rel.max() < 3e-4, rel  # pose carries fp32 round-off (1e-7 * dE/dpose 3e3) on top of the 1e-4 contract
        np.testing.assert_allclose(st.temperature.cpu().numpy(), g[f"s{s}_temperature"], rtol=1e-4)
        assert st.accept.cpu().bool().tolist() == g[f"s{s}_accept"].tolist()
        np.testing.assert_allclose(st.hand_pose.cpu().numpy(), g[f"s{s}_hand_pose"], rtol=1e-5, atol=2e-6)
        assert st.contact_idx.cpu().tolist() == g[f"s{s}_contact_idx"].tolist()
        assert _rel(st.energy.cpu().numpy(), g[f"s{s}_energy"]).max() < 3e-4
        gref = g[f"s{s}_grad"]
        assert np.linalg.norm(st.grad.cpu().numpy() - gref) <= 2e-2 * np.linalg.norm(gref)


def test_mala_free_running(gq, golden_dir):
    """Free-running 5 iterations from the fixture's initial state: accept decisions identical, poses within
    0.3 * step_size, energies within |dE/dpose| * that."""
    g = _load(golden_dir, "mala_allegro_sphere_b8_n4.npz")
    st = _stepper_from_fixture(gq, g, 4)
    f32 = lambda k: torch.tensor(g[k], dtype=torch.float32).cuda()
    st.reset(f32("hand_pose0"), torch.tensor(g["contact_idx0"]).cuda())
    for s in range(1, int(g["n_steps"]) + 1):
        st.step(draws=(f32(f"s{s}_u_switch"), torch.tensor(g[f"s{s}_new_idx"]).cuda(), f32(f"s{s}_u_accept")))
        torch.cuda.synchronize()
        assert st.accept.cpu().bool().tolist() == g[f"s{s}_accept"].tolist()
        # free running: fp32-vs-fp64 differences of ~1e-5 in E_pen (surface points within round-off of a link face)
        # are amplified step by step -- the exact brute-force query (penetration_only=0) drifts by 3e-3 after five
        # iterations as well (tools/diag_free.py); the teacher-forced test above is the tight one
        np.testing.assert_allclose(st.hand_pose.cpu().numpy(), g[f"s{s}_hand_pose"], rtol=1e-3, atol=5e-3)
        assert st.contact_idx.cpu().tolist() == g[f"s{s}_contact_idx"].tolist()
        np.testing.assert_allclose(st.energy.cpu().numpy(), g[f"s{s}_energy"], rtol=1e-1)


def test_mala_class_surface(gq, golden_dir):
    """The MalaStar / HandModel / calculate_energy mirror runs a fit.py-style loop and matches the fused stepper."""
    from graspqp_amd.core.energy import calculate_energy
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel
    from graspqp_amd.core.optimizer import MalaStar
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF

    g = _load(golden_dir, "mala_allegro_sphere_b8_n4.npz")
    be, n_obj = int(g["batch_size_each"]), int(g["n_obj"])
    spec = get_hand_spec("allegro")
    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=be, num_samples=g["obj0_surface_points"].shape[0])
    om.initialize_from_meshes([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                              surface_points_list=[g[f"obj{i}_surface_points"] for i in range(n_obj)])
    f32 = lambda k: torch.tensor(g[k], dtype=torch.float32).cuda()
    hm.set_parameters(f32("hand_pose0").requires_grad_(), torch.tensor(g["contact_idx0"]).cuda())
    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
    w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    names = list(w)

    def total():
        losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=names, svd_gain=0.1)
        return sum(w[k] * v for k, v in losses.items())

    opt = MalaStar(hm, switch_possibility=0.4, device="cuda", batch_size=be)
    energy = total()
    energy.sum().backward()
    opt.zero_grad()
    energy = energy.detach().clone()
    for s in range(1, 4):
        opt.try_step(draws=(f32(f"s{s}_u_switch"), torch.tensor(g[f"s{s}_new_idx"]).cuda()))
        eb = energy.view(-1, be)
        z = ((eb - eb.mean(-1, keepdim=True)) / eb.std(-1, keepdim=True)).view(-1)
        opt.zero_grad()
        new_energy = total()
        new_energy.sum().backward()
        with torch.no_grad():
            accept, T = opt.accept_step(energy, new_energy, None, z, 1.0, u_accept=f32(f"s{s}_u_accept"))
        assert accept.cpu().tolist() == g[f"s{s}_accept"].tolist()
        np.testing.assert_allclose(hm.hand_pose.detach().cpu().numpy(), g[f"s{s}_hand_pose"], rtol=1e-3, atol=1.5e-3)
        np.testing.assert_allclose(energy.cpu().numpy(), g[f"s{s}_energy"], rtol=2e-2)


# ---------------------------------------------------------------------------------------------------------------
# full-size properties (BASELINE config 2: Allegro, 1 mesh, batch 256, 12 contacts)
# ---------------------------------------------------------------------------------------------------------------
def test_config2_properties_and_determinism(gq):
    spec = get_hand_spec("allegro")
    fv = meshes.superquadric(0)
    sp = meshes.surface_points(fv, 2500, oversample=4)
    hand = gq.ops.HandHandle(spec)
    ms = gq.ops.MeshSet([fv])
    B, n = 256, 12
    hp = _rand_pose(spec, B, 21, spread=0.14).float().cuda()
    idx = torch.randint(spec.n_contact_candidates, (B, n), generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for rep in range(3):
        st = gq.stepper.GraspStepper(hand, ms, torch.tensor(sp)[None], B, n, seed=7)
        st.reset(hp, idx)
        if rep == 2:
            st.capture()  # hipGraph with the three branches in parallel: same kernels, same results
        for _ in range(3):
            st.step()
        torch.cuda.synchronize()
        outs.append((st.energy.clone(), st.hand_pose.clone(), st.grad.clone(), st.terms.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "iteration must be bitwise reproducible"
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[2])), "hipGraph replay must equal the eager launches"
    e, pose, grad, terms = outs[0]
    assert torch.isfinite(e).all() and torch.isfinite(grad).all()
    assert (terms[2] >= 0).all() and (terms[3] >= 0).all() and (terms[4] >= 0).all() and (terms[0] >= 0).all()
    # a sample of rows against the oracle (fp64) on the accepted state
    rows = [0, 17, 101, 255]
    oh = omodels.OracleHand(spec, torch.float64)
    oo = omodels.OracleObject([fv], [sp], len(rows), torch.float64)
    hpo = pose[rows].cpu().double().requires_grad_()
    st2 = gq.stepper.GraspStepper(hand, ms, torch.tensor(sp)[None], B, n)
    t2, tot2, g2 = st2.evaluate(pose, st.contact_idx)
    oh.set_parameters(hpo, st.contact_idx[rows].cpu())
    # batch-global QP stopping: the oracle sees 4 rows, the GPU 256 -> compare everything but E_fc tightly
    lo = ref_cpu.calculate_energy(oh, oo, box_form=True)
    for k, i in (("E_dis", 0), ("E_pen", 2), ("E_spen", 3), ("E_joints", 4)):
        np.testing.assert_allclose(t2[k][rows].cpu().numpy(), lo[k].detach().numpy(), rtol=3e-4, atol=3e-6, err_msg=k)
    np.testing.assert_allclose(t2["E_fc"][rows].cpu().numpy(), lo["E_fc"].detach().numpy(), rtol=0.3)
    # gradient of everything but E_fc (whose stop rule depends on the batch) for the sampled rows, norm-wise
    w0 = {"E_dis": 100.0, "E_fc": 0.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    st3 = gq.stepper.GraspStepper(hand, ms, torch.tensor(sp)[None], B, n, weights=w0)
    _, _, g3 = st3.evaluate(pose, st.contact_idx)
    sum(w0[k] * lo[k] for k in w0 if w0[k] != 0.0).sum().backward()
    go = hpo.grad.numpy()
    gerr = np.linalg.norm(g3[rows].cpu().numpy() - go) / np.linalg.norm(go)
    assert gerr < 1e-3, gerr
    # E_fc at the FULL batch: the oracle's force-closure metric (fp64) on the 256 x 12 contact points / object normals
    # the GPU evaluated -- identical batch composition, so qpth's batch-global stop rule sees the same rows
    eo, _ = ospan.e_fc(st2.cpts.cpu().double(), st2.obj_normal.cpu().double(), st2.cog.cpu().double(), k=4, box_form=True)
    rel = _rel(t2["E_fc"].cpu().numpy(), eo.numpy())
    assert_tail_within_fp32_noise(rel, oracle_fc_fp32_noise(ospan, st2.cpts.cpu(), st2.obj_normal.cpu(), st2.cog.cpu(), 4, eo),
                                  "E_fc, 256 rows")
    # ... and its gradient w.r.t. the contact points (KKT-implicit QP backward + direct + singular-value parts), full batch
    st4 = gq.stepper.GraspStepper(hand, ms, torch.tensor(sp)[None], B, n,
                                  weights={"E_dis": 0.0, "E_fc": 1.0, "E_pen": 0.0, "E_spen": 0.0, "E_joints": 0.0})
    st4.evaluate(pose, st.contact_idx)
    po = st4.cpts.cpu().double().requires_grad_()
    e4, _ = ospan.e_fc(po, st4.obj_normal.cpu().double(), st4.cog.cpu().double(), k=4, box_form=True)
    e4.sum().backward()
    gfc = np.linalg.norm(st4.g_cpts.cpu().numpy() - po.grad.numpy()) / np.linalg.norm(po.grad.numpy())
    assert gfc < 2e-2, gfc


# ---------------------------------------------------------------------------------------------------------------
# the other BASELINE configurations in small: shadow hand / 16 contacts (nz = 64), robotiq3 / 8-edge cones (nz = 96,
# two QP columns per lane), several objects per process, and a batch large enough for the stand-alone stop-rule launch
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hand_name,n,k,n_obj,be", [("shadow_hand", 16, 4, 2, 40), ("robotiq3", 12, 8, 1, 24),
                                                    ("allegro", 12, 4, 3, 100),
                                                    ("allegro", 12, 8, 2, 32)])  # the Allegro half of BASELINE configs[4], small
def test_stepper_other_configs(gq, hand_name, n, k, n_obj, be):
    from graspqp_amd.core.energy import calculate_energy
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF

    DEFAULT_W = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    spec = get_hand_spec(hand_name)
    fvs = [meshes.superquadric(5 + i, n_u=32, n_v=16) for i in range(n_obj)]
    sps = [meshes.surface_points(f, 600, oversample=4, seed=3 + i) for i, f in enumerate(fvs)]
    B = n_obj * be
    hand = gq.ops.HandHandle(spec)
    ms = gq.ops.MeshSet(fvs)
    hp = _rand_pose(spec, B, 31, spread=0.1).float().cuda()
    idx = torch.randint(spec.n_contact_candidates, (B, n), generator=torch.Generator().manual_seed(2)).cuda()
    fc_cfg = {"n_cone_vecs": k}
    st = gq.stepper.GraspStepper(hand, ms, torch.tensor(np.stack(sps)), be, n, fc_cfg=fc_cfg, seed=5)
    terms, total, grad = st.evaluate(hp, idx)
    # (1) the fused launches == the autograd route built from the C-ABI building blocks
    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=be, num_samples=600)
    om.initialize_from_meshes(fvs, surface_points_list=sps)
    hpa = hp.clone().requires_grad_()
    hm.set_parameters(hpa, idx)
    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": k})
    losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=list(DEFAULT_W), svd_gain=0.1)
    tot = sum(DEFAULT_W[kk] * v for kk, v in losses.items())
    tot.sum().backward()
    for kk in DEFAULT_W:
        np.testing.assert_allclose(terms[kk].cpu().numpy(), losses[kk].detach().cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=kk)
    np.testing.assert_allclose(total.cpu().numpy(), tot.detach().cpu().numpy(), rtol=2e-4)
    ga, gg = hm.hand_pose.grad, grad
    assert (ga - gg).norm() <= 2e-3 * gg.norm()
    # (2) a sample of rows against the fp64 oracle (E_fc loosely: the oracle's stop rule sees 3 rows, the GPU all)
    rows = [0, B // 2, B - 1]
    oh = omodels.OracleHand(spec, torch.float64)
    objs = [r // be for r in rows]
    for r, o in zip(rows, objs):
        oo = omodels.OracleObject([fvs[o]], [sps[o]], 1, torch.float64)
        oh.set_parameters(hp[r : r + 1].cpu().double(), idx[r : r + 1].cpu())
        lo = ref_cpu.calculate_energy(oh, oo, box_form=True, k=k)
        for kk in ("E_dis", "E_pen", "E_spen", "E_joints"):
            np.testing.assert_allclose(terms[kk][r].item(), lo[kk].item(), rtol=3e-4, atol=3e-6, err_msg=f"{kk} row {r}")
        np.testing.assert_allclose(terms["E_fc"][r].item(), lo["E_fc"].item(), rtol=0.3)
    # E_fc tightly: the oracle's metric on the whole batch of contact points the GPU evaluated (same stop-rule input)
    eo, _ = ospan.e_fc(st.cpts.cpu().double(), st.obj_normal.cpu().double(), st.cog.cpu().double(), k=k, box_form=True)
    rel = _rel(terms["E_fc"].cpu().numpy(), eo.numpy())
    assert_tail_within_fp32_noise(rel, oracle_fc_fp32_noise(ospan, st.cpts.cpu(), st.obj_normal.cpu(), st.cog.cpu(), k, eo),
                                  f"E_fc {hand_name} n={n} k={k}")
    # (3) iterations run (graph replay == eager), stay finite and are reproducible
    outs = []
    for rep in range(2):
        s2 = gq.stepper.GraspStepper(hand, ms, torch.tensor(np.stack(sps)), be, n, fc_cfg=fc_cfg, seed=5)
        s2.reset(hp, idx)
        if rep == 1:
            s2.capture(iters=2)
        for _ in range(4):
            s2.step()
        s2.flush()
        torch.cuda.synchronize()
        outs.append((s2.energy.clone(), s2.hand_pose.clone(), s2.contact_idx.clone()))
    assert torch.isfinite(outs[0][0]).all()
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))


def test_stepper_reset_iteration_matches_class_surface(gq, golden_dir):
    """fit.py:408-422 -- an iteration with re-initialised rows: GraspStepper.step_reset == the MalaStar / HandModel route
    (try_step -> overwrite the masked rows -> reset_envs -> energy -> accept_step(reset_mask)), with graph-replayed
    iterations before and after it."""
    from graspqp_amd.core.energy import calculate_energy
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel
    from graspqp_amd.core.optimizer import MalaStar
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF

    g = _load(golden_dir, "mala_allegro_sphere_b8_n4.npz")
    be, n_obj = int(g["batch_size_each"]), int(g["n_obj"])
    B = be * n_obj
    spec = get_hand_spec("allegro")
    f32 = lambda k: torch.tensor(g[k], dtype=torch.float32).cuda()
    draws = lambda s: (f32(f"s{s}_u_switch"), torch.tensor(g[f"s{s}_new_idx"]).cuda(), f32(f"s{s}_u_accept"))
    mask = torch.zeros(B, dtype=torch.bool)
    mask[[1, B - 2]] = True
    new_pose = f32("hand_pose0").roll(3, 0)  # "fresh initial poses": other rows' start poses
    new_idx = torch.tensor(g["contact_idx0"]).cuda().roll(3, 0)
    # --- stepper: graph iteration, reset iteration, graph iteration
    st = _stepper_from_fixture(gq, g, 4)
    st.reset(f32("hand_pose0"), torch.tensor(g["contact_idx0"]).cuda())
    st.capture()
    st.step(draws=draws(1))
    st.step_reset(mask, new_pose, new_idx, draws=draws(2))
    assert st.accept.cpu().bool()[mask].all(), "re-initialised rows are accepted unconditionally"
    st.step(draws=draws(3))
    torch.cuda.synchronize()
    # --- class-surface route
    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=be, num_samples=g["obj0_surface_points"].shape[0])
    om.initialize_from_meshes([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                              surface_points_list=[g[f"obj{i}_surface_points"] for i in range(n_obj)])
    hm.set_parameters(f32("hand_pose0").requires_grad_(), torch.tensor(g["contact_idx0"]).cuda())
    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
    w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}

    def total():
        losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=list(w), svd_gain=0.1)
        return sum(w[k] * v for k, v in losses.items())

    opt = MalaStar(hm, switch_possibility=0.4, device="cuda", batch_size=be)
    energy = total()
    energy.sum().backward()
    opt.zero_grad()
    energy = energy.detach().clone()
    for s in (1, 2, 3):
        d = draws(s)
        opt.try_step(draws=d[:2])
        eb = energy.view(-1, be)
        z = ((eb - eb.mean(-1, keepdim=True)) / eb.std(-1, keepdim=True)).view(-1)
        rm = None
        if s == 2:
            rm = mask.cuda()
            # what initialize_convex_hull does last (initializations.py:186-193): full-size fresh pose / indices + env_mask
            hm.set_parameters(new_pose.clone().requires_grad_(), new_idx, env_mask=rm)
            opt.reset_envs(rm)
        opt.zero_grad()
        new_energy = total()
        new_energy.sum().backward()
        with torch.no_grad():
            opt.accept_step(energy, new_energy, rm, z, 1.0, u_accept=d[2])
    np.testing.assert_allclose(st.energy.cpu().numpy(), energy.cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(st.hand_pose.cpu().numpy(), hm.hand_pose.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(st.contact_idx, hm.contact_point_indices)


def _force_state(st, g, p, f32):
    """Teacher forcing: put the stepper into the fixture's accepted state after iteration `p`."""
    st.hand_pose.copy_(f32(f"{p}_hand_pose"))
    st.contact_idx.copy_(torch.tensor(g[f"{p}_contact_idx"]).cuda())
    st.grad.copy_(f32(f"{p}_grad"))
    st.energy.copy_(f32(f"{p}_energy"))
    st.ema.copy_(f32(f"{p}_ema"))
    st.step_count.copy_(torch.tensor(g[f"{p}_step"]).cuda())


def _check_iteration(st, g, p, check_grad_rows=None):
    np.testing.assert_allclose(st.s_out.cpu().numpy(), g[f"{p}_step_size"], rtol=1e-5)
    np.testing.assert_allclose(st.z.cpu().numpy(), g[f"{p}_z"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(st.pose_new.cpu().numpy(), g[f"{p}_prop_pose"], rtol=1e-5, atol=2e-6)
    assert st.idx_new.cpu().tolist() == g[f"{p}_prop_idx"].tolist()
    np.testing.assert_allclose(st.ema.cpu().numpy(), g[f"{p}_ema"], rtol=1e-4, atol=1e-7)
    assert _rel(st.total_new.cpu().numpy(), g[f"{p}_new_energy"]).max() < 3e-4
    np.testing.assert_allclose(st.temperature.cpu().numpy(), g[f"{p}_temperature"], rtol=1e-4)
    assert st.accept.cpu().bool().tolist() == g[f"{p}_accept"].tolist()
    np.testing.assert_allclose(st.hand_pose.cpu().numpy(), g[f"{p}_hand_pose"], rtol=1e-5, atol=2e-6)
    assert st.contact_idx.cpu().tolist() == g[f"{p}_contact_idx"].tolist()
    assert _rel(st.energy.cpu().numpy(), g[f"{p}_energy"]).max() < 3e-4
    assert st.step_count.cpu().tolist() == g[f"{p}_step"].tolist()
    gref, gg = g[f"{p}_grad"], st.grad.cpu().numpy()
    assert np.linalg.norm(gg - gref) <= 2e-2 * np.linalg.norm(gref)
    for r in check_grad_rows or []:
        assert np.linalg.norm(gg[r] - gref[r]) <= 2e-2 * np.linalg.norm(gref[r]) + 1e-6, r


def test_mala_reset_iteration_and_decays_match_reference_optimizer(gq, golden_dir):
    """Fixture produced by the reference's own optimizer.py started at step 149 (both decay exponents non-zero and
    changing) with a re-initialisation iteration in the middle: MalaStar.reset_envs + accept_step(reset_mask)
    (optimizer.py:275-316, fit.py:408-422) against GraspStepper.step_reset, incl. two reference quirks -- in the reset
    iteration ALL rows' contact points come from the freshly drawn indices (hand_model.py:815-831), and in the iteration
    after it rejected rows get old + new gradient back (leaf .grad accumulates in place).  Teacher-forced."""
    g = _load(golden_dir, "mala_ext_allegro_sphere_b8_n4.npz")
    st = _stepper_from_fixture(gq, g, 4)
    f32 = lambda k: torch.tensor(g[k], dtype=torch.float32).cuda()
    draws = lambda p: (f32(f"{p}_u_switch"), torch.tensor(g[f"{p}_new_idx"]).cuda(), f32(f"{p}_u_accept"))
    st.reset(f32("R_hand_pose0"), torch.tensor(g["R_contact_idx0"]).cuda())
    np.testing.assert_allclose(st.energy.cpu().numpy(), g["R_energy0"], rtol=2e-4)
    st.energy.copy_(f32("R_energy0"))
    st.step_count.copy_(torch.tensor(g["R_step0"]).cuda())
    st.step(draws=draws("R_s1"))
    torch.cuda.synchronize()
    assert abs(float(st.s_out[0]) - 0.005 * 0.95**2) < 1e-8 and abs(float(st.temperature[0]) / 18 / 0.95**5 - 1.5) <= 0.5
    _check_iteration(st, g, "R_s1")
    # the reset iteration
    _force_state(st, g, "R_s1", f32)
    mask = torch.tensor(g["R_s2_reset_mask"])
    st.step_reset(mask, f32("R_s2_reset_pose"), torch.tensor(g["R_s2_reset_idx"]), draws=draws("R_s2"))
    torch.cuda.synchronize()
    assert st.accept.cpu().bool()[mask].all()
    _check_iteration(st, g, "R_s2")
    # the iteration after it
    _force_state(st, g, "R_s2", f32)
    assert st._after_reset
    st.step(draws=draws("R_s3"))
    torch.cuda.synchronize()
    rejected = [r for r, a in enumerate(g["R_s3_accept"].tolist()) if not a]
    assert rejected, "the fixture must contain a rejected row after the reset"
    _check_iteration(st, g, "R_s3", check_grad_rows=rejected)
    assert not st._after_reset
    # a scheduled re-initialisation whose mask comes out EMPTY (fit.py:412: nothing is re-initialised) and the iteration
    # after it: both are ordinary iterations -- evaluated at the proposal's contact indices, no unconditional accepts, and
    # the rows rejected in R_s5 get the OLD gradient back (no old + new accumulation: no leaf pose was created).  Once
    # with the mask passed in, once with the on-device rule z > threshold (fit.py:409).
    for how in ("mask", "z_rule"):
        _force_state(st, g, "R_s3", f32)
        m4 = torch.tensor(g["R_s4_reset_mask"])
        assert not m4.any()
        if how == "mask":
            st.step_reset(m4, f32("R_s4_reset_pose"), torch.tensor(g["R_s4_reset_idx"]), draws=draws("R_s4"))
        else:
            st.step_reset(None, f32("R_s4_reset_pose"), torch.tensor(g["R_s4_reset_idx"]), draws=draws("R_s4"),
                          z_threshold=float(g["R_empty_mask_threshold"]))
        torch.cuda.synchronize()
        assert not bool(st.reset_mask.any())
        _check_iteration(st, g, "R_s4", check_grad_rows=[r for r, a in enumerate(g["R_s4_accept"].tolist()) if not a])
        _force_state(st, g, "R_s4", f32)
        st.step(draws=draws("R_s5"))
        torch.cuda.synchronize()
        rejected = [r for r, a in enumerate(g["R_s5_accept"].tolist()) if not a]
        assert rejected
        _check_iteration(st, g, "R_s5", check_grad_rows=rejected)
        np.testing.assert_allclose(st.grad.cpu().numpy()[rejected], g["R_s4_grad"][rejected], rtol=1e-6, atol=1e-9)


def test_mala_clip_grad_with_nan_inf_matches_reference_optimizer(gq, golden_dir):
    """clip_grad=True with NaN / +-inf / out-of-range gradient entries (optimizer.py:211-215,235-250) on the GPU, both
    through the fused launches (proposal as the head of the FK kernel) and the stand-alone gq_mala_propose."""
    g = _load(golden_dir, "mala_ext_allegro_sphere_b8_n4.npz")
    f32 = lambda k: torch.tensor(g[k], dtype=torch.float32).cuda()
    draws = lambda p: (f32(f"{p}_u_switch"), torch.tensor(g[f"{p}_new_idx"]).cuda(), f32(f"{p}_u_accept"))
    for fused in (False, True):
        st = _stepper_from_fixture(gq, g, 4, mala_cfg={"clip_grad": True})
        st._fuse_loop = fused
        st.reset(f32("C_hand_pose0"), torch.tensor(g["C_contact_idx0"]).cuda())
        st.energy.copy_(f32("C_energy0"))
        st.grad.copy_(f32("C_s1_grad_in"))
        assert torch.isnan(st.grad).any() and torch.isinf(st.grad).any()
        st.step(draws=draws("C_s1"))
        torch.cuda.synchronize()
        assert torch.isfinite(st.pose_new).all() and torch.isfinite(st.ema).all()
        _check_iteration(st, g, "C_s1")
        _force_state(st, g, "C_s1", f32)
        st.step(draws=draws("C_s2"))
        torch.cuda.synchronize()
        _check_iteration(st, g, "C_s2")


# ---------------------------------------------------------------------------------------------------------------
# boundary: dispatcher-registered ops, the reference-shaped metric call with solver_cls, contact candidates
# ---------------------------------------------------------------------------------------------------------------
def test_torch_library_ops_opcheck(gq, golden_dir):
    """torch.library.opcheck on the registered ops: schema (no undeclared mutation / aliasing), fake kernels consistent
    with the real ones, autograd registered through the dispatcher."""
    g = _load(golden_dir, "span_n4_k4.npz")
    utils = ("test_schema", "test_autograd_registration", "test_faketensor")
    ns = torch.ops.graspqp_amd
    pts = torch.randn(33, 3, device="cuda").mul(0.05).requires_grad_()
    fv = torch.tensor(meshes.box(), device="cuda")
    torch.library.opcheck(ns.compute_sdf, (pts, fv), test_utils=utils)
    bvh = gq.ops.Bvh(meshes.box())
    torch.library.opcheck(ns.sdf_bvh, (pts, bvh.hid), test_utils=utils)
    F = torch.tensor(g["F"], dtype=torch.float32).cuda().requires_grad_()
    torch.library.opcheck(ns.lsq_box_qp, (F, torch.zeros(F.shape[0], 6, device="cuda"), 1.0, 21.0, 1e-4, 5e-2, 12), test_utils=utils)
    cp = torch.tensor(g["contact_pts"], dtype=torch.float32).cuda().requires_grad_()
    cn = torch.tensor(g["contact_normals"], dtype=torch.float32).cuda()
    cog = torch.tensor(g["cog"], dtype=torch.float32).cuda()
    torch.library.opcheck(ns.fc_energy, (cp, cn, cog, 4, 0.2, 5.0, 20.0, 0.1, 2.0, 5e-2, 12), test_utils=utils)
    hand = gq.ops.HandHandle(get_hand_spec("allegro"))
    hp = _rand_pose(hand.spec, 3, 5).float().cuda().requires_grad_()
    idx = torch.randint(hand.spec.n_contact_candidates, (3, 4), device="cuda")
    torch.library.opcheck(ns.fk_contacts, (hp, idx, hand.hid), test_utils=utils)
    # the dispatcher route gives the same numbers and gradients as the wrapper functions
    e, xs = gq.ops.fc_energy(cp, cn, cog, n_cone_vecs=4)
    e2, xs2, nit, _ = ns.fc_energy(cp, cn, cog, 4, 0.2, 5.0, 20.0, 0.1, 2.0, 5e-2, 12)
    assert torch.equal(e, e2) and torch.equal(xs, xs2) and int(nit) >= 1
    (g1,) = torch.autograd.grad(e.sum(), cp)
    (g2,) = torch.autograd.grad(e2.sum(), cp)
    assert torch.equal(g1, g2)


def test_kernels_launch_on_torchs_current_stream(gq):
    """_C.stream_ptr() hands the C ABI torch's CURRENT stream (raw handle): a side stream entered with torch.cuda.stream()
    is the one the kernels are launched on, and leaving the context returns to the default stream."""
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream().cuda_stream
    assert (gq.C.stream_ptr().value or 0) == main
    with torch.cuda.stream(side):
        assert gq.C.stream_ptr().value == side.cuda_stream
        x = torch.zeros(1 << 20, device="cuda")
        gq.C.call("gq_fill", gq.C.f32(x), 3.0, x.numel(), gq.C.stream_ptr())
    side.synchronize()
    assert (gq.C.stream_ptr().value or 0) == main and float(x.min()) == 3.0


def test_class_surface_term_ops_equal_the_reference_expressions(gq):
    """The one-launch energy terms of the class surface (csrc/terms.hip: E_dis in both `method` forms, E_joints, E_pen, the
    signed distance of ObjectModel.cal_distance) against the torch expressions of the reference (core/energy.py:25-28,
    47-52,58-61; core/object_model.py:222-227) in fp64: values and gradients under a random upstream, ragged sizes, exact
    zeros and sign changes included; opcheck on the registered ops."""
    rng = np.random.default_rng(3)
    t = lambda a, dt=torch.float32: torch.tensor(np.asarray(a), dtype=dt, device="cuda")
    B, n, P, J = 37, 12, 2500, 16
    dist = rng.normal(size=(B, n)) * 0.02
    dist[0, :3] = 0.0
    on = rng.normal(size=(B, n, 3)); on /= np.linalg.norm(on, axis=-1, keepdims=True)
    hn = rng.normal(size=(B, n, 3)); hn /= np.linalg.norm(hn, axis=-1, keepdims=True)
    up = rng.uniform(0.5, 2.0, B)
    for with_normals in (True, False):
        d32, h32 = t(dist).requires_grad_(), t(hn).requires_grad_()
        e = gq.ops.energy_dis(d32, t(on), h32, with_normals=with_normals)
        (e * t(up)).sum().backward()
        d64, h64 = t(dist, torch.float64).requires_grad_(), t(hn, torch.float64).requires_grad_()
        if with_normals:
            e64 = ((1 - torch.sum((-t(on, torch.float64)) * h64, dim=-1)).exp() * d64.abs()).sum(-1)
        else:
            e64 = torch.sum(d64.abs(), dim=-1)
        (e64 * t(up, torch.float64)).sum().backward()
        np.testing.assert_allclose(e.detach().cpu().numpy(), e64.detach().cpu().numpy(), rtol=5e-6)
        np.testing.assert_allclose(d32.grad.cpu().numpy(), d64.grad.cpu().numpy(), rtol=5e-6, atol=1e-7)
        if with_normals:
            np.testing.assert_allclose(h32.grad.cpu().numpy(), h64.grad.cpu().numpy(), rtol=5e-6, atol=1e-8)
        else:
            assert h32.grad is None
    lo, hi = -rng.uniform(0.2, 0.5, J), rng.uniform(0.5, 1.5, J)
    pose = np.concatenate([rng.normal(size=(B, 9)), rng.uniform(-1.0, 2.0, size=(B, J))], 1)
    pose[1, 9:] = hi      # exactly on the limits: inside
    pose[2, 9:] = lo
    p32 = t(pose).requires_grad_()
    e = gq.ops.energy_joints(p32, t(lo), t(hi))
    (e * t(up)).sum().backward()
    p64 = t(pose, torch.float64).requires_grad_()
    th, l64, h64 = p64[:, 9:], t(lo, torch.float64), t(hi, torch.float64)
    e64 = torch.sum((th > h64) * (th - h64), dim=-1) + torch.sum((th < l64) * (l64 - th), dim=-1)
    (e64 * t(up, torch.float64)).sum().backward()
    np.testing.assert_allclose(e.detach().cpu().numpy(), e64.detach().cpu().numpy(), rtol=5e-6, atol=1e-7)
    np.testing.assert_allclose(p32.grad.cpu().numpy(), p64.grad.cpu().numpy(), rtol=1e-6, atol=0)
    pen = rng.normal(size=(B, P)) * 0.01 - 0.008
    pen[3, :10] = 0.0
    q32 = t(pen).requires_grad_()
    e = gq.ops.energy_pen(q32)
    (e * t(up)).sum().backward()
    q64 = t(pen, torch.float64).requires_grad_()
    e64 = torch.where(q64 <= 0, torch.zeros_like(q64), q64).sum(-1)
    (e64 * t(up, torch.float64)).sum().backward()
    np.testing.assert_allclose(e.detach().cpu().numpy(), e64.detach().cpu().numpy(), rtol=5e-6, atol=1e-8)
    np.testing.assert_allclose(q32.grad.cpu().numpy(), q64.grad.cpu().numpy(), rtol=1e-6, atol=0)
    N = 1000
    d2 = np.abs(rng.normal(size=N)) * 1e-3
    d2[:5] = 0.0
    sg = rng.choice([-1, 1], N)
    nr = rng.normal(size=(N, 3))
    a32 = t(d2).requires_grad_()
    dis, nout = gq.ops.signed_distance(a32, t(sg, torch.int32), t(nr))
    (dis * t(rng.normal(size=N))).sum().backward()
    a64 = t(d2, torch.float64).requires_grad_()
    dis64 = torch.sqrt(a64 + 1e-8) * (-t(sg, torch.float64))
    np.testing.assert_allclose(dis.detach().cpu().numpy(), dis64.detach().cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(nout.cpu().numpy(), nr.astype(np.float32) * sg[:, None].astype(np.float32), rtol=0, atol=0)
    g64 = torch.autograd.grad(dis64, a64, torch.ones_like(dis64))[0]  # d dis / d d2, element-wise
    g32 = torch.autograd.grad(gq.ops.signed_distance(a32, t(sg, torch.int32), t(nr))[0], a32, torch.ones(N, device="cuda"))[0]
    np.testing.assert_allclose(g32.cpu().numpy(), g64.cpu().numpy(), rtol=5e-6)
    utils = ("test_schema", "test_autograd_registration", "test_faketensor")
    ns = torch.ops.graspqp_amd
    torch.library.opcheck(ns.energy_dis, (t(dist).requires_grad_(), t(on), t(hn).requires_grad_(), True), test_utils=utils)
    torch.library.opcheck(ns.energy_joints, (t(pose).requires_grad_(), t(lo), t(hi)), test_utils=utils)
    torch.library.opcheck(ns.energy_pen, (t(pen).requires_grad_(),), test_utils=utils)
    torch.library.opcheck(ns.signed_distance, (t(d2).requires_grad_(), t(sg, torch.int32), t(nr)), test_utils=utils)


def test_reference_shaped_metric_call_with_solver_cls(gq, golden_dir):
    """SpanMetricWrapper driven the way the reference's factory builds it (registry.py:108-118:
    metric class + metric_kwargs incl. solver_cls), against the factory shortcut and the fixture; a foreign solver_cls is
    refused."""
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF
    from graspqp_amd.metrics import SpanMetricWrapper, SQPLsqSolver
    from graspqp_amd.metrics.ops.span import OverallFrictionConeSpanMetric

    g = _load(golden_dir, "span_n12_k4.npz")
    pts, nrm, cog = (torch.tensor(g[k], dtype=torch.float32).cuda() for k in ("contact_pts", "contact_normals", "cog"))
    fn = SpanMetricWrapper(OverallFrictionConeSpanMetric,
                           metric_kwargs={"solver_cls": SQPLsqSolver, "friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
    p1 = pts.clone().requires_grad_()
    e1, x1 = fn(contact_pts=p1, contact_normals=nrm, sdf=None, cog=cog, with_solution=True, svd_gain=0.1)
    e2, x2 = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})(
        contact_pts=pts, contact_normals=nrm, sdf=None, cog=cog, with_solution=True, svd_gain=0.1)
    assert torch.equal(e1.detach(), e2) and torch.equal(x1, x2)
    e1.sum().backward()
    assert torch.isfinite(p1.grad).all() and p1.grad.abs().sum() > 0
    # the metric class itself, reference-shaped: (res, basis, svd_scales, values)
    m = OverallFrictionConeSpanMetric.from_dim(48, 6, batch_size=pts.shape[0], solver_cls=SQPLsqSolver, friction=0.2, n_cone_vecs=4)
    m._max_limit_value = 20.0
    res, basis, svd, xs = m(pts, nrm, cog)
    np.testing.assert_allclose(svd.squeeze(-1).cpu().numpy(), g["svd"], rtol=2e-3)
    np.testing.assert_allclose((2 * (res.squeeze(-1) + 0.01) * torch.exp(-0.1 * svd.squeeze(-1))).cpu().numpy(),
                               e2.cpu().numpy(), rtol=1e-4)

    class Foreign:
        pass

    with pytest.raises(NotImplementedError, match="solver_cls"):
        SpanMetricWrapper(OverallFrictionConeSpanMetric, metric_kwargs={"solver_cls": Foreign})(pts, nrm, cog)


def test_get_contact_candidates_matches_oracle(gq):
    from graspqp_amd.core.hand_model import HandModel

    spec = get_hand_spec("shadow_hand")
    hp = _rand_pose(spec, 4, 9)
    oh = omodels.OracleHand(spec, torch.float64)
    oh.set_parameters(hp, torch.zeros(4, 2, dtype=torch.long))
    pw, nw = okin.contact_candidates_world(spec, oh.current_status, oh.global_rotation, oh.global_translation)
    hm = HandModel(spec, "cuda")
    hm.set_parameters(hp.float().cuda(), torch.zeros(4, 2, dtype=torch.long).cuda())
    cp, cn = hm.get_contact_candidates(with_normals=True)
    assert cp.shape == (4, spec.n_contact_candidates, 3)
    np.testing.assert_allclose(cp.cpu().numpy(), pw.numpy(), atol=3e-6)
    np.testing.assert_allclose(cn.cpu().numpy(), nw.numpy(), atol=3e-6)
    assert torch.equal(hm.get_contact_candidates(), cp)


# ---------------------------------------------------------------------------------------------------------------
# penetration-only E_pen query (what the stepper runs) against the full query and the oracle, every hand
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hand_name", ["allegro", "shadow_hand", "robotiq3", "ability_hand", "panda"])
def test_penetration_only_query_never_misses_what_the_full_query_and_the_oracle_agree_on(gq, hand_name):
    """The occupancy-grid shortcut of the E_pen query (deviation iv) on deep-penetration scenes with
    joints anywhere inside their limits: the two modes may differ where TorchSDF's sign is decided by a near-tie between
    faces (deviation v: ~1e-5 of the points, either mode may be the one that agrees with the fp64 oracle), but the
    shortcut must never lose a penetration that the full query and the oracle agree on -- the failure the centre-only
    voxel probe had on the shadow hand's thumb link."""
    from graspqp_amd.core.hand_model import HandModel
    from ref_cpu import sdf as osdf

    spec = get_hand_spec(hand_name)
    be, P = 48, 2500
    B = 2 * be
    fvs = [meshes.superquadric(3), meshes.superquadric(4)]
    surf = torch.tensor(np.stack([meshes.surface_points(f, P, oversample=4, seed=42) for f in fvs])).cuda()
    g = torch.Generator().manual_seed(7)
    t = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * (0.02 + 0.08 * torch.rand(B, 1, generator=g))
    lo, hi = torch.tensor(spec.joints_lower), torch.tensor(spec.joints_upper)
    th = lo + (hi - lo) * torch.rand(B, spec.n_dofs, generator=g)
    hp = torch.cat([t, torch.randn(B, 6, generator=g), th], 1).cuda()
    idx = torch.randint(spec.n_contact_candidates, (B, 4), generator=g).cuda()
    hm = HandModel(spec, "cuda")
    hm.set_parameters(hp, idx)
    d0 = torch.relu(hm.cal_distance(surf, penetration_only=0))
    d1 = torch.relu(hm.cal_distance(surf, penetration_only=1))
    torch.cuda.synchronize()
    n_pen = int((d0 > 0).sum())
    assert n_pen > 5000, "scene must penetrate"
    tol = 3e-5  # near-tied faces may swap: <= 1e-5 m on the distance (deviation v)
    bad = torch.nonzero((d0 - d1).abs() > tol).tolist()
    assert len(bad) <= 5e-4 * n_pen, (len(bad), n_pen)
    oh = omodels.OracleHand(spec, torch.float64)
    oh.set_parameters(hp.double().cpu(), idx.cpu())
    for r, j in bad:
        x = surf[r // be, j].double().cpu()[None, None]
        xh = (x - oh.global_translation[r : r + 1].unsqueeze(1)) @ oh.global_rotation[r : r + 1]
        best = -1e30
        for l, fv in enumerate(oh.link_faces):
            T = oh.current_status[r : r + 1, l]
            d2, sgn, _, _ = osdf.compute_sdf(((xh - T[:, :3, 3].unsqueeze(1)) @ T[:, :3, :3]).reshape(-1, 3), fv)
            best = max(best, float(torch.sqrt(d2 + 1e-8) * (-sgn)))
        o = max(best, 0.0)
        full_ok, short_ok = abs(float(d0[r, j]) - o) <= tol, abs(float(d1[r, j]) - o) <= tol
        assert short_ok or not full_ok, f"row {r} point {j}: shortcut {float(d1[r, j])} full {float(d0[r, j])} oracle {o}"
