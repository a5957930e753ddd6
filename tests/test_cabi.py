"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/graspqp_hip.h declares, the host mirror has the reference's surface, and nothing falls back to the CPU."""
import os
import re

import numpy as np
import pytest
import torch

from graspqp_amd import _C


def test_library_exports_every_declared_symbol():
    protos = _C.parse_header()
    assert len(protos) >= 30
    lib = _C.lib()
    for name in protos:
        assert hasattr(lib, name), name
    assert lib.gq_version() >= 100
    for required in ("gq_sdf_forward", "gq_sdf_backward", "gq_boxqp_forward", "gq_boxqp_backward", "gq_fk_forward",
                     "gq_fk_backward", "gq_hand_pen_forward", "gq_hand_pen_backward", "gq_self_pen_forward",
                     "gq_mala_propose", "gq_mala_accept", "gq_fc_forward", "gq_fc_backward"):
        assert required in protos, required


def test_argument_validation_without_gpu():
    """Size helpers are pure host code: callable without a GPU, bad arguments give an error code + message."""
    import ctypes

    lib = _C.lib()
    out = ctypes.c_size_t(0)
    assert lib.gq_boxqp_workspace_bytes(256, 48, 12, ctypes.byref(out)) == 0 and out.value > 256 * 12 * 5 * 48 * 4
    assert lib.gq_boxqp_workspace_bytes(256, 0, 12, ctypes.byref(out)) != 0
    assert b"bad arguments" in lib.gq_last_error()
    assert lib.gq_fc_workspace_bytes(4, 4, 4, 12, ctypes.byref(out)) == 0 and out.value > 0
    assert lib.gq_sdf_workspace_bytes(1000, ctypes.byref(out)) == 0 and out.value >= 64000


def test_no_cpu_fallback():
    from graspqp_amd import ops

    with pytest.raises(RuntimeError):
        ops.compute_sdf(torch.zeros(4, 3), torch.zeros(2, 3, 3))
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.hands import get_hand_spec

    with pytest.raises(RuntimeError):
        HandModel(get_hand_spec("allegro"), device="cpu")


def test_product_does_not_import_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "graspqp_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+(ref_cpu|oracle)", src, flags=re.M), f"{f} imports the oracle"
                assert "sys.path" not in src or "oracle" not in src, f"{f} reaches into oracle/"


def test_hand_specs():
    from graspqp_amd.hands import AVAILABLE_HANDS, get_hand_spec

    expect = {"allegro": (16, 14, 92, 25), "shadow_hand": (24, 18, 80, 22), "robotiq3": (11, 12, 48, 12),
              "ability_hand": (6, 11, 64, 19), "panda": (1, 3, 16, 2), "schunk2": (1, 7, 16, 0)}
    assert sorted(AVAILABLE_HANDS) == sorted(expect)
    for h in AVAILABLE_HANDS:
        s = get_hand_spec(h)
        assert (s.n_dofs, s.n_links, s.n_contact_candidates, s.n_spheres) == expect[h]
        assert (s.node_parent < np.arange(s.n_nodes)).all()
        assert s.coupling.shape == (s.n_nodes, s.n_dofs) and s.is_coupled == (h in ("ability_hand", "panda", "schunk2"))
        assert np.allclose(np.linalg.norm(s.cand_nrm, axis=1), 1.0, atol=1e-4)
        assert (s.joints_lower < s.joints_upper).all()


def test_metric_factory_surface():
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF
    from graspqp_amd.metrics import SQPLsqSolver

    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
    assert callable(fn)
    with pytest.raises(NotImplementedError):
        GF.create(GF.MetricType.GRASPQP_SCIPY)
    from graspqp_amd.metrics import DexgraspSpanMetric

    assert isinstance(GF.create(GF.MetricType.DEXGRASP), DexgraspSpanMetric)
    s = SQPLsqSolver.from_mat(torch.zeros(2, 1, 6, 48), torch.zeros(2, 1, 6))
    assert s._batch_size == 2 and s._num_wrenches == 48


def test_descriptor_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of the descriptor structs (graspqp_amd/_C.py) must have the size and field offsets that a C
    compiler gives the structs of include/graspqp_hip.h -- a silent mismatch would hand garbage pointers to kernels."""
    import ctypes
    import shutil
    import subprocess

    from graspqp_amd import _C

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = {"gqHandDesc": _C.HandDesc, "gqRowEnergyDesc": _C.RowEnergyDesc, "gqFcStepDesc": _C.FcStepDesc,
             "gqPenStepDesc": _C.PenStepDesc, "gqProposeDesc": _C.ProposeDesc, "gqAcceptDesc": _C.AcceptDesc,
             "gqSdfDesc": _C.SdfDesc, "gqInitDesc": _C.InitDesc, "gqAltFcDesc": _C.AltFcDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "graspqp_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            m = re.fullmatch(r"(\w+_axis)(\d)", fname)  # float[3] members are mirrored element by element
            cfield = f"{m.group(1)}[{m.group(2)}]" if m else fname
            lines.append(f'  printf(" %zu", offsetof({cname}, {cfield}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include")
    subprocess.check_call(["gcc", "-std=c11", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().strip().splitlines()
    for line in out:
        name, *nums = line.split()
        cls = pairs[name]
        want = [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(x) for x in nums] == want, f"{name}: C {nums} vs ctypes {want}"


def test_solver_cls_is_honoured_or_refused_loudly():
    """reference span.py:23-37,299 / registry.py:112: `solver_cls` selects the QP back end.  The HIP class is accepted, a
    foreign one is refused (never silently replaced)."""
    from graspqp_amd.metrics import SpanMetricWrapper, SQPLsqSolver
    from graspqp_amd.metrics.ops.span import OverallFrictionConeSpanMetric

    class Foreign:
        pass

    class Mine(SQPLsqSolver):
        pass

    OverallFrictionConeSpanMetric(solver_cls=SQPLsqSolver)
    OverallFrictionConeSpanMetric.from_dim(48, 6, solver_cls=Mine, friction=0.2, n_cone_vecs=4)
    with pytest.raises(NotImplementedError, match="solver_cls"):
        OverallFrictionConeSpanMetric(solver_cls=Foreign)
    w = SpanMetricWrapper(OverallFrictionConeSpanMetric, metric_kwargs={"solver_cls": Foreign, "friction": 0.2})
    with pytest.raises(NotImplementedError, match="solver_cls"):
        w(torch.zeros(2, 4, 3), torch.zeros(2, 4, 3), torch.zeros(2, 3))


def test_ops_are_registered_with_the_dispatcher():
    """north_star: "exposed as torch ops" -- torch.library custom ops with fake kernels (traceable) and autograd."""
    from graspqp_amd import ops  # noqa: F401

    ns = torch.ops.graspqp_amd
    for name in ("compute_sdf", "sdf_backward", "sdf_meshset", "sdf_bvh", "box_qp", "box_qp_backward", "lsq_box_qp", "lsq_box_qp_backward",
                 "fc_energy", "fc_energy_backward", "fk_contacts", "fk_backward", "hand_pen", "hand_pen_backward", "self_pen"):
        assert hasattr(ns, name), name
    # fake (meta) kernels: shapes without touching a GPU
    from torch._subclasses.fake_tensor import FakeTensorMode

    with FakeTensorMode():
        pts, fv = torch.empty(7, 3, device="cuda"), torch.empty(20, 3, 3, device="cuda")
        d2, sgn, nrm, cls = ns.compute_sdf(pts, fv)
        assert d2.shape == (7,) and sgn.dtype == torch.int32 and cls.shape == (7, 3)
        x, lam, slack, nit = ns.lsq_box_qp(torch.empty(5, 6, 48, device="cuda"), torch.empty(5, 6, device="cuda"), 1.0, 21.0,
                                           1e-4, 5e-2, 12)
        assert x.shape == (5, 48) and lam.shape == (5, 96)
    # no CPU kernel: a CPU tensor is refused by the dispatcher
    with pytest.raises(NotImplementedError):
        ns.compute_sdf(torch.zeros(4, 3), torch.zeros(2, 3, 3))


@pytest.mark.skipif(not os.path.isdir("/root/reference/graspqp/src/graspqp"), reason="needs the reference tree (build container)")
def test_integration_aliases_import_the_reference_modules():
    """INTEGRATION.md sections 1-2: with `torchsdf` and `qpth` aliased to graspqp_amd, the reference's own
    metrics/solver/qp_solver.py imports and builds its solver around OUR QPFunction (no compute without a GPU)."""
    import importlib.util
    import sys
    import types

    import graspqp_amd.metrics
    import graspqp_amd.torchsdf

    saved = {k: sys.modules.get(k) for k in ("torchsdf", "qpth", "qpth.qp")}
    try:
        sys.modules["torchsdf"] = graspqp_amd.torchsdf
        qp = types.ModuleType("qpth.qp")
        qp.QPFunction = graspqp_amd.metrics.QPFunction
        pkg = types.ModuleType("qpth")
        pkg.qp = qp
        sys.modules["qpth"], sys.modules["qpth.qp"] = pkg, qp
        from torchsdf import compute_sdf, index_vertices_by_faces

        assert compute_sdf is graspqp_amd.torchsdf.compute_sdf
        assert index_vertices_by_faces(torch.arange(12.0).view(4, 3), torch.tensor([[0, 1, 2]])).shape == (1, 3, 3)
        spec = importlib.util.spec_from_file_location(
            "_ref_qp_solver", "/root/reference/graspqp/src/graspqp/metrics/solver/qp_solver.py")
        mod = importlib.util.module_from_spec(spec)
        sys.dont_write_bytecode = True
        spec.loader.exec_module(mod)
        assert mod.QPFunction is graspqp_amd.metrics.QPFunction
        solver = mod.SQPLsqSolver.from_mat(torch.zeros(2, 6, 48), torch.zeros(2, 6))
        assert solver._batch_size == 2
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_coupled_hands_and_grasp_types():
    """hands/ability_hand.py:9-41, hands/panda.py:6-26 (joint coupling) and hand_model.py:438-451,550-589 (grasp types)."""
    from graspqp_amd.hands import get_hand_spec

    ab = get_hand_spec("ability_hand")
    assert ab.full_joint_names == ["index_q1", "index_q2", "middle_q1", "middle_q2", "pinky_q1", "pinky_q2", "ring_q1", "ring_q2",
                                   "thumb_q1", "thumb_q2"]
    th = np.array([[0.1, 0.2, 0.3, 0.4, 0.5, 0.6]])
    m = 1.05851325
    np.testing.assert_allclose(ab.full_joint_angles(th)[0], [0.1, 0.1 * m, 0.2, 0.2 * m, 0.3, 0.3 * m, 0.4, 0.4 * m, 0.5, 0.6], rtol=1e-6)
    pa = get_hand_spec("panda")
    np.testing.assert_allclose(pa.full_joint_angles(np.array([[0.03]]))[0], [0.03, 0.03])
    al = get_hand_spec("allegro")
    assert "pinch" in al.grasp_types() or len(al.grasp_types()) > 0
    for gt in al.grasp_types():
        try:
            sub = get_hand_spec("allegro", grasp_type=gt)
        except NotImplementedError:
            continue  # asks for more candidates per link than the reference's dump holds
        links = set(np.asarray(al.link_names)[np.unique(sub.cand_link)])
        import json

        assert links <= set(json.loads(al.eigengrasps)[gt]) and 0 < sub.n_contact_candidates <= al.n_contact_candidates
        if gt == "pinch":
            for i, jn in enumerate(sub.joint_names):
                folded = ("middle" in jn or "ring" in jn) and "joint_0" not in jn
                assert sub.default_state[i] == (al.joints_upper[i] if folded else al.default_state[i])
    with pytest.raises(ValueError):
        get_hand_spec("allegro", grasp_type="no_such_type")
    assert get_hand_spec("allegro", grasp_type="all").n_contact_candidates == 92


def test_schunk_gripper_spec_and_mesh_readers(tmp_path):
    """reference hands/schunk.py:12-37,66-83 + assets/schunk_2f: one actuated prismatic joint, the other finger mirrors it
    (theta, -theta); collision meshes only, read from binary STL and COLLADA (hands/mesh_io.py); 2 x 8 contact candidates
    sampled on the contact patches at set-up (no dump in the reference: parity unpinned, checked by construction)."""
    from graspqp_amd.hands import get_hand_spec, mesh_io
    from graspqp_amd.utils import meshes

    s = get_hand_spec("schunk2")
    assert s.n_dofs == 1 and s.joint_names == ["egu_50_prismatic_1"]
    assert s.full_joint_names == ["egu_50_prismatic_1", "egu_50_prismatic_2"] and s.is_coupled
    np.testing.assert_allclose(s.full_joint_angles(np.array([[0.01]]))[0], [0.01, -0.01])
    assert list(s.node_type) == [2, 2]  # prismatic
    np.testing.assert_allclose([s.joints_lower[0], s.joints_upper[0]], [-0.012, 0.039], rtol=1e-6)
    assert s.link_names == ["egu_50_base_link", "egu_50_translational_left", "egu_50_base_link_left", "egu_50_finger_down",
                            "egu_50_translational_right", "egu_50_base_link_right", "egu_50_finger_up"]
    assert [s.link_faces(l).shape[0] for l in range(s.n_links)] == [4520, 1720, 2088, 3272, 1720, 2088, 3272]
    assert s.n_spheres == 0 and s.n_contact_candidates == 16
    fingers = {s.link_names[l] for l in np.unique(s.cand_link)}
    assert fingers == {"egu_50_finger_down", "egu_50_finger_up"} and (np.bincount(s.cand_link)[[3, 6]] == 8).all()
    np.testing.assert_allclose(np.linalg.norm(s.cand_nrm, axis=1), 1.0, rtol=1e-5)
    for l in (3, 6):  # every candidate lies on its link's contact patch and on the link mesh; the normal is that face's
        sel = s.cand_link == l
        patch = s.patch_verts[s.patch_face_offset[l]:s.patch_face_offset[l + 1]].astype(np.float64)
        assert len(patch) in (24, 32)  # gripper_finger_down / _up patches
        _, d_patch, _ = meshes.closest_face(s.cand_pos[sel].astype(np.float64), patch)
        _, d_mesh, fi = meshes.closest_face(s.cand_pos[sel].astype(np.float64), s.link_faces(l).astype(np.float64))
        assert d_patch.max() < 1e-6 and d_mesh.max() < 1e-5
        fv = s.link_faces(l).astype(np.float64)
        nn = np.cross(fv[fi, 1] - fv[fi, 0], fv[fi, 2] - fv[fi, 0])
        nn /= np.linalg.norm(nn, axis=1, keepdims=True)
        assert (np.abs((nn * s.cand_nrm[sel]).sum(1)) > 1 - 1e-5).all()
    # the finger meshes open along +-x of the base: a positive joint value moves finger_down and finger_up apart symmetrically
    import sys

    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    from ref_cpu import kin as okin

    T0 = okin.forward_kinematics(s, torch.tensor([[0.0]], dtype=torch.float64))[0]
    T1 = okin.forward_kinematics(s, torch.tensor([[0.02]], dtype=torch.float64))[0]
    move = (T1[:, :3, 3] - T0[:, :3, 3]).numpy()
    np.testing.assert_allclose(move[3], -move[6], atol=1e-12)
    assert abs(np.linalg.norm(move[3]) - 0.02) < 1e-9 and np.abs(move[0]).max() == 0
    # readers: ASCII STL == binary STL; a COLLADA scene with a scaled + rotated node chain, <polylist> quads and a <translate>
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 1], [1, 0, 1], [0, 2, 1]]], dtype=np.float32)
    rec = np.zeros(2, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    rec["v"] = tri
    (tmp_path / "b.stl").write_bytes(b"\0" * 80 + np.uint32(2).tobytes() + rec.tobytes())
    txt = "solid t\n" + "".join("facet normal 0 0 1\nouter loop\n" + "".join(f"vertex {v[0]} {v[1]} {v[2]}\n" for v in t)
                                + "endloop\nendfacet\n" for t in tri) + "endsolid t\n"
    (tmp_path / "a.stl").write_text(txt)
    np.testing.assert_array_equal(mesh_io.load_mesh_triangles(str(tmp_path / "b.stl")), tri.astype(np.float64))
    np.testing.assert_array_equal(mesh_io.load_mesh_triangles(str(tmp_path / "a.stl")), tri.astype(np.float64))
    dae = """<?xml version="1.0"?><COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
<asset><unit name="inch" meter="0.0254"/><up_axis>Z_UP</up_axis></asset>
<library_geometries><geometry id="g"><mesh>
<source id="p"><float_array id="pa" count="12">0 0 0 1 0 0 1 1 0 0 1 0</float_array>
<technique_common><accessor source="#pa" count="4" stride="3"/></technique_common></source>
<vertices id="v"><input semantic="POSITION" source="#p"/></vertices>
<polylist count="1"><input semantic="VERTEX" source="#v" offset="0"/><input semantic="NORMAL" source="#p" offset="1"/>
<vcount>4</vcount><p>0 0 1 0 2 0 3 0</p></polylist></mesh></geometry></library_geometries>
<library_visual_scenes><visual_scene id="s"><node id="a"><matrix>2 0 0 0 0 2 0 0 0 0 2 0 0 0 0 1</matrix>
<node id="b"><translate>0 0 5</translate><rotate>0 0 1 90</rotate><instance_geometry url="#g"/></node></node></visual_scene>
</library_visual_scenes><scene><instance_visual_scene url="#s"/></scene></COLLADA>"""
    (tmp_path / "q.dae").write_text(dae)
    got = mesh_io.load_mesh_triangles(str(tmp_path / "q.dae"))
    # quad fan (0,1,2),(0,2,3); p -> 2 * (Rz(90) p + (0,0,5)); the <unit> is metadata only (as in trimesh)
    quad = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float64)
    Rz = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    want = 2 * (quad @ Rz.T + np.array([0, 0, 5.0]))
    np.testing.assert_allclose(got, want[[[0, 1, 2], [0, 2, 3]]], atol=1e-12)
    ref = "/root/reference/graspqp/assets/schunk_2f"
    if os.path.isdir(ref):  # build container: the packaged spec == a rebuild from the reference's files, and the COLLADA
        # reader's node-chain handling (0.0254 scale x axis swap) against the extents of the STL twin shipped for the same link
        r = get_hand_spec("schunk2", os.path.dirname(ref))
        for f in ("face_verts", "cand_pos", "cand_nrm", "cand_link", "node_pre", "coupling", "link_offset"):
            np.testing.assert_array_equal(getattr(r, f), getattr(s, f))
        for stem in ("gripper_finger_down", "base_link"):
            a = mesh_io.load_mesh_triangles(f"{ref}/meshes/collisions/{stem}.dae")
            b = mesh_io.load_mesh_triangles(f"{ref}/meshes/collisions/{stem}.STL")
            np.testing.assert_allclose(a.reshape(-1, 3).min(0), b.reshape(-1, 3).min(0), atol=2e-6)
            np.testing.assert_allclose(a.reshape(-1, 3).max(0), b.reshape(-1, 3).max(0), atol=2e-6)
            # (the STL twins are re-meshed hulls without the screw holes: same extents, not the same area)


def test_contact_patches_and_grasp_types_asking_for_more_candidates():
    """hand_model.py:269-296,333-335: candidates are farthest-point samples of a link's contact patch, their normals the
    normals of the closest link-mesh face.  (i) The patches baked into the specs are pinned by the reference's own dump:
    every dumped candidate lies on its link's patch.  (ii) A grasp type that wants more candidates than the dump holds
    (shadow_hand / ability_hand "pinch": 16 on two links) keeps the dumped ones first and continues the farthest-point
    sequence on the patch (parity unpinned for the added points: the reference's sample stream needs trimesh)."""
    import sys

    from graspqp_amd.hands import AVAILABLE_HANDS, get_hand_spec
    from graspqp_amd.utils import meshes

    for h in AVAILABLE_HANDS:
        s = get_hand_spec(h)
        for li in range(s.n_links):
            idx = np.nonzero(s.cand_link == li)[0]
            pv = s.patch_verts[s.patch_face_offset[li]:s.patch_face_offset[li + 1]]
            assert (len(idx) == 0) == (len(pv) == 0), (h, s.link_names[li])
            if len(idx):
                _, d2, _ = meshes.closest_face(s.cand_pos[idx], pv)
                assert np.sqrt(d2.max()) < 1e-7, (h, s.link_names[li])
    n_grown = 0
    for h, per_link in (("shadow_hand", 16), ("ability_hand", 16)):
        base, sub = get_hand_spec(h), get_hand_spec(h, grasp_type="pinch")
        assert sub.n_contact_candidates == 2 * per_link and np.all(np.diff(sub.cand_link) >= 0)
        for li in np.unique(sub.cand_link):
            i_b, i_s = np.nonzero(base.cand_link == li)[0], np.nonzero(sub.cand_link == li)[0]
            k0 = min(len(i_b), per_link)
            assert len(i_s) == per_link
            np.testing.assert_array_equal(sub.cand_pos[i_s][:k0], base.cand_pos[i_b][:k0])
            np.testing.assert_array_equal(sub.cand_nrm[i_s][:k0], base.cand_nrm[i_b][:k0])
            new_p, new_n = sub.cand_pos[i_s][k0:], sub.cand_nrm[i_s][k0:]
            if len(new_p) == 0:
                continue  # the dump already holds as many as the grasp type wants on this link
            n_grown += 1
            pv = base.patch_verts[base.patch_face_offset[li]:base.patch_face_offset[li + 1]]
            assert np.sqrt(meshes.closest_face(new_p, pv)[1].max()) < 1e-6  # on the patch
            np.testing.assert_allclose(np.linalg.norm(new_n, axis=1), 1.0, atol=1e-6)
            # normals = normal of the closest link-mesh face, cross-checked with the oracle's closest-point search
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
            import torch
            from ref_cpu import sdf as osdf

            fv = torch.tensor(base.link_faces(li), dtype=torch.float64)
            d2, sgn, nrm, closest = osdf.compute_sdf(torch.tensor(new_p, dtype=torch.float64), fv)
            q, d2m, fi = meshes.closest_face(new_p, base.link_faces(li))
            np.testing.assert_allclose(d2m, d2.numpy(), atol=1e-12)
            fvn = base.link_faces(li).astype(np.float64)
            fn = np.cross(fvn[fi, 1] - fvn[fi, 0], fvn[fi, 2] - fvn[fi, 0])
            fn /= np.linalg.norm(fn, axis=1, keepdims=True)
            np.testing.assert_allclose(new_n, fn, atol=1e-6)
            # farthest-point property: every added point is at least as far from the earlier ones as the dumped set's spacing / 2
            P = sub.cand_pos[i_s].astype(np.float64)
            D = np.linalg.norm(P[:, None] - P[None], axis=-1) + 1e9 * np.eye(len(P))
            assert D.min() > 1e-3
    assert n_grown == 3  # shadow_hand: both distal links; ability_hand: index_L2
