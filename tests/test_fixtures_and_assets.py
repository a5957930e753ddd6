"""CPU checks of the inputs every other test relies on:

* the hand data (graspqp_amd/assets/hands/*.npz, built by graspqp_amd/hands/spec.py from the reference's URDF / OBJ /
  JSON assets) against hard-coded link face counts / joint orders and -- where the reference tree is present (build
  container only) -- against an INDEPENDENT stdlib reading of the same asset files (oracle and product share one
  parser, so a parsing error would otherwise be invisible to every parity test);
* tools/make_golden.py reproduces the committed fixtures (array for array) from the reference's own files;
* the oracle's MALA* restatement against the extended fixture of the reference's optimizer.py (step counter started at
  149, a re-initialisation iteration, clip_grad with NaN / inf gradient entries)."""
import os
import subprocess
import sys
import xml.etree.ElementTree as ET

import numpy as np
import pytest
import torch

import ref_cpu
from ref_cpu import mala, models
from graspqp_amd.hands import get_hand_spec

REF_ASSETS = "/root/reference/graspqp/assets"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

EXPECT = {
    "allegro": dict(
        urdf="allegro/allegro_hand.urdf", mesh="allegro/meshes",
        faces=[324, 188, 216, 342, 188, 216, 342, 188, 216, 342, 232, 188, 216, 342],
        joints=["index_joint_0", "index_joint_1", "index_joint_2", "index_joint_3", "middle_joint_0", "middle_joint_1",
                "middle_joint_2", "middle_joint_3", "ring_joint_0", "ring_joint_1", "ring_joint_2", "ring_joint_3",
                "thumb_joint_0", "thumb_joint_1", "thumb_joint_2", "thumb_joint_3"]),
    "shadow_hand": dict(
        urdf="shadow_hand/shadow_hand.urdf", mesh="shadow_hand/meshes",
        faces=[386, 904, 502, 456, 1052, 640, 502, 456, 1052, 502, 456, 1052, 502, 456, 1052, 422, 1262, 1212],
        joints=["robot0_WRJ1", "robot0_WRJ0", "robot0_FFJ3", "robot0_FFJ2", "robot0_FFJ1", "robot0_FFJ0", "robot0_LFJ4",
                "robot0_LFJ3", "robot0_LFJ2", "robot0_LFJ1", "robot0_LFJ0", "robot0_MFJ3", "robot0_MFJ2", "robot0_MFJ1",
                "robot0_MFJ0", "robot0_RFJ3", "robot0_RFJ2", "robot0_RFJ1", "robot0_RFJ0", "robot0_THJ4", "robot0_THJ3",
                "robot0_THJ2", "robot0_THJ1", "robot0_THJ0"]),
    "robotiq3": dict(
        urdf="robotiq3/robotiq_3finger_flat.urdf", mesh="robotiq3/meshes",
        faces=[1904, 1728, 592, 804, 820, 1728, 592, 804, 820, 592, 804, 820],
        joints=["RIQ_palm_RIQ_link_0_joint", "RIQ_link_0_RIQ_link_1_joint_c", "RIQ_link_1_RIQ_link_2_joint_c",
                "RIQ_link_2_RIQ_link_3_joint_c", "RIQ_palm_RIQ_link_0_joint_b", "RIQ_link_0_RIQ_link_1_joint_b",
                "RIQ_link_1_RIQ_link_2_joint_b", "RIQ_link_2_RIQ_link_3_joint_b", "RIQ_palm_RIQ_link_1_joint_a",
                "RIQ_link_1_RIQ_link_2_joint_a", "RIQ_link_2_RIQ_link_3_joint_a"]),
}


def _obj_triangles(path):
    """Triangles of an OBJ file: an n-gon face line counts n - 2 (fan triangulation)."""
    return sum(len(line.split()) - 3 for line in open(path, errors="ignore") if line.startswith("f "))


def _independent_reading(urdf_path, mesh_dir):
    """(non-fixed joint names in depth-first document order from the root link, [(link, faces)] for links with a mesh:
    collision meshes if the link has any, else visual meshes) -- reference hand_model.py:224-235 / pytorch_kinematics'
    get_joint_parameter_names, re-read here with nothing but ElementTree."""
    root = ET.parse(urdf_path).getroot()
    joints = root.findall("joint")
    children = {j.find("child").get("link") for j in joints}
    base = [l.get("name") for l in root.findall("link") if l.get("name") not in children][0]
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.find("parent").get("link"), []).append(j)
    order, links = [], []

    def visit(link):
        links.append(link)
        for j in by_parent.get(link, []):
            if j.get("type") != "fixed":
                order.append(j.get("name"))
            visit(j.find("child").get("link"))

    visit(base)
    elems = {l.get("name"): l for l in root.findall("link")}
    faces = []
    for name in links:
        geoms = elems[name].findall("collision") or elems[name].findall("visual")
        n = 0
        for g in geoms:
            m = g.find("geometry").find("mesh")
            if m is not None:
                n += _obj_triangles(os.path.join(mesh_dir, os.path.basename(m.get("filename"))))
        if n:
            faces.append((name, n))
    return order, faces


@pytest.mark.parametrize("hand", sorted(EXPECT))
def test_hand_spec_face_counts_and_joint_order(hand):
    s = get_hand_spec(hand)
    e = EXPECT[hand]
    assert list(s.joint_names) == e["joints"]
    assert [s.link_faces(l).shape[0] for l in range(s.n_links)] == e["faces"]
    assert (np.asarray(s.node_parent) < np.arange(s.n_dofs)).all()  # parents precede children (DFS order)
    if os.path.isdir(REF_ASSETS):
        order, faces = _independent_reading(os.path.join(REF_ASSETS, e["urdf"]), os.path.join(REF_ASSETS, e["mesh"]))
        assert order == e["joints"]
        assert [n for _, n in faces] == e["faces"] and [l for l, _ in faces] == list(s.link_names)


@pytest.mark.skipif(not os.path.isdir("/root/reference/graspqp/src/graspqp"), reason="needs the reference tree (build container)")
def test_make_golden_reproduces_the_committed_fixtures(tmp_path, golden_dir):
    """tools/make_golden.py executes the reference's own span.py / registry.py / energy.py / optimizer.py in place; its
    output must equal the committed fixtures array for array (npz containers carry zip timestamps, so files are compared
    by content)."""
    src = open(os.path.join(ROOT, "tools", "make_golden.py")).read()
    assert 'OUT = os.path.join(ROOT, "tests", "golden")' in src
    env = dict(os.environ, GRASPQP_GOLDEN_OUT=str(tmp_path))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_golden.py")], env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    names = sorted(f for f in os.listdir(golden_dir) if f.endswith(".npz"))
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")) == names
    for f in names:
        a, b = np.load(os.path.join(golden_dir, f)), np.load(os.path.join(tmp_path, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k], equal_nan=True), (f, k)


def _scene(g, dtype):
    spec = get_hand_spec("allegro")
    hand = models.OracleHand(spec, dtype=dtype)
    n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
    obj = models.OracleObject([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                              [g[f"obj{i}_surface_points"] for i in range(n_obj)], be, dtype=dtype)
    return hand, obj, be


def _replay(g, run, hand, obj, be, clip):
    """Replay run `run` of the extended fixture with the oracle's pure-function MALA* (recorded draws)."""
    dt = torch.float64
    T = lambda k: torch.tensor(g[k])
    hp, idx, energy = T(f"{run}_hand_pose0").to(dt), T(f"{run}_contact_idx0"), T(f"{run}_energy0").to(dt)
    B, D = hp.shape
    ema = torch.zeros(B, D, dtype=dt)
    if run == "R":
        grad = torch.zeros(B, D, dtype=dt)  # fit.py:396
        step = T("R_step0").clone()
    else:
        grad = None
        step = torch.zeros(B, dtype=torch.long)
    after_reset = False
    for s in range(1, int(g[f"{run}_n_steps"]) + 1):
        p = f"{run}_s{s}"
        if f"{p}_grad_in" in g.files:
            grad = T(f"{p}_grad_in").to(dt)
        hp2, idx2, ema, step, ss = mala.propose(hp, grad, ema, step, idx, T(f"{p}_u_switch"), T(f"{p}_new_idx"), clip_grad=clip)
        np.testing.assert_allclose(ss.numpy(), g[f"{p}_step_size"], rtol=1e-6)
        z = mala.z_score(energy, be)
        np.testing.assert_allclose(z.numpy(), g[f"{p}_z"], rtol=1e-5, atol=1e-7)
        rm, reinit = None, False
        if f"{p}_reset_mask" in g.files:  # fit.py:408-422 + optimizer.py:275-287
            rm = T(f"{p}_reset_mask")
            reinit = bool(rm.any())  # fit.py:412: an empty mask re-initialises nothing (accept_step still receives it)
        if reinit:
            hp2 = torch.where(rm[:, None], T(f"{p}_reset_pose").to(dt), hp2)
            idx2 = torch.where(rm[:, None], T(f"{p}_reset_idx"), idx2)
            step = torch.where(rm, torch.zeros_like(step), step)
            ema = torch.where(rm[:, None], torch.zeros_like(ema), ema)
            hp = torch.where(rm[:, None], hp2, hp)
            idx = torch.where(rm[:, None], idx2, idx)
            grad = torch.where(rm[:, None], torch.zeros_like(grad), grad)
        np.testing.assert_allclose(hp2.numpy(), g[f"{p}_prop_pose"], rtol=1e-6, atol=1e-7)
        assert idx2.tolist() == g[f"{p}_prop_idx"].tolist()
        hpr = hp2.clone().requires_grad_()
        # reference quirk (hand_model.py:815-831): with env_mask the contact points of ALL rows are gathered with the
        # indices passed in (initialize_convex_hull draws fresh ones for the whole batch, initializations.py:190-193),
        # while the stored contact_point_indices of the other rows stay the proposal's
        hand.set_parameters(hpr, T(f"{p}_reset_idx") if reinit else idx2)
        new_e = ref_cpu.total_energy(ref_cpu.calculate_energy(hand, obj))
        new_e.sum().backward()
        g2 = hand.hand_pose.grad.detach()
        np.testing.assert_allclose(new_e.detach().numpy(), g[f"{p}_new_energy"], rtol=1e-6)
        acc, Tm = mala.accept(energy, new_e.detach(), step, T(f"{p}_u_accept"), z=z, reset_mask=rm)
        np.testing.assert_allclose(Tm.numpy(), g[f"{p}_temperature"], rtol=1e-5)
        assert acc.tolist() == g[f"{p}_accept"].tolist()
        # reference quirk: set_parameters(env_mask=...) makes hand_pose a LEAF (hand_model.py:846-851); in the following
        # iteration the backward pass accumulates the new gradient IN PLACE into that leaf's .grad -- the very tensor
        # MalaStar keeps as old_grad_hand_pose (optimizer.py:266) -- so rejected rows get old + new gradient back.  (A
        # non-leaf pose with retain_grad, the normal case, accumulates out of place and leaves the kept tensor alone.)
        grad_old = grad + g2 if after_reset else grad
        after_reset = reinit
        hp, idx, grad = mala.merge(acc, hp2, hp), mala.merge(acc, idx2, idx), mala.merge(acc, g2, grad_old)
        energy = mala.merge(acc, new_e.detach(), energy)
        np.testing.assert_allclose(hp.numpy(), g[f"{p}_hand_pose"], rtol=1e-6, atol=1e-7)
        assert idx.tolist() == g[f"{p}_contact_idx"].tolist()
        np.testing.assert_allclose(energy.numpy(), g[f"{p}_energy"], rtol=1e-6)
        gref = g[f"{p}_grad"]
        assert np.linalg.norm(grad.numpy() - gref) <= 1e-5 * np.linalg.norm(gref)
        np.testing.assert_allclose(ema.numpy(), g[f"{p}_ema"], rtol=1e-5, atol=1e-9)
        assert step.tolist() == g[f"{p}_step"].tolist()


def test_oracle_mala_reset_and_decay_match_reference_optimizer(golden_dir):
    g = np.load(os.path.join(golden_dir, "mala_ext_allegro_sphere_b8_n4.npz"), allow_pickle=False)
    hand, obj, be = _scene(g, torch.float64)
    # both decay exponents are non-zero in this run, and they change inside it
    assert float(g["R_s1_step_size"][0]) == pytest.approx(0.005 * 0.95**2, rel=1e-6)
    assert float(g["R_s3_step_size"][int(np.argmin(g["R_s2_reset_mask"]))]) == pytest.approx(0.005 * 0.95**3, rel=1e-6)
    assert g["R_s2_accept"][g["R_s2_reset_mask"]].all() and 0 < g["R_s2_reset_mask"].sum() < len(g["R_s2_accept"])
    # iteration 4: a scheduled re-initialisation with an EMPTY mask (fit.py:412), iteration 5 the one after it -- both are
    # ordinary iterations, and both contain rejected rows (whose restored gradient would show an old + new accumulation)
    assert int(g["R_n_steps"]) == 5 and not g["R_s4_reset_mask"].any()
    assert (~g["R_s4_accept"]).any() and (~g["R_s5_accept"]).any()
    _replay(g, "R", hand, obj, be, clip=False)


def test_oracle_mala_clip_grad_with_nan_inf_matches_reference_optimizer(golden_dir):
    g = np.load(os.path.join(golden_dir, "mala_ext_allegro_sphere_b8_n4.npz"), allow_pickle=False)
    hand, obj, be = _scene(g, torch.float64)
    assert np.isnan(g["C_s1_grad_in"]).any() and np.isinf(g["C_s1_grad_in"]).any()
    assert np.isfinite(g["C_s1_prop_pose"]).all() and np.isfinite(g["C_s1_ema"]).all()
    _replay(g, "C", hand, obj, be, clip=True)


# ---------------------------------------------------------------------------------------------------------------
# (re-)initialisation: oracle/ref_cpu/init.py against the reference's initialize_convex_hull (fixture init_*.npz)
# ---------------------------------------------------------------------------------------------------------------
def test_oracle_initialization_matches_reference_initialize_convex_hull(golden_dir):
    from ref_cpu import init as oinit

    g = np.load(os.path.join(golden_dir, "init_allegro_sq_b12.npz"), allow_pickle=False)
    spec = get_hand_spec("allegro")
    n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
    hulls = [g[f"obj{i}_hull_face_verts"] for i in range(n_obj)]
    draws = {"u_face": torch.tensor(g["u_face"]), "u_len": torch.tensor(g["u_len"]), "u_pose": torch.tensor(g["u_pose"])}
    pose, p, n = oinit.initialize_convex_hull(spec, hulls, be, draws, joints=torch.tensor(g["joints"]))
    # the reference assembles in float32
    np.testing.assert_allclose(pose.numpy(), g["hand_pose"], rtol=2e-5, atol=2e-6)
    assert g["contact_idx"].tolist() == g["randint"].tolist()
    # geometry of what the reference produced: the hand looks at the object from 5-10 cm outside the 1 cm shell
    R = okin_gs(torch.tensor(g["hand_pose"][:, 3:9], dtype=torch.float64))
    fwd_world = R @ torch.tensor(spec.forward_axis, dtype=torch.float64)
    cos_tilt = (fwd_world * n).sum(-1)  # forward axis vs direction to the hull, tilted by at most 45 deg / 15 deg
    assert (cos_tilt > np.cos(np.deg2rad(48))).all()
    d = ((p - torch.tensor(g["hand_pose"][:, :3], dtype=torch.float64)) * n).sum(-1)
    assert ((d > 0.05 - 1e-5) & (d < 0.1 + 1e-5)).all()
    lo, hi = np.asarray(spec.joints_lower), np.asarray(spec.joints_upper)
    assert ((g["joints"] >= lo - 2e-6) & (g["joints"] <= hi + 2e-6)).all()
    # env_mask call (fit.py:421): rows outside the mask untouched
    m = g["env_mask"]
    assert np.array_equal(g["hand_pose_masked"][~m], g["hand_pose"][~m]) and not np.array_equal(g["hand_pose_masked"][m], g["hand_pose"][m])
    assert np.array_equal(g["contact_idx_masked"][~m], g["contact_idx"][~m])


def okin_gs(six):
    from ref_cpu import kin

    return kin.special_gramschmidt(six)


def test_oracle_init_pieces_against_scipy_and_torch():
    """The restated third-party pieces: euler2mat 'rxyz' == scipy's intrinsic XYZ; truncated normal == torch's
    trunc_normal_ on the same uniform stream; farthest points / hull sanity."""
    from scipy.spatial.transform import Rotation

    from ref_cpu import init as oinit

    g = torch.Generator().manual_seed(0)
    ang = (torch.rand(40, 3, generator=g, dtype=torch.float64) - 0.5) * 6
    np.testing.assert_allclose(oinit.euler2mat_rxyz(ang[:, 0], ang[:, 1], ang[:, 2]).numpy(),
                               Rotation.from_euler("XYZ", ang.numpy()).as_matrix(), atol=1e-12)
    for mean, std, a, b in ((0.2, 0.15, -0.3, 0.5), (1.0, 0.05, 0.9, 1.6), (0.0, 0.3, -0.1, 0.1)):
        torch.manual_seed(3)
        want = torch.nn.init.trunc_normal_(torch.empty(1000, dtype=torch.float64), mean, std, a, b)
        torch.manual_seed(3)
        # trunc_normal_ draws tensor.uniform_(2l-1, 2u-1) = from + U[0,1) * (to - from) on the same generator stream
        u = torch.empty(1000, dtype=torch.float64).uniform_(0, 1)
        t64 = lambda v: torch.tensor(v, dtype=torch.float64)
        got = oinit.trunc_normal_from_uniform(u, t64(mean), t64(std), t64(a), t64(b))
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-9, atol=1e-12)
    from graspqp_amd.utils import meshes

    fv = meshes.superquadric(3, 24, 12)
    hull = oinit.convex_hull_faces(fv.reshape(-1, 3))
    nrm = oinit.face_normals(hull).numpy()
    cen = hull.mean(1)
    assert ((cen - hull.reshape(-1, 3).mean(0)) * nrm).sum(1).min() > 0, "hull faces must be oriented outward"
    pts, f = oinit.sample_surface(hull, torch.rand(500, generator=g, dtype=torch.float64), torch.rand(500, 2, generator=g, dtype=torch.float64))
    # samples lie on their faces
    assert np.abs(((pts.numpy() - hull[f.numpy(), 0]) * nrm[f.numpy()]).sum(1)).max() < 1e-12
    sel = oinit.farthest_points(pts, 20)
    assert sel[0] == 0 and len(set(sel.tolist())) == 20


def test_oracle_annealing_dexgraspnet_matches_reference_optimizer(golden_dir):
    """reference AnnealingDexGraspNet + dexgrasp metric (fixture annealing_*.npz, float32 like the reference's metric)
    against the oracle's pure functions: same proposal / acceptance arithmetic as MALA* without the z-score."""
    from ref_cpu import metrics_alt as oalt

    g = np.load(os.path.join(golden_dir, "annealing_dexgrasp_allegro_sphere_b8_n4.npz"), allow_pickle=False)
    dt = torch.float64
    hand, obj, be = _scene(g, dt)
    T = lambda k: torch.tensor(g[k])
    hp, idx, energy = T("hand_pose0").to(dt), T("contact_idx0"), T("energy0").to(dt)
    B, D = hp.shape
    grad, ema = torch.zeros(B, D, dtype=dt), torch.zeros(B, D, dtype=dt)
    step = torch.full((B,), int(g["step0"]), dtype=torch.long)
    fc = lambda p, nrm, cog: oalt.dexgrasp_e_fc(p, nrm, cog, 0.0)
    for s in range(1, int(g["n_steps"]) + 1):
        p = f"s{s}"
        hp2, idx2, ema, step, ss = mala.propose(hp, grad, ema, step, idx, T(f"{p}_u_switch"), T(f"{p}_new_idx"))
        np.testing.assert_allclose(ss.numpy(), g[f"{p}_step_size"], rtol=1e-6)
        np.testing.assert_allclose(hp2.numpy(), g[f"{p}_prop_pose"], rtol=2e-5, atol=2e-6)
        assert idx2.tolist() == g[f"{p}_prop_idx"].tolist()
        hpr = hp2.clone().requires_grad_()
        hand.set_parameters(hpr, idx2)
        lo = ref_cpu.calculate_energy(hand, obj, e_fc_fn=fc)
        np.testing.assert_allclose(lo["E_fc"].detach().numpy(), g[f"{p}_new_E_fc"], rtol=1e-4, atol=1e-5)
        new_e = ref_cpu.total_energy(lo)
        new_e.sum().backward()
        np.testing.assert_allclose(new_e.detach().numpy(), g[f"{p}_new_energy"], rtol=2e-4)
        acc, Tm = mala.accept(energy, new_e.detach(), step, T(f"{p}_u_accept"), z=None)
        np.testing.assert_allclose(Tm.numpy(), g[f"{p}_temperature"], rtol=1e-5)
        assert acc.tolist() == g[f"{p}_accept"].tolist()
        # teacher forcing (the fixture is float32 arithmetic): continue from the reference's accepted state
        hp, idx, grad = T(f"{p}_hand_pose").to(dt), T(f"{p}_contact_idx"), T(f"{p}_grad").to(dt)
        energy, ema = T(f"{p}_energy").to(dt), T(f"{p}_ema").to(dt)
        assert step.tolist() == g[f"{p}_step"].tolist()


def test_closest_face_helper_matches_the_oracle():
    """graspqp_amd/utils/meshes.closest_face (numpy, asset set-up: normals of re-sampled contact candidates) against the
    oracle's closest-point search on meshes of every kind the hands have (closed, several shells, slivers)."""
    from graspqp_amd.utils import meshes
    from ref_cpu import sdf as osdf

    rng = np.random.default_rng(5)
    cases = [meshes.icosphere(2, 0.05), meshes.box((0.02, 0.03, 0.01)), meshes.superquadric(4, 24, 12)]
    sh = get_hand_spec("shadow_hand")
    cases += [sh.link_faces(4), sh.link_faces(16)]  # a distal link and the thumb link whose mesh is several shells
    for fv in cases:
        fv = np.asarray(fv, dtype=np.float64)
        lo, hi = fv.reshape(-1, 3).min(0), fv.reshape(-1, 3).max(0)
        pts = lo + (hi - lo) * (rng.random((300, 3)) * 1.6 - 0.3)  # inside, on and around the mesh
        q, d2, fi = meshes.closest_face(pts, fv)
        d2o, _, _, closest = osdf.compute_sdf(torch.tensor(pts), torch.tensor(fv))
        np.testing.assert_allclose(d2, d2o.numpy(), rtol=1e-9, atol=1e-15)
        # the closest POINT may differ where two faces tie; its distance may not
        np.testing.assert_allclose(((pts - q) ** 2).sum(1), d2, rtol=1e-9, atol=1e-15)
        tie_free = np.abs(q - closest.numpy()).max(1) < 1e-9
        assert tie_free.mean() > 0.95
        # the reported face really contains the closest point: it lies in the face's plane and inside its edges
        a, b, c = fv[fi, 0], fv[fi, 1], fv[fi, 2]
        n = np.cross(b - a, c - a)
        nn = np.linalg.norm(n, axis=1)
        ok = nn > 1e-14
        assert np.abs(((q - a) * n).sum(1)[ok] / nn[ok]).max() < 1e-9


def test_plugin_surface_bench_inputs_are_the_reference_grasp_matrices():
    """tools/plugin_surface.py times QPFunction / SQPLsqSolver on grasp matrices it builds in plain torch; they must be the
    matrices of the reference's span.py (restated and fixture-pinned in oracle/ref_cpu/span.py), or the timed problems would
    not be the force-closure QPs."""
    import importlib.util

    from _scenes import hetero_contacts
    from ref_cpu import span as ospan

    spec = importlib.util.spec_from_file_location("plugin_surface", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools",
                                                                                 "plugin_surface.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    for n, k in ((12, 4), (12, 8), (16, 8)):
        pts, nrm, cog = hetero_contacts(9, n, 3)
        F = ps.grasp_matrix(pts, nrm, cog, k)
        Fo = ospan.grasp_matrix(pts, nrm, cog, 0.2, k)
        assert F.shape == (9, 6, n * k)
        np.testing.assert_allclose(F.numpy(), Fo.numpy(), rtol=1e-12, atol=1e-14)

