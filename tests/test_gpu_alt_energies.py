"""GPU parity of the reference's other energy types and optimizer (scripts/fit.py:335-347): dexgrasp / tdg force-closure
energies against fixtures produced by the reference's own metrics/ops/dexgrasp.py and tdg.py (energy 2e-4 rel, gradient
5e-3 norm-wise: fp32 acos / normalisations), the stepper with those energy types against the oracle composition, and
AnnealingDexGraspNet against the fixture of the reference class (teacher-forced)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import ref_cpu  # noqa: E402
from ref_cpu import metrics_alt as oalt  # noqa: E402
from ref_cpu import models as omodels  # noqa: E402

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


@pytest.mark.parametrize("n", [4, 12, 20])
def test_dexgrasp_and_tdg_match_reference_modules(gq, golden_dir, n):
    from graspqp_amd.metrics import DexgraspSpanMetric, TDGSpanMetric

    g = _load(golden_dir, "alt_metrics.npz")
    t = f"n{n}"
    pts, nrm, cog = (torch.tensor(g[f"{t}_{k}"]).cuda() for k in ("contact_pts", "contact_normals", "cog"))
    dex = DexgraspSpanMetric()
    for tw in (0, 1, 5):
        p = pts.clone().requires_grad_()
        e, ones = dex(p, nrm, cog, torque_weight=float(tw), with_solution=True)
        assert ones.shape == (pts.shape[0], n) and bool((ones == 1).all())
        np.testing.assert_allclose(e.detach().cpu().numpy(), g[f"{t}_dex_tw{tw}_e"], rtol=2e-5, atol=1e-6)
        e.sum().backward()
        gref = g[f"{t}_dex_tw{tw}_grad"]
        assert np.linalg.norm(p.grad.cpu().numpy() - gref) <= 1e-4 * np.linalg.norm(gref) + 1e-7
    m = TDGSpanMetric(device="cuda", directions=g[f"{t}_tdg_directions"])
    p = pts.clone().requires_grad_()
    e, none = m(p, nrm, cog)
    assert none is None
    np.testing.assert_allclose(e.detach().cpu().numpy(), g[f"{t}_tdg_e"], rtol=2e-4)
    e.sum().backward()
    gref = g[f"{t}_tdg_grad"]
    assert np.linalg.norm(p.grad.cpu().numpy() - gref) <= 5e-3 * np.linalg.norm(gref)
    # upstream weights and bitwise reproducibility
    p2 = pts.clone().requires_grad_()
    w = torch.rand(pts.shape[0], device="cuda")
    (m(p2, nrm, cog)[0] * w).sum().backward()
    torch.testing.assert_close(p2.grad, p.grad * w.view(-1, 1, 1), rtol=1e-6, atol=1e-9)
    assert torch.equal(m(pts, nrm, cog)[0], e.detach())


@pytest.mark.parametrize("energy_type", ["dexgrasp", "tdg"])
def test_stepper_with_other_energy_types(gq, energy_type):
    """Energy + gradient of the whole composition (E_dis, E_fc of the chosen type, E_pen, E_spen, E_joints) against the
    fp64 oracle, the class-surface route through GraspSpanMetricFactory, and graph replay == eager."""
    from graspqp_amd.core.energy import calculate_energy
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF

    spec = get_hand_spec("allegro")
    n_obj, be, n = 2, 6, 12
    B = n_obj * be
    fvs = [meshes.superquadric(5 + i, 32, 16) for i in range(n_obj)]
    sps = [meshes.surface_points(f, 600, oversample=4, seed=3 + i) for i, f in enumerate(fvs)]
    g0 = torch.Generator().manual_seed(31)
    t = torch.nn.functional.normalize(torch.randn(B, 3, generator=g0, dtype=torch.float64), dim=-1) * 0.1
    hp = torch.cat([t, torch.randn(B, 6, generator=g0, dtype=torch.float64),
                    torch.tensor(spec.default_state, dtype=torch.float64)[None] + 0.3 * torch.randn(B, spec.n_dofs, generator=g0, dtype=torch.float64)], 1)
    idx = torch.randint(spec.n_contact_candidates, (B, n), generator=g0)
    dirs = torch.nn.functional.normalize(torch.randn(1000, 3, generator=g0), dim=-1)
    hand = gq.ops.HandHandle(spec)
    st = gq.stepper.GraspStepper(hand, gq.ops.MeshSet(fvs), torch.tensor(np.stack(sps)), be, n, energy_type=energy_type,
                                 tdg_directions=dirs, seed=5)
    terms, total, grad = st.evaluate(hp.float().cuda(), idx.cuda())
    torch.cuda.synchronize()
    oh = omodels.OracleHand(spec, torch.float64)
    oo = omodels.OracleObject(fvs, sps, be, torch.float64)
    hpo = hp.clone().requires_grad_()
    oh.set_parameters(hpo, idx)
    if energy_type == "dexgrasp":
        fc = lambda p, nrm, cog: oalt.dexgrasp_e_fc(p, nrm, cog, 0.0)
    else:
        fc = lambda p, nrm, cog: oalt.tdg_energy(p, nrm, cog, dirs.double())
    lo = ref_cpu.calculate_energy(oh, oo, e_fc_fn=fc)
    tot = ref_cpu.total_energy(lo)
    tot.sum().backward()
    for k in ("E_dis", "E_fc", "E_pen", "E_spen", "E_joints"):
        np.testing.assert_allclose(terms[k].cpu().numpy(), lo[k].detach().numpy(), rtol=3e-4, atol=3e-6, err_msg=k)
    np.testing.assert_allclose(total.cpu().numpy(), tot.detach().numpy(), rtol=2e-4)
    go = oh.hand_pose.grad.numpy()
    assert np.linalg.norm(grad.cpu().numpy() - go) <= 5e-3 * np.linalg.norm(go)  # fp32 E_pen gradient noise, as in the golden energy test
    # class-surface route with the factory's metric (fit.py:343-347)
    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=be, num_samples=600)
    om.initialize_from_meshes(fvs, surface_points_list=sps)
    hm.set_parameters(hp.float().cuda().requires_grad_(), idx.cuda())
    fn = GF.create(GF.MetricType.DEXGRASP if energy_type == "dexgrasp" else GF.MetricType.TDG)
    if energy_type == "tdg":
        fn.target_direction = dirs.cuda().contiguous()
    W = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=list(W), svd_gain=0.1)
    tot2 = sum(W[k] * v for k, v in losses.items())
    tot2.sum().backward()
    np.testing.assert_allclose(tot2.detach().cpu().numpy(), total.cpu().numpy(), rtol=2e-5)
    assert (hm.hand_pose.grad - grad).norm() <= 1e-3 * grad.norm()
    # iterations: graph replay == eager, finite
    outs = []
    for rep in range(3):
        s2 = gq.stepper.GraspStepper(hand, gq.ops.MeshSet(fvs), torch.tensor(np.stack(sps)), be, n, energy_type=energy_type,
                                     tdg_directions=dirs, seed=5)
        s2.reset(hp.float().cuda(), idx.cuda())
        if rep == 1:  # the fused four-launch form (gq_alt_pen_step): contact terms + this energy beside the penetration query
            s2.capture(iters=2)
            assert s2.graph_mode == "one grid"
        if rep == 2:  # per-role launches on two graph branches (what batches >= 384 rows take)
            s2.capture(fork=True, iters=2)
            assert s2.graph_mode == "graph branches"
        for _ in range(4):
            s2.step()
        s2.flush()
        torch.cuda.synchronize()
        outs.append((s2.energy.clone(), s2.hand_pose.clone(), s2.contact_idx.clone()))
    assert torch.isfinite(outs[0][0]).all()
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])) and all(torch.equal(a, b) for a, b in zip(outs[0], outs[2]))


def test_annealing_dexgraspnet_matches_reference_optimizer(gq, golden_dir):
    """GraspStepper(optimizer="dexgraspnet", energy_type="dexgrasp") teacher-forced against the fixture of the reference's
    AnnealingDexGraspNet + DexgraspSpanMetric run in fit.py order (started at step 61)."""
    g = _load(golden_dir, "annealing_dexgrasp_allegro_sphere_b8_n4.npz")
    spec = get_hand_spec("allegro")
    n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
    fvs = [g[f"obj{i}_face_verts"] for i in range(n_obj)]
    sps = np.stack([g[f"obj{i}_surface_points"] for i in range(n_obj)])
    st = gq.stepper.GraspStepper(gq.ops.HandHandle(spec), gq.ops.MeshSet(fvs), torch.tensor(sps), be, 4,
                                 energy_type="dexgrasp", optimizer="dexgraspnet")
    f32 = lambda k: torch.tensor(g[k], dtype=torch.float32).cuda()
    st.reset(f32("hand_pose0"), torch.tensor(g["contact_idx0"]).cuda())
    np.testing.assert_allclose(st.energy.cpu().numpy(), g["energy0"], rtol=3e-4)
    np.testing.assert_allclose(st.terms[1].cpu().numpy(), g["E_fc0"], rtol=1e-4, atol=1e-5)
    st.energy.copy_(f32("energy0"))
    st.step_count.fill_(int(g["step0"]))
    rel = lambda a, b: np.abs(np.asarray(a, dtype=np.float64) - b) / np.maximum(np.abs(b), 1e-12)
    for s in range(1, int(g["n_steps"]) + 1):
        p = f"s{s}"
        if s > 1:
            q = f"s{s-1}"
            st.hand_pose.copy_(f32(f"{q}_hand_pose"))
            st.contact_idx.copy_(torch.tensor(g[f"{q}_contact_idx"]).cuda())
            st.grad.copy_(f32(f"{q}_grad"))
            st.energy.copy_(f32(f"{q}_energy"))
            st.ema.copy_(f32(f"{q}_ema"))
            st.step_count.copy_(torch.tensor(g[f"{q}_step"]).cuda())
        st.step(draws=(f32(f"{p}_u_switch"), torch.tensor(g[f"{p}_new_idx"]).cuda(), f32(f"{p}_u_accept")))
        torch.cuda.synchronize()
        np.testing.assert_allclose(st.s_out.cpu().numpy(), g[f"{p}_step_size"], rtol=1e-5)
        np.testing.assert_allclose(st.pose_new.cpu().numpy(), g[f"{p}_prop_pose"], rtol=2e-5, atol=3e-6)
        assert st.idx_new.cpu().tolist() == g[f"{p}_prop_idx"].tolist()
        np.testing.assert_allclose(st.terms_new[1].cpu().numpy(), g[f"{p}_new_E_fc"], rtol=2e-4, atol=1e-5)
        assert rel(st.total_new.cpu().numpy(), g[f"{p}_new_energy"]).max() < 5e-4
        np.testing.assert_allclose(st.temperature.cpu().numpy(), g[f"{p}_temperature"], rtol=1e-4)  # no (1 + Phi(z)) factor
        assert st.accept.cpu().bool().tolist() == g[f"{p}_accept"].tolist()
        np.testing.assert_allclose(st.hand_pose.cpu().numpy(), g[f"{p}_hand_pose"], rtol=2e-5, atol=3e-6)
        assert st.contact_idx.cpu().tolist() == g[f"{p}_contact_idx"].tolist()
        np.testing.assert_allclose(st.ema.cpu().numpy(), g[f"{p}_ema"], rtol=2e-4, atol=1e-7)
        assert st.step_count.cpu().tolist() == g[f"{p}_step"].tolist()


@pytest.mark.parametrize("tag,n", [("allegro_sphere_b4_n4", 4), ("allegro_sq_b6_n12", 12)])
def test_optional_energy_terms_match_reference(gq, golden_dir, tag, n):
    """E_prior and E_wall (core/energy.py:68-78, selectable with --w_prior / --w_wall) through calculate_energy against
    the fixture produced by the reference's own energy.py; E_manipulativity (fit.py cannot select it) in value and gradient
    against the oracle's restatement."""
    from graspqp_amd.core.energy import calculate_energy
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF

    g = _load(golden_dir, f"energy_{tag}.npz")
    spec = get_hand_spec("allegro")
    n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
    hm = HandModel(spec, "cuda")
    hm.set_surface_points(g["opt_surface_points"], g["opt_surface_link"])
    om = ObjectModel(batch_size_each=be, num_samples=g["obj0_surface_points"].shape[0])
    om.initialize_from_meshes([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                              surface_points_list=[g[f"obj{i}_surface_points"] for i in range(n_obj)])
    hp = torch.tensor(g["opt_hand_pose"], dtype=torch.float32).cuda().requires_grad_()
    hm.set_parameters(hp, torch.tensor(g["contact_idx"]).cuda())
    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
    names = ["E_dis", "E_fc", "E_pen", "E_spen", "E_joints", "E_prior", "E_wall", "E_manipulativity"]
    losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=names, svd_gain=0.1)
    np.testing.assert_allclose(losses["E_prior"].detach().cpu().numpy(), g["opt_E_prior"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(losses["E_wall"].detach().cpu().numpy(), g["opt_E_wall"], rtol=1e-5, atol=1e-6)
    assert losses["E_manipulativity"].shape == (n_obj * be,) and torch.isfinite(losses["E_manipulativity"]).all()
    # its VALUE against the oracle's restatement of core/energy.py:80-87 (contact Jacobian J_v + J_w x r, damped
    # pseudo-inverse with lambda = 1e-3, mean squared residual of the contact velocities that the joints cannot produce)
    from ref_cpu import export as oexp

    oh = omodels.OracleHand(spec, torch.float64)
    oo = omodels.OracleObject([g[f"obj{i}_face_verts"] for i in range(n_obj)],
                              [g[f"obj{i}_surface_points"] for i in range(n_obj)], be, torch.float64)
    oh.set_parameters(torch.tensor(g["opt_hand_pose"], dtype=torch.float64), torch.tensor(g["contact_idx"]))
    dist_o, cn_o = oo.cal_distance(oh.contact_points)
    _, res_o = oexp.get_req_joint_velocities(oh, cn_o * dist_o.unsqueeze(-1).abs().clamp(min=5e-3), oh.contact_point_indices)
    np.testing.assert_allclose(losses["E_manipulativity"].detach().cpu().numpy(), res_o.mean(-1).numpy(), rtol=2e-3, atol=1e-9)
    # ... and its GRADIENT (joint angles through the contact Jacobian -- kinematic Hessian --, root rotation through R' d,
    # contact points through |distance|) against autograd through the oracle's restatement in fp64, under a non-uniform
    # upstream; the reference gets it from autograd through pytorch_kinematics (core/energy.py:80-87)
    up = torch.linspace(0.5, 2.0, n_obj * be)
    (gm,) = torch.autograd.grad((losses["E_manipulativity"] * up.cuda()).sum(), hm.hand_pose, retain_graph=True)
    hm.hand_pose.grad = None  # (hand_pose retains its gradient: the call above left this term's there)
    oh2 = omodels.OracleHand(spec, torch.float64)
    hp64 = torch.tensor(g["opt_hand_pose"], dtype=torch.float64).requires_grad_()
    oh2.set_parameters(hp64, torch.tensor(g["contact_idx"]))
    dist2, cn2 = oo.cal_distance(oh2.contact_points)
    _, res2 = oexp.get_req_joint_velocities(oh2, cn2.detach() * dist2.unsqueeze(-1).abs().clamp(min=5e-3), oh2.contact_point_indices)
    (go,) = torch.autograd.grad((res2.mean(-1) * up.double()).sum(), hp64)
    gm, go = gm.cpu().double().numpy(), go.numpy()
    assert np.linalg.norm(go) > 0 and np.linalg.norm(go[:, 9:]) > 0 and np.linalg.norm(go[:, 3:9]) > 0
    print("E_manipulativity gradient rel err:", np.linalg.norm(gm - go) / np.linalg.norm(go),
          "joints", np.linalg.norm(gm[:, 9:] - go[:, 9:]) / np.linalg.norm(go[:, 9:]),
          "rotation", np.linalg.norm(gm[:, 3:9] - go[:, 3:9]) / np.linalg.norm(go[:, 3:9]),
          "translation", np.linalg.norm(gm[:, :3] - go[:, :3]) / max(np.linalg.norm(go[:, :3]), 1e-30))
    assert np.linalg.norm(gm - go) <= 1e-3 * np.linalg.norm(go)
    (2.0 * losses["E_prior"] + 3.0 * losses["E_wall"]).sum().backward()
    gref = g["opt_grad"]
    assert np.linalg.norm(hm.hand_pose.grad.cpu().numpy() - gref) <= 1e-4 * np.linalg.norm(gref)
    # the default hand surface samples: n_surface_points of them, on the link meshes, spread over the links by area
    hm2 = HandModel(spec, "cuda")
    hm2.set_parameters(hp.detach(), torch.tensor(g["contact_idx"]).cuda())
    sp = hm2.get_surface_points()
    assert sp.shape == (n_obj * be, 512, 3) and torch.isfinite(sp).all()


@pytest.mark.parametrize("hand_name", ["allegro", "shadow_hand", "ability_hand", "panda", "schunk2"])
def test_joint_velocity_residuals_gradient(gq, hand_name):
    """get_req_joint_velocities (hand_model.py:1155-1218, coupled form) is differentiable: residuals = (J theta - R'd)^2 with the
    damped pseudo-inverse, gradient to the joint angles through the contact Jacobian (closed-form kinematic Hessian,
    gq_contact_jacobian_backward -- revolute and PRISMATIC joints, COUPLED tree joints folded with C), to the root rotation
    through R'd and to the directions.  Against autograd through the oracle's restatement in fp64, random upstream."""
    from graspqp_amd.core.hand_model import HandModel
    from ref_cpu import export as oexp

    spec = get_hand_spec(hand_name)
    B, n = 6, 3 if hand_name in ("panda", "schunk2") else 5
    g0 = torch.Generator().manual_seed(23)
    t = torch.randn(B, 3, generator=g0, dtype=torch.float64) * 0.1
    lo, hi = torch.tensor(spec.joints_lower, dtype=torch.float64), torch.tensor(spec.joints_upper, dtype=torch.float64)
    th = lo + (hi - lo) * torch.rand(B, spec.n_dofs, generator=g0, dtype=torch.float64)
    hp = torch.cat([t, torch.randn(B, 6, generator=g0, dtype=torch.float64), th], 1)
    idx = torch.randint(spec.n_contact_candidates, (B, n), generator=g0)
    md = torch.randn(B, n, 3, generator=g0, dtype=torch.float64) * 0.02
    up = torch.rand(B, 3 * n, generator=g0, dtype=torch.float64) + 0.5
    hm = HandModel(spec, "cuda")
    hm.set_parameters(hp.float().cuda().requires_grad_(), idx.cuda())
    mdc = md.float().cuda().requires_grad_()
    theta, res = hm.get_req_joint_velocities(mdc, idx.cuda())
    assert res.requires_grad and not theta.requires_grad and res.shape == (B, 3 * n)
    (res * up.float().cuda()).sum().backward()
    oh = omodels.OracleHand(spec, torch.float64)
    hpo, mdo = hp.clone().requires_grad_(), md.clone().requires_grad_()
    oh.set_parameters(hpo, idx)
    th_o, res_o = oexp.get_req_joint_velocities(oh, mdo, idx)
    (res_o * up).sum().backward()
    np.testing.assert_allclose(theta.detach().cpu().numpy(), th_o.detach().numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(res.detach().cpu().numpy(), res_o.detach().numpy(), rtol=2e-3, atol=1e-9)
    gp, go = hm.hand_pose.grad.cpu().double().numpy(), hpo.grad.numpy()
    assert np.abs(go[:, :3]).max() == 0 and np.abs(gp[:, :3]).max() == 0          # no dependence on the root translation
    assert np.linalg.norm(go[:, 3:9]) > 0
    if hand_name in ("panda", "schunk2"):  # prismatic fingers only: the contact Jacobian does not depend on the joint values
        assert np.abs(go[:, 9:]).max() == 0 and np.abs(gp[:, 9:]).max() == 0
    else:
        assert np.linalg.norm(go[:, 9:]) > 0
        assert np.linalg.norm(gp[:, 9:] - go[:, 9:]) <= 2e-3 * np.linalg.norm(go[:, 9:]), "joint angles"
    assert np.linalg.norm(gp[:, 3:9] - go[:, 3:9]) <= 2e-3 * np.linalg.norm(go[:, 3:9]), "root rotation"
    gd, gdo = mdc.grad.cpu().double().numpy(), mdo.grad.numpy()
    assert np.linalg.norm(gd - gdo) <= 2e-3 * np.linalg.norm(gdo), "directions"
    # without gradients asked for, the plain (value) route answers the same numbers
    with torch.no_grad():
        th2, res2 = hm.get_req_joint_velocities(md.float().cuda(), idx.cuda())
    assert torch.equal(th2, theta.detach()) and torch.equal(res2, res.detach())


def test_mixed_revolute_prismatic_tree(gq, tmp_path):
    """A synthetic hand whose chains MIX joint types (none of the reference's hands does: they are all-revolute or all-
    prismatic): base -r(z)-> a -p(x)-> b -r(y)-> c, and a second branch a -r(x)-> d -p(z)-> e, built from a URDF by
    hands/spec.py.  Forward kinematics, contact Jacobian, and the gradient of the joint-velocity residuals (every branch
    of the closed-form kinematic Hessian: a prismatic joint above / below a revolute one and vice versa) against the
    oracle's restatement and autograd through it in fp64."""
    import json

    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.hands.spec import build_hand_spec
    from ref_cpu import export as oexp

    box = meshes.box() * 0.02  # (12,3,3) triangles of a small box
    with open(tmp_path / "box.obj", "w") as f:
        v = box.reshape(-1, 3)
        for p in v:
            f.write(f"v {p[0]} {p[1]} {p[2]}\n")
        for i in range(len(box)):
            f.write(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n")
    link = lambda n: (f'<link name="{n}"><collision><origin xyz="0.01 0 0.005" rpy="0 0 0.3"/><geometry><mesh filename="box.obj"/>'
                      f'</geometry></collision></link>')
    joint = lambda n, t, par, ch, xyz, rpy, ax, lo, hi: (
        f'<joint name="{n}" type="{t}"><parent link="{par}"/><child link="{ch}"/><origin xyz="{xyz}" rpy="{rpy}"/>'
        f'<axis xyz="{ax}"/><limit lower="{lo}" upper="{hi}"/></joint>')
    urdf = ('<robot name="mixed">' + "".join(link(n) for n in ("base", "a", "b", "c", "d", "e"))
            + joint("j_a", "revolute", "base", "a", "0.03 0 0.01", "0 0.2 0", "0 0 1", -1.0, 1.0)
            + joint("j_b", "prismatic", "a", "b", "0.04 0.01 0", "0.1 0 0", "1 0 0", -0.02, 0.03)
            + joint("j_c", "revolute", "b", "c", "0.03 0 0.02", "0 0 0.4", "0 1 0", -1.2, 1.2)
            + joint("j_d", "revolute", "a", "d", "0 0.04 0", "0.3 0 0", "1 0 0", -0.8, 0.8)
            + joint("j_e", "prismatic", "d", "e", "0.01 0.03 0", "0 0.2 0.1", "0 0 1", -0.01, 0.02) + "</robot>")
    (tmp_path / "mixed.urdf").write_text(urdf)
    rng = np.random.default_rng(4)
    cinfo = {n: {"contact_candidates": (rng.normal(size=(3, 3)) * 0.01).tolist(),
                 "normal_candidates": np.tile([[0.0, 0.0, 1.0]], (3, 1)).tolist()} for n in ("b", "c", "d", "e")}
    (tmp_path / "cinfo.json").write_text(json.dumps(cinfo))
    (tmp_path / "pen.json").write_text("{}")
    spec = build_hand_spec("mixed", str(tmp_path / "mixed.urdf"), str(tmp_path), str(tmp_path / "pen.json"),
                           str(tmp_path / "cinfo.json"), default_state=[0.0] * 5)
    assert spec.n_dofs == 5 and spec.n_contact_candidates == 12
    B, n = 7, 6
    g0 = torch.Generator().manual_seed(3)
    lo, hi = torch.tensor(spec.joints_lower, dtype=torch.float64), torch.tensor(spec.joints_upper, dtype=torch.float64)
    th = lo + (hi - lo) * torch.rand(B, 5, generator=g0, dtype=torch.float64)
    hp = torch.cat([torch.randn(B, 3, generator=g0, dtype=torch.float64) * 0.1, torch.randn(B, 6, generator=g0, dtype=torch.float64), th], 1)
    idx = torch.randint(12, (B, n), generator=g0)
    md = torch.randn(B, n, 3, generator=g0, dtype=torch.float64) * 0.02
    up = torch.rand(B, 3 * n, generator=g0, dtype=torch.float64) + 0.5
    hm = HandModel(spec, "cuda")
    hm.set_parameters(hp.float().cuda().requires_grad_(), idx.cuda())
    oh = omodels.OracleHand(spec, torch.float64)
    hpo, mdo = hp.clone().requires_grad_(), md.clone().requires_grad_()
    oh.set_parameters(hpo, idx)
    np.testing.assert_allclose(hm.contact_points.detach().cpu().numpy(), oh.contact_points.detach().numpy(), rtol=1e-5, atol=2e-7)
    jc = gq.ops.contact_jacobian(hm._hand, idx.cuda(), hm.current_status, hm._fk_ws).cpu().numpy()
    np.testing.assert_allclose(jc, oexp.contact_jacobian(spec, th, idx).numpy(), rtol=1e-4, atol=2e-7)
    mdc = md.float().cuda().requires_grad_()
    theta, res = hm.get_req_joint_velocities(mdc, idx.cuda())
    (res * up.float().cuda()).sum().backward()
    th_o, res_o = oexp.get_req_joint_velocities(oh, mdo, idx)
    (res_o * up).sum().backward()
    np.testing.assert_allclose(res.detach().cpu().numpy(), res_o.detach().numpy(), rtol=2e-3, atol=1e-9)
    gp, go = hm.hand_pose.grad.cpu().double().numpy(), hpo.grad.numpy()
    assert (np.abs(go[:, 9:]).max(0) > 0).all()  # every joint, prismatic ones included, moves some Jacobian column
    assert np.linalg.norm(gp[:, 9:] - go[:, 9:]) <= 2e-3 * np.linalg.norm(go[:, 9:]), "joint values"
    assert np.linalg.norm(gp[:, 3:9] - go[:, 3:9]) <= 2e-3 * np.linalg.norm(go[:, 3:9]), "root rotation"
    assert np.linalg.norm(mdc.grad.cpu().double().numpy() - mdo.grad.numpy()) <= 2e-3 * np.linalg.norm(mdo.grad.numpy())


# ---------------------------------------------------------------------------------------------------------------
# coupled-joint hands and grasp-type subsets
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hand_name,grasp_type", [("ability_hand", None), ("panda", None), ("allegro", "pinch"),
                                                  ("ability_hand", "precision"), ("shadow_hand", "pinch"), ("schunk2", None)])
def test_coupled_hands_and_grasp_types_whole_iteration(gq, hand_name, grasp_type):
    """Energy + gradient of the whole composition for hands whose tree joints follow fewer actuated ones (ability_hand:
    q2 = 1.0585 q1 on four fingers; panda: both fingers on one value; schunk2: (theta, -theta) prismatic fingers, STL + COLLADA
    link meshes, no penetration spheres -- reference hands/ability_hand.py, panda.py, schunk.py) and
    for grasp-type contact subsets, against the fp64 oracle; then iterations (graph == eager)."""
    spec = get_hand_spec(hand_name, grasp_type=grasp_type)
    n_obj, be, n = 2, 5, 4 if hand_name in ("panda", "schunk2") else 12
    B = n_obj * be
    fvs = [meshes.superquadric(9 + i, 32, 16) for i in range(n_obj)]
    sps = [meshes.surface_points(f, 600, oversample=4, seed=3 + i) for i, f in enumerate(fvs)]
    g0 = torch.Generator().manual_seed(17)
    t = torch.nn.functional.normalize(torch.randn(B, 3, generator=g0, dtype=torch.float64), dim=-1) * 0.09
    th = torch.tensor(spec.default_state, dtype=torch.float64)[None] + 0.2 * torch.randn(B, spec.n_dofs, generator=g0, dtype=torch.float64)
    th[0] = torch.tensor(spec.joints_upper, dtype=torch.float64) + 0.05  # E_joints active on the actuated joints
    hp = torch.cat([t, torch.randn(B, 6, generator=g0, dtype=torch.float64), th], 1)
    idx = torch.randint(spec.n_contact_candidates, (B, n), generator=g0)
    assert hp.shape[1] == 9 + spec.n_dofs and spec.n_nodes >= spec.n_dofs
    hand = gq.ops.HandHandle(spec)
    st = gq.stepper.GraspStepper(hand, gq.ops.MeshSet(fvs), torch.tensor(np.stack(sps)), be, n, seed=5)
    terms, total, grad = st.evaluate(hp.float().cuda(), idx.cuda())
    torch.cuda.synchronize()
    oh = omodels.OracleHand(spec, torch.float64)
    oo = omodels.OracleObject(fvs, sps, be, torch.float64)
    hpo = hp.clone().requires_grad_()
    oh.set_parameters(hpo, idx)
    w0 = {"E_dis": 100.0, "E_fc": 0.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    lo = ref_cpu.calculate_energy(oh, oo, box_form=True)
    for k in ("E_dis", "E_pen", "E_spen", "E_joints"):
        np.testing.assert_allclose(terms[k].cpu().numpy(), lo[k].detach().numpy(), rtol=3e-4, atol=3e-6, err_msg=k)
    assert float(lo["E_joints"][0]) > 0
    rel = np.abs(terms["E_fc"].cpu().numpy() - lo["E_fc"].detach().numpy()) / np.abs(lo["E_fc"].detach().numpy())
    assert np.median(rel) < 1e-4 and rel.max() < 5e-3
    s0 = gq.stepper.GraspStepper(hand, gq.ops.MeshSet(fvs), torch.tensor(np.stack(sps)), be, n, weights=w0)
    _, _, g0_ = s0.evaluate(hp.float().cuda(), idx.cuda())
    sum(w0[k] * lo[k] for k in w0 if w0[k] != 0.0).sum().backward()
    go = oh.hand_pose.grad.numpy()
    assert np.linalg.norm(g0_.cpu().numpy() - go) <= 5e-3 * np.linalg.norm(go)
    outs = []
    for rep in range(2):
        s2 = gq.stepper.GraspStepper(hand, gq.ops.MeshSet(fvs), torch.tensor(np.stack(sps)), be, n, seed=5)
        s2.reset(hp.float().cuda(), idx.cuda())
        if rep == 1:
            s2.capture(iters=2)
        for _ in range(4):
            s2.step()
        s2.flush()
        torch.cuda.synchronize()
        outs.append((s2.energy.clone(), s2.hand_pose.clone(), s2.contact_idx.clone()))
    assert torch.isfinite(outs[0][0]).all() and all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert int(outs[0][2].max()) < spec.n_contact_candidates
