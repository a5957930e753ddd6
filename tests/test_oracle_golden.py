"""Pin the oracle (oracle/ref_cpu) against fixtures produced by the reference's own files
(tools/make_golden.py) and against the reference's known-answer test.  CPU only."""
import os

import numpy as np
import pytest
import torch
from scipy.optimize import lsq_linear

import ref_cpu
from ref_cpu import mala, models, qp, span
from graspqp_amd.hands import get_hand_spec

W = ref_cpu.energy.DEFAULT_WEIGHTS


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _scipy_solver(A, b, lo, hi):
    """Exact bounded LSQ (what the reference's ScipyLsqSolver computes) for the span fixtures."""
    xs, vs = [], []
    for Ai, bi in zip(A.detach().numpy().astype(float), b.detach().numpy().astype(float)):
        r = lsq_linear(Ai, bi, bounds=(lo, hi))
        xs.append(r.x)
        vs.append(r.cost)
    return torch.tensor(np.stack(vs), dtype=A.dtype), torch.tensor(np.stack(xs), dtype=A.dtype)


def test_kat_reference_solver_test(golden_dir):
    """reference tests/metrics/test_solver.py:5-27: value ~ 0 (atol 1e-4) for both QP forms, fp32 and fp64."""
    g = _load(golden_dir, "kat_solver.npz")
    assert abs(float(g["value"][0])) < 1e-4  # the reference's scipy answer stored in the fixture
    for dt in (torch.float32, torch.float64):
        A, b = torch.tensor(g["A"], dtype=dt), torch.tensor(g["b"], dtype=dt)
        for box in (False, True):
            v, x = qp.lsq_box_qp(A, b, float(g["min_bound"]), float(g["max_bound"]), box_form=box)
            assert torch.allclose(v, torch.zeros_like(v), atol=1e-4)


@pytest.mark.parametrize("n,k", [(4, 4), (12, 4), (16, 4), (12, 8)])
def test_span_matches_reference(golden_dir, n, k):
    g = _load(golden_dir, f"span_n{n}_k{k}.npz")
    for dt, tol in ((torch.float32, 2e-5), (torch.float64, 2e-5)):  # fixture itself is fp32 arithmetic
        pts = torch.tensor(g["contact_pts"], dtype=dt)
        nrm = torch.tensor(g["contact_normals"], dtype=dt)
        cog = torch.tensor(g["cog"], dtype=dt)
        F = span.grasp_matrix(pts, nrm, cog, mu=float(g["friction"]), k=k)
        np.testing.assert_allclose(F.numpy(), g["F"], rtol=tol, atol=1e-6)
        np.testing.assert_allclose(span.svd_scale(F).numpy(), g["svd"], rtol=1e-3, atol=1e-6)
        e, xs = span.e_fc(pts, nrm, cog, svd_gain=float(g["svd_gain"]), mu=float(g["friction"]), k=k,
                          max_limit=float(g["max_limit"]), solver=_scipy_solver)
        np.testing.assert_allclose(e.numpy(), g["e_fc"], rtol=2e-3, atol=1e-5)


def _scene(g, hand_name, dtype):
    spec = get_hand_spec(hand_name)
    hand = models.OracleHand(spec, dtype=dtype)
    n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
    fvs = [g[f"obj{i}_face_verts"] for i in range(n_obj)]
    sps = [g[f"obj{i}_surface_points"] for i in range(n_obj)]
    obj = models.OracleObject(fvs, sps, be, dtype=dtype)
    return spec, hand, obj


@pytest.mark.parametrize("tag", ["allegro_sphere_b4_n4", "allegro_sq_b6_n12"])
def test_energy_matches_reference_composition(golden_dir, tag):
    """reference core/energy.py run on the same models == oracle calculate_energy (fp64, same QP)."""
    g = _load(golden_dir, f"energy_{tag}.npz")
    spec, hand, obj = _scene(g, "allegro", torch.float64)
    hp = torch.tensor(g["hand_pose"], dtype=torch.float64, requires_grad=True)
    hand.set_parameters(hp, torch.tensor(g["contact_idx"]))
    losses = ref_cpu.calculate_energy(hand, obj)
    for k in ("E_dis", "E_fc", "E_joints", "E_pen", "E_spen"):
        np.testing.assert_allclose(losses[k].detach().numpy(), g[k], rtol=1e-9, atol=1e-12)
    tot = ref_cpu.total_energy(losses)
    np.testing.assert_allclose(tot.detach().numpy(), g["total"], rtol=1e-9)
    tot.sum().backward()
    np.testing.assert_allclose(hand.hand_pose.grad.numpy(), g["grad"], rtol=1e-7, atol=1e-9)
    # the two QP forms agree
    losses_b = ref_cpu.calculate_energy(hand, obj, box_form=True)
    np.testing.assert_allclose(losses_b["E_fc"].detach().numpy(), g["E_fc"], rtol=1e-7)


def test_mala_loop_matches_reference_optimizer(golden_dir):
    """reference core/optimizer.py::MalaStar + fit.py loop order, replayed with the recorded draws."""
    g = _load(golden_dir, "mala_allegro_sphere_b8_n4.npz")
    dt = torch.float64
    spec, hand, obj = _scene(g, "allegro", dt)
    be = int(g["batch_size_each"])
    hp = torch.tensor(g["hand_pose0"], dtype=dt)
    idx = torch.tensor(g["contact_idx0"])
    energy = torch.tensor(g["energy0"], dtype=dt)
    B, D = hp.shape
    grad = torch.zeros(B, D, dtype=dt)  # fit.py:396 zeroes the first gradient
    ema = torch.zeros(B, D, dtype=dt)
    step = torch.zeros(B, dtype=torch.long)
    for s in range(1, int(g["n_steps"]) + 1):
        hp2, idx2, ema, step, ss = mala.propose(
            hp, grad, ema, step, idx, torch.tensor(g[f"s{s}_u_switch"]), torch.tensor(g[f"s{s}_new_idx"]))
        np.testing.assert_allclose(ss.numpy(), g[f"s{s}_step_size"], rtol=1e-6)
        np.testing.assert_allclose(hp2.numpy(), g[f"s{s}_prop_pose"], rtol=1e-6, atol=1e-7)
        z = mala.z_score(energy, be)
        hpr = hp2.clone().requires_grad_()
        hand.set_parameters(hpr, idx2)
        losses = ref_cpu.calculate_energy(hand, obj)
        new_e = ref_cpu.total_energy(losses)
        new_e.sum().backward()
        g2 = hand.hand_pose.grad.detach()
        np.testing.assert_allclose(new_e.detach().numpy(), g[f"s{s}_new_energy"], rtol=1e-6)
        acc, T = mala.accept(energy, new_e.detach(), step, torch.tensor(g[f"s{s}_u_accept"]), z=z)
        np.testing.assert_allclose(T.numpy(), g[f"s{s}_temperature"], rtol=1e-5)
        assert acc.tolist() == g[f"s{s}_accept"].tolist()
        hp = mala.merge(acc, hp2, hp)
        idx = mala.merge(acc, idx2, idx)
        grad = mala.merge(acc, g2, grad)
        energy = mala.merge(acc, new_e.detach(), energy)
        np.testing.assert_allclose(hp.numpy(), g[f"s{s}_hand_pose"], rtol=1e-6, atol=1e-7)
        assert idx.tolist() == g[f"s{s}_contact_idx"].tolist()
        np.testing.assert_allclose(energy.numpy(), g[f"s{s}_energy"], rtol=1e-6)
        gref = g[f"s{s}_grad"]
        assert np.linalg.norm(grad.numpy() - gref) <= 1e-5 * np.linalg.norm(gref)
        np.testing.assert_allclose(ema.numpy(), g[f"s{s}_ema"], rtol=1e-5, atol=1e-9)
