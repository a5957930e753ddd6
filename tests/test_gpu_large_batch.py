"""GPU parity at the batch sizes of BASELINE configs 3 / 4 / 5 (per rank), which take a different launch sequence than
config 2: for more than 512 rows the contact queries of the object SDF get their own launch (gq_sdf_forward_meshset
instead of riding in the FK block), above 256 rows qpth's batch-global stop rule runs as a launch of its own (one
wavefront up to 1024 rows, the tiled 1024-thread kernel above), and nz = 96 (8-edge cones) puts two QP columns on a lane.

Also here: the stop rule itself on arbitrary residual tables (bit-exact against oracle/ref_cpu/qp.py::stop_rule, with the
deciding row in the LAST tile), a heterogeneous >= 3000-row QP batch against the oracle's PDIPM, and a scene that
overflows the LDS item list of the hand-penetration query on purpose.  Tolerances: tests/test_gpu_parity.py header."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import ref_cpu  # noqa: E402
from ref_cpu import models as omodels  # noqa: E402
from ref_cpu import qp as oqp  # noqa: E402
from ref_cpu import span as ospan  # noqa: E402

from _parity import assert_tail_within_fp32_noise, oracle_fc_fp32_noise  # noqa: E402
from _scenes import hetero_contacts  # noqa: E402
from graspqp_amd.hands import get_hand_spec  # noqa: E402
from graspqp_amd.utils import meshes  # noqa: E402


@pytest.fixture(scope="module")
def gq():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from graspqp_amd import _C, ops, stepper

    _C.lib()
    return type("gq", (), {"ops": ops, "C": _C, "stepper": stepper})


def _rel(a, b, floor=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


# ---------------------------------------------------------------------------------------------------------------
# whole iteration at 2048 / 4096 / 8192 rows
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hand_name,n,k,n_obj,be", [
    ("allegro", 12, 4, 8, 256),       # BASELINE configs[3] per rank: 8 meshes x 256 = 2048 rows
    ("shadow_hand", 16, 4, 8, 512),   # configs[2]: 8 meshes x 512 = 4096 rows, nz = 64
    ("robotiq3", 12, 8, 8, 1024),     # configs[4] share: 8 meshes x 1024 = 8192 rows, 8-edge cones, nz = 96
    ("allegro", 12, 4, 3, 171),       # 513 rows: the last force-closure head block holds ONE row (stop-rule epilogue)
    ("robotiq3", 12, 8, 32, 1024),    # configs[4] per rank at its stated size: 32 meshes x 1024 = 32 768 rows
    ("allegro", 12, 8, 32, 1024),     # ... and its Allegro half (16 joints, 14 links, 25 penetration spheres, nz = 96)
])
def test_stepper_large_batch_launch_sequence(gq, hand_name, n, k, n_obj, be):
    from bench import make_initial_state

    spec = get_hand_spec(hand_name)
    fvs = [meshes.superquadric(o) for o in range(n_obj)]  # the bench's meshes (9024 faces each)
    sps = [meshes.surface_points(f, 2500, oversample=4, seed=42) for f in fvs]
    B = n_obj * be
    hand = gq.ops.HandHandle(spec)
    ms = gq.ops.MeshSet(fvs)
    surf = torch.tensor(np.stack(sps))
    hps, idxs = zip(*[make_initial_state(spec, f, be, n, 1000 + o) for o, f in enumerate(fvs)])
    hp, idx = torch.cat(hps).cuda(), torch.cat(idxs).cuda()
    # a few rows pushed into their object so that E_pen and the inside branch of E_dis are non-trivial
    hp[::97, :3] *= 0.45
    fc_cfg = {"n_cone_vecs": k}

    # (a) captured graphs (default for this size = per-role launches on two graph branches; then both roles in one grid)
    # == eager unfused launches, bit for bit, over several iterations
    outs = []
    lean = hand_name == "allegro" and B >= 32768  # second full-size case: keep the suite's run time in check (the forced one-grid
    # capture and the row sample (c) are covered by the Robotiq-3F case of the same size and by the smaller Allegro cases)
    for rep in range(2 if lean else 3):
        s = gq.stepper.GraspStepper(hand, ms, surf, be, n, fc_cfg=fc_cfg, seed=5)
        s.reset(hp, idx)
        if rep == 1:
            s.capture(iters=2)
            assert s.graph_mode == "graph branches"
        if rep == 2:
            s.capture(fused=True, iters=2)
            assert s.graph_mode == "one grid"
        for _ in range(4):
            s.step()
        s.flush()
        torch.cuda.synchronize()
        outs.append((s.energy.clone(), s.hand_pose.clone(), s.contact_idx.clone(), s.grad.clone(), s.terms.clone()))
        assert int(s.n_iter.item()) >= 1
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][3]).all()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b), "hipGraph replay differs from the eager unfused launches"
    pose, cidx = outs[0][1], outs[0][2]  # accepted state after four iterations: a mix of moved and unmoved rows

    # (b) E_fc and its contact-point gradient of the WHOLE batch against the oracle's metric (fp64) run on the same
    # B x n contact points / object normals: identical batch composition, so qpth's batch-global stop rule sees the
    # same rows (1e-4 median, p99 / max <= 2x the oracle's own fp32 noise on the same rows; gradient 2e-2 norm-wise, n_iter equal)
    st = gq.stepper.GraspStepper(hand, ms, surf, be, n, fc_cfg=fc_cfg,
                                 weights={"E_dis": 0.0, "E_fc": 1.0, "E_pen": 0.0, "E_spen": 0.0, "E_joints": 0.0})
    t_fc, _, _ = st.evaluate(pose, cidx)
    torch.cuda.synchronize()
    po = st.cpts.cpu().double().requires_grad_()
    eo, _ = ospan.e_fc(po, st.obj_normal.cpu().double(), st.cog.cpu().double(), k=k, box_form=True)
    n_iter_o = oqp.LAST["n_iter"]
    eo.sum().backward()
    rel = _rel(t_fc["E_fc"].cpu().numpy(), eo.detach().numpy())
    assert_tail_within_fp32_noise(rel, oracle_fc_fp32_noise(ospan, st.cpts.cpu(), st.obj_normal.cpu(), st.cog.cpu(), k, eo.detach()),
                                  f"E_fc {hand_name} {B} rows")
    assert int(st.n_iter.item()) == n_iter_o, (int(st.n_iter.item()), n_iter_o)
    gfc = np.linalg.norm(st.g_cpts.cpu().numpy() - po.grad.numpy()) / np.linalg.norm(po.grad.numpy())
    assert gfc < 2e-2, gfc

    # (c) a sample of rows, one per third of the batch, against the fp64 oracle: the four other terms and the gradient
    # of their weighted sum (E_fc's stop rule depends on the batch: checked in (b))
    w0 = {"E_dis": 100.0, "E_fc": 0.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    s3 = gq.stepper.GraspStepper(hand, ms, surf, be, n, fc_cfg=fc_cfg, weights=w0)
    t3, _, g3 = s3.evaluate(pose, cidx)
    torch.cuda.synchronize()
    assert (t3["E_pen"] > 0).sum() > 0, "scene must contain penetrating rows"
    pen_rows = torch.nonzero(t3["E_pen"] > 1e-4).flatten().tolist()
    rows = sorted({0, B // 2 + 1, B - 1, pen_rows[0], pen_rows[-1]}) if not lean else [pen_rows[0]]
    oh = omodels.OracleHand(spec, torch.float64)
    for r in rows:
        o = r // be
        oo = omodels.OracleObject([fvs[o]], [sps[o]], 1, torch.float64)
        hpo = pose[r : r + 1].cpu().double().requires_grad_()
        oh.set_parameters(hpo, cidx[r : r + 1].cpu())
        lo = ref_cpu.calculate_energy(oh, oo, box_form=True, k=k)
        for kk in ("E_dis", "E_pen", "E_spen", "E_joints"):
            np.testing.assert_allclose(t3[kk][r].item(), lo[kk].item(), rtol=3e-4, atol=3e-6, err_msg=f"{kk} row {r}")
        sum(w0[kk] * lo[kk] for kk in w0 if w0[kk] != 0.0).sum().backward()
        go = oh.hand_pose.grad.numpy()[0]
        gerr = np.linalg.norm(g3[r].cpu().numpy() - go) / np.linalg.norm(go)
        assert gerr < 2e-3, (r, gerr)

    # (d) the stepper's fused evaluation == the autograd route built from the C-ABI building blocks (2048-row case only:
    # the class surface materialises (B,P,3) tensors like the reference does)
    if B <= 2048:
        from graspqp_amd.core.energy import calculate_energy
        from graspqp_amd.core.hand_model import HandModel
        from graspqp_amd.core.object_model import ObjectModel
        from graspqp_amd.metrics import GraspSpanMetricFactory as GF

        W = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
        s4 = gq.stepper.GraspStepper(hand, ms, surf, be, n, fc_cfg=fc_cfg)
        t4, tot4, g4 = s4.evaluate(pose, cidx)
        hm = HandModel(spec, "cuda")
        om = ObjectModel(batch_size_each=be, num_samples=2500)
        om.initialize_from_meshes(fvs, surface_points_list=sps)
        hm.set_parameters(pose.clone().requires_grad_(), cidx)
        fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": k})
        losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=list(W), svd_gain=0.1)
        tot = sum(W[kk] * v for kk, v in losses.items())
        tot.sum().backward()
        for kk in W:
            np.testing.assert_allclose(t4[kk].cpu().numpy(), losses[kk].detach().cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=kk)
        assert (hm.hand_pose.grad - g4).norm() <= 2e-3 * g4.norm()


# ---------------------------------------------------------------------------------------------------------------
# qpth's batch-global stop rule
# ---------------------------------------------------------------------------------------------------------------
def _gpu_stop(gq, resid, mu, eps, lim):
    B, T = resid.shape
    r, m = resid.float().contiguous().cuda(), mu.float().contiguous().cuda()
    scratch = torch.empty(B, device="cuda")
    ks = torch.full((2,), -7, dtype=torch.int32, device="cuda")
    nit = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    gq.C.call("gq_boxqp_stop_rule", gq.C.f32(r), gq.C.f32(m), ctypes.c_int64(B), T, float(eps), int(lim),
              gq.C.f32(scratch), gq.C.i32(ks), gq.C.i32(nit), gq.C.stream_ptr())
    torch.cuda.synchronize()
    return ks.tolist(), int(nit.item())


@pytest.mark.parametrize("B", [64, 700, 1024, 1025, 3000, 5000])
@pytest.mark.parametrize("T", [12, 20])  # > 16 iterations: the generic kernel
def test_stop_rule_kernels_on_tables(gq, B, T):
    """Every stop condition with its deciding row in the LAST 1024-row tile (a kernel that looks at the first tile, or at
    any periodic subset, stops at a different iteration), NaN residuals, and random tables -- bit-exact against the
    oracle's restatement of the rule evaluated on the same fp32 tables."""
    g = torch.Generator().manual_seed(B * 31 + T)
    eps = 5e-2
    last = B - 1 - int(torch.randint(0, min(300, B - 1024 * ((B - 1) // 1024)), (1,), generator=g))  # a row of the last tile
    cases = {}
    # 1) eps rule: every row is below eps from iteration 3 on, the deciding row only from iteration 8 on
    r = torch.rand(B, T, generator=g) * 0.01 + 1e-3
    r[:, :3] += 1.0
    r[last, :8] = 0.9 - 0.01 * torch.arange(8.0)
    cases["eps"] = (r, torch.ones(B, T), eps, 3, 8)
    # 2) not-improved rule: all rows stop improving after iteration 2 except the deciding row, which improves until 6
    r = torch.ones(B, T)
    r[:, 1] = 0.9
    r[:, 2] = 0.8
    r[:, 3:] = 0.8 + 0.05 * torch.rand(B, T - 3, generator=g)
    r[last, :7] = 2.0 - 0.1 * torch.arange(7.0)
    r[last, 7:] = 1.5
    cases["not_improved"] = (r, torch.ones(B, T), 1e-6, 3, 9)
    # 3) mu rule: every mu explodes at iteration 4 except the deciding row's, which does so at iteration 7 (residuals
    # keep improving, so the other rules never fire)
    r = (2.0 - 0.05 * torch.arange(float(T)))[None].repeat(B, 1) + torch.rand(B, 1, generator=g)
    m = torch.ones(B, T)
    m[:, 4:] = 3e33
    m[last, 4:7] = 5.0
    cases["mu"] = (r, m, 1e-6, 3, 7)
    # 4) NaN never improves: from iteration 2 on every row but the deciding one produces NaN residuals; the deciding row
    # improves until iteration 6
    r = torch.ones(B, T)
    r[:, 1] = 0.5
    r[:, 2:] = float("nan")
    r[last, 2:7] = torch.tensor([0.4, 0.3, 0.2, 0.1, 0.05])
    r[last, 7:] = 0.3
    cases["nan"] = (r, torch.ones(B, T), 1e-6, 3, 9)
    for name, (r, m, e, lim, want) in cases.items():
        ks_o, _ = oqp.stop_rule(r.float(), m.float(), e, lim)
        assert ks_o == want, (name, ks_o, want)  # the scenario does what it says
        if B > 1024:  # ... and would be decided differently without the last tile
            ks_first, _ = oqp.stop_rule(r[:1024].float(), m[:1024].float(), e, lim)
            assert ks_first != ks_o, name
        ks, nit = _gpu_stop(gq, r, m, e, lim)
        assert ks == [ks_o, ks_o + 1] and nit == ks_o + 1, (name, B, T, ks, ks_o)
    # 5) random tables with NaNs sprinkled in
    for trial in range(4):
        r = torch.rand(B, T, generator=g) * torch.logspace(0, -3, T)[None]
        r[torch.rand(B, T, generator=g) < 0.02] = float("nan")
        m = torch.rand(B, T, generator=g)
        e = [5e-2, 1e-2, 1e-3, 2e-1][trial]
        ks_o, _ = oqp.stop_rule(r.float(), m.float(), e, 3)
        ks, nit = _gpu_stop(gq, r, m, e, 3)
        assert ks[0] == ks_o and nit == ks_o + 1, ("random", trial, B, T, ks, ks_o)


@pytest.mark.parametrize("n,k,B", [(12, 4, 3072), (4, 4, 5000)])
def test_qp_heterogeneous_batch_deciding_row_in_last_tile(gq, n, k, B):
    """A heterogeneous batch (rows converge after 4 .. 10 PDIPM iterations) arranged so that the slowest rows -- the ones
    that decide the batch-global stop iteration -- sit in the last 1024-row tile: n_iter and x against the oracle's
    PDIPM (fp64) on the same batch."""
    pts, nrm, cog = hetero_contacts(B + B // 2, n, 11)
    F = ospan.grasp_matrix(pts, nrm, cog, 0.2, k)
    nz = F.shape[-1]
    Q = F.transpose(1, 2) @ F + 1e-4 * torch.eye(nz, dtype=F.dtype)
    Bc = F.shape[0]
    lo, hi = torch.ones(Bc, nz, dtype=F.dtype), 21 * torch.ones(Bc, nz, dtype=F.dtype)
    full = []
    oqp.pdipm_forward_box(Q, torch.zeros(Bc, nz, dtype=F.dtype), lo, hi, eps=-1.0, notImprovedLim=99, history=full)
    resid = torch.stack([h["resids"] for h in full], 1)
    runbest = torch.cummin(torch.nan_to_num(resid, nan=float("inf")), 1)[0]  # a NaN iterate never becomes best
    eps = 5e-2
    # target stop iteration ks: every kept row is clearly below eps at ks; the "deciders" are clearly above eps at ks - 1,
    # all other rows clearly below (25 % margins: the comparison is about the rule, not about fp32-vs-fp64 round-off of
    # a residual that passes eps by a hair)
    for ks in range(11, 3, -1):
        ok = (torch.isfinite(resid[:, : ks + 1]).all(1) & (runbest[:, ks] < 0.75 * eps)
              & ((runbest[:, ks - 1] < 0.75 * eps) | (runbest[:, ks - 1] > 1.25 * eps)))
        dec = ok & (runbest[:, ks - 1] > 1.25 * eps)
        if ok.sum() >= B and dec.sum() >= 1:
            break
    else:
        raise AssertionError("no feasible stop iteration in the synthetic batch")
    easy, decid = torch.nonzero(ok & ~dec).flatten(), torch.nonzero(dec).flatten()[:64]
    order = torch.cat([easy[: B - decid.numel()], decid])  # the deciding rows close the batch: last 1024-row tile
    assert order.numel() == B
    Fb = F[order].contiguous()
    b0 = torch.zeros(B, 6, dtype=F.dtype)
    val_o, x_o = oqp.lsq_box_qp(Fb, b0, 1.0, 21.0, box_form=True)
    n_iter_o = oqp.LAST["n_iter"]
    assert n_iter_o == ks + 1
    oqp.lsq_box_qp(Fb[:1024], b0[:1024], 1.0, 21.0, box_form=True)
    assert oqp.LAST["n_iter"] < n_iter_o, "the first tile alone must stop earlier (or the test cannot fail)"
    Fg = Fb.float().cuda().requires_grad_()
    x, nit = gq.ops.lsq_box_qp(Fg, None, 1.0, 21.0, return_n_iter=True)
    torch.cuda.synchronize()
    assert int(nit.item()) == n_iter_o, (int(nit.item()), n_iter_o)
    val = 0.5 * ((Fg @ x.unsqueeze(-1)).squeeze(-1) ** 2).sum(-1)
    rel = _rel(2 * (val.detach().cpu().numpy() + 0.01), 2 * (val_o.numpy() + 0.01))
    val_32, _ = oqp.lsq_box_qp(Fb.float(), b0.float(), 1.0, 21.0, box_form=True)  # the oracle's own fp32 noise on these rows
    assert_tail_within_fp32_noise(rel, _rel(2 * (val_32.double().numpy() + 0.01), 2 * (val_o.numpy() + 0.01)), f"QP value B={B}")
    # x itself: Q = F'F + 1e-4 I has rank 6 + ridge, so rows with bunched contacts have directions along which x is
    # determined by the ridge alone (the value is flat there): nearly all entries agree tightly, a few outliers may not
    dx = np.abs(x.detach().cpu().numpy() - x_o.numpy())
    assert np.quantile(dx, 0.999) < 5e-2 and dx.max() < 2.0, (np.quantile(dx, 0.999), dx.max())
    assert (x.min() >= 1.0 - 1e-4) and (x.max() <= 21.0 + 1e-3)


# ---------------------------------------------------------------------------------------------------------------
# hand-penetration query: LDS capacity overflow on purpose
# ---------------------------------------------------------------------------------------------------------------
def test_hand_penetration_item_list_overflow(gq):
    """A tiny dense object placed inside an Allegro fingertip: each of the 256 surface points of a block lies deep inside
    the link, where a voxel's candidate list holds ~30 faces -> ~7 000 items against GQ_PG_ICAP = 4096, so many
    entries of every block are ranked inline and part of the reserved item slots stay unused (the path whose
    uninitialised-slot bug was fixed in round 1).  The result must equal the exact query (penetration_only = 0)."""
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel

    spec = get_hand_spec("allegro")
    link = 3  # fingertip mesh, 342 faces
    P, be = 1024, 3
    fv = meshes.icosphere(3, 0.004)
    sp = meshes.surface_points(fv, P, oversample=4, seed=1)
    th = torch.tensor(spec.default_state, dtype=torch.float64)[None].repeat(be, 1)
    th[1] += 0.2
    th[2] -= 0.1
    oh = omodels.OracleHand(spec, torch.float64)
    hp = torch.cat([torch.zeros(be, 3, dtype=torch.float64), torch.tensor([[1.0, 0, 0, 0, 1, 0]], dtype=torch.float64).repeat(be, 1), th], 1)
    oh.set_parameters(hp, torch.zeros(be, 4, dtype=torch.long))
    lf = spec.link_faces(link).reshape(-1, 3)
    c = torch.tensor(0.5 * (lf.min(0) + lf.max(0)), dtype=torch.float64)
    T = oh.current_status[:, link]
    hp[:, :3] = -(T[:, :3, :3] @ c + T[:, :3, 3])  # global rotation = identity: the link centre lands on the object
    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=be, num_samples=P)
    om.initialize_from_meshes([fv], surface_points_list=[sp])
    hm.set_parameters(hp.float().cuda(), torch.zeros(be, 4, dtype=torch.long).cuda())
    dis0 = hm.cal_distance(om.surface_points_each, penetration_only=0)
    cnt = torch.zeros(12, dtype=torch.int64, device="cuda")
    gq.C.call("gq_debug_set_pen_counters", ctypes.c_void_p(cnt.data_ptr()))
    try:
        dis1 = hm.cal_distance(om.surface_points_each, penetration_only=1)
        torch.cuda.synchronize()
    finally:
        gq.C.call("gq_debug_set_pen_counters", None)
    entries, rankings, inline, blocks = (int(v) for v in cnt[4:8])
    assert blocks == be * (P // 256)
    assert rankings > 4096 * blocks and inline > 0.2 * entries, (entries, rankings, inline, blocks)
    pos = dis0 > 1e-6
    assert pos.float().mean() > 0.9, "the object must sit inside the link"
    torch.testing.assert_close(dis1[pos], dis0[pos], rtol=2e-4, atol=3e-6)
    assert (dis1[dis0 <= -1e-6] <= 0).all()
    # against the fp64 oracle as well
    oh.set_parameters(hp, torch.zeros(be, 4, dtype=torch.long))
    oo = omodels.OracleObject([fv], [sp], be, torch.float64)
    do = oh.cal_distance(oo.surface_points_tensor).numpy()
    big = np.abs(dis1.cpu().numpy() - do)[do > 1e-6] > 3e-6
    assert big.mean() < 2e-3, big.mean()


@pytest.mark.parametrize("scene", ["bench", "deep"])
def test_penetration_query_two_points_per_thread_equals_one(gq, scene):
    """gq_pen_grid_body with two surface points per thread (the default of the fused stage-A role) against one point per
    thread (the default of the stand-alone launch): dis bit for bit -- the list phases are order-independent and overflow
    is ranked inline with the same arithmetic.  2500 points: the last block of a row is ragged in both forms."""
    from bench import make_initial_state
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel

    spec = get_hand_spec("allegro")
    n_obj, be, P = 2, 24, 2500
    fvs = [meshes.superquadric(o) for o in range(n_obj)]
    sps = [meshes.surface_points(f, P, oversample=4, seed=42) for f in fvs]
    hp, idx = zip(*[make_initial_state(spec, f, be, 12, 1000 + o) for o, f in enumerate(fvs)])
    hp, idx = torch.cat(hp), torch.cat(idx)
    if scene == "deep":
        hp[:, :3] *= 0.3  # the hand well inside the object: many entries per block, list overflow
    else:
        hp[::5, :3] *= 0.5
    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=be, num_samples=P)
    om.initialize_from_meshes(fvs, surface_points_list=sps)
    hm.set_parameters(hp.cuda(), idx.cuda())
    outs = []
    try:
        for ppt in (1, 2):
            gq.C.call("gq_debug_set_pen_ppt", ppt)
            outs.append(hm.cal_distance(om.surface_points_each, penetration_only=1).clone())
            torch.cuda.synchronize()
    finally:
        gq.C.call("gq_debug_set_pen_ppt", 0)
    assert (outs[0] > 0).sum() > 0, "scene must contain penetrating points"
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("scene", ["bench", "deep", "fingertip"])
@pytest.mark.parametrize("cells", [8, 4, 16])
def test_link_driven_query_equals_point_driven(gq, scene, cells):
    """gq_hand_pen_forward_cells (a row's block walks the grid cells each link box can touch) against the point-driven
    query gq_hand_pen_forward(penetration_only = 1) it replaces in the fused launches and against the exact query
    (penetration_only = 0): bench-like poses, hands deep inside the object (> GQ_PC_ECAP overlaps per row: the dense
    fall-back) and the fingertip scene (item-list overflow: inline ranking)."""
    from bench import make_initial_state

    spec = get_hand_spec("allegro")
    C, f32, i32 = gq.C.call, gq.C.f32, gq.C.i32
    if scene == "fingertip":
        n_obj, be, P = 1, 3, 1024
        fvs = [meshes.icosphere(3, 0.004)]
        sps = [meshes.surface_points(fvs[0], P, oversample=4, seed=1)]
        th = torch.tensor(spec.default_state, dtype=torch.float64)[None].repeat(be, 1)
        hp = torch.cat([torch.zeros(be, 3, dtype=torch.float64), torch.tensor([[1.0, 0, 0, 0, 1, 0]], dtype=torch.float64).repeat(be, 1), th], 1)
        oh = omodels.OracleHand(spec, torch.float64)
        oh.set_parameters(hp, torch.zeros(be, 4, dtype=torch.long))
        lf = spec.link_faces(3).reshape(-1, 3)
        c = torch.tensor(0.5 * (lf.min(0) + lf.max(0)), dtype=torch.float64)
        T = oh.current_status[:, 3]
        hp[:, :3] = -(T[:, :3, :3] @ c + T[:, :3, 3])
        hp = hp.float()
    else:
        n_obj, be, P = 2, 48, 2500
        fvs = [meshes.superquadric(o) for o in range(n_obj)]
        sps = [meshes.surface_points(f, P, oversample=4, seed=42) for f in fvs]
        hp = torch.cat([make_initial_state(spec, f, be, 12, 1000 + o)[0] for o, f in enumerate(fvs)])
        if scene == "deep":
            hp[:, :3] *= 0.15  # hands inside the objects: thousands of (point, link) overlaps per row
        else:
            hp[::5, :3] *= 0.5
    B = n_obj * be
    hand = gq.ops.HandHandle(spec)
    surf = torch.tensor(np.stack(sps)).cuda().contiguous()
    grid = gq.ops.PointGrid(surf, cells)
    patch = torch.empty(n_obj, (P + 255) // 256, 4, device="cuda")
    C("gq_surface_patches", f32(surf), ctypes.c_int64(n_obj), ctypes.c_int64(P), f32(patch), gq.C.stream_ptr())
    ctr = patch[..., :3].unsqueeze(2)  # every point of a slice inside its sphere
    for o in range(n_obj):
        for b in range(patch.shape[1]):
            pts = surf[o, b * 256 : (b + 1) * 256]
            assert float((pts - patch[o, b, :3]).norm(dim=-1).max()) <= float(patch[o, b, 3])
    pose = hp.cuda().contiguous()
    idx = torch.zeros(B, 1, dtype=torch.long, device="cuda")
    Rg, LT, *_ = gq.ops.fk_contacts(pose, idx, hand)
    Rg, LT = Rg.reshape(B, 9).contiguous(), LT.reshape(B, hand.L, 12).contiguous()

    def run(mode):
        dis = torch.empty(B, P, device="cuda")
        link = torch.zeros(B, P, dtype=torch.int32, device="cuda")
        gvec = torch.zeros(B, P, 3, device="cuda")
        if mode == "cells":
            C("gq_hand_pen_forward_cells", hand.links.handle, grid.handle, f32(surf), n_obj, P, be, f32(pose), pose.shape[1],
              f32(Rg), f32(LT), f32(dis), i32(link), f32(gvec), None, None, gq.C.stream_ptr())
        else:
            C("gq_hand_pen_forward", hand.links.handle, f32(surf), n_obj, P, be, f32(pose), pose.shape[1], f32(Rg), f32(LT),
              1 if mode == "patch" else int(mode), f32(dis), i32(link), f32(gvec), None, 0, None, None, f32(patch) if mode == "patch" else None,
              gq.C.stream_ptr())
        torch.cuda.synchronize()
        return dis, link, gvec

    cnt = torch.zeros(12, dtype=torch.int64, device="cuda")
    C("gq_debug_set_pen_counters", ctypes.c_void_p(cnt.data_ptr()))
    try:
        d_c, l_c, g_c = run("cells")
    finally:
        C("gq_debug_set_pen_counters", None)
    d_p, l_p, g_p = run(1)
    d_0, l_0, g_0 = run(0)
    # the point-driven query with the block-level link pre-cull (bounding spheres of the 256-point slices): same bits
    d_s, l_s, g_s = run("patch")
    assert torch.equal(d_s, d_p) and torch.equal(l_s, l_p) and torch.equal(g_s, g_p)
    pos = d_p > 0
    assert pos.sum() > 0, "scene must contain penetrating points"
    assert torch.equal(d_c > 0, pos), "the two queries must find the same penetrating points"
    # same arithmetic in two kernels, but instruction contraction may differ in the last bit of a ranking distance, and
    # near-tied faces then swap (<= 1e-5 m at points equidistant to two faces, as between the modes in test_gpu_parity)
    far = (d_c[pos] - d_p[pos]).abs() > 3e-6
    assert far.float().mean() < 2e-3 and float((d_c[pos] - d_p[pos]).abs().max()) < 1e-4
    same = ~far
    assert (l_c[pos][same] == l_p[pos][same]).float().mean() > 0.999
    # a point equidistant to two faces may take its gradient direction from either of them
    g_far = (g_c[pos][same] - g_p[pos][same]).abs().amax(-1) > 2e-3
    assert g_far.float().mean() < 1e-2
    assert bool((d_c[~pos] == -1e30).all())
    big = (d_c[pos] - d_0[pos]).abs() > 3e-6  # against the exact query: near-tied faces may swap (tests/test_gpu_parity.py)
    assert big.float().mean() < 2e-3
    entries, rankings, inline, blocks, pairs = int(cnt[4]), int(cnt[5]), int(cnt[6]), int(cnt[7]), int(cnt[0])
    assert blocks == B and entries >= int(pos.sum())
    if scene == "deep":
        assert inline > 0, "the deep scene must overflow the entry list of some row (dense fall-back)"
    if scene == "fingertip":
        assert inline > 0, "the fingertip scene must overflow the item list (inline ranking)"
    if scene == "bench":
        assert pairs < 0.05 * B * P * hand.L, "the link-driven walk must visit far fewer pairs than points x links"
