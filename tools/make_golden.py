#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference's own Python files in this container.

Only the build container has /root/reference; the GPU box does not.  This script loads the parts
of the reference that import here *from where they lie* -- nothing is copied -- and
stores plain input/output tensors:

  kat_solver.npz   the A, b of reference tests/metrics/test_solver.py and the ScipyLsqSolver answer
  span_*.npz       reference metrics/ops/span.py + registry.py (GRASPQP_SCIPY): inputs -> F, svd, E_fc, x
  energy_*.npz     reference core/energy.py::calculate_energy composed over the oracle models
  mala_*.npz       reference core/optimizer.py::MalaStar driven through the fit.py loop order for a
                   few iterations on the oracle models, with every torch.rand/randint draw recorded
  init_*.npz       reference core/initializations.py::initialize_convex_hull with oracle stand-ins for trimesh /
                   pytorch3d / transforms3d: look_at, pose assembly, truncated-normal joints, env_mask hand-over
  annealing_*.npz  reference core/optimizer.py::AnnealingDexGraspNet + metrics/ops/dexgrasp.py in fit.py order
  alt_metrics.npz  reference metrics/ops/dexgrasp.py and tdg.py: energies + contact-point gradients
  mala_ext_*.npz   the same with (R) the step counter started at 149 and a re-initialisation iteration
                   (reset_envs + accept_step(reset_mask)), (C) clip_grad=True and NaN / inf gradient entries

qpth / TorchSDF / pytorch_kinematics / roma are not importable, so E_fc inside energy_/mala_
fixtures uses the oracle's PDIPM restatement (that part stays "parity unpinned"); what these
fixtures pin is everything *around* it: loop order, optimizer state handling, energy formulas.
"""
import importlib
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
REF = os.environ.get("GRASPQP_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "graspqp", "src", "graspqp")
OUT = os.path.join(ROOT, "tests", "golden")
OUT = os.environ.get("GRASPQP_GOLDEN_OUT", OUT)  # tests regenerate into a temporary directory
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np
import torch

# graspqp_amd data helpers without importing the HIP-backed package __init__
_pkg = types.ModuleType("graspqp_amd")
_pkg.__path__ = [os.path.join(ROOT, "graspqp_amd")]
sys.modules.setdefault("graspqp_amd", _pkg)
from graspqp_amd.hands import get_hand_spec  # noqa: E402
from graspqp_amd.utils import meshes  # noqa: E402

import ref_cpu  # noqa: E402
from ref_cpu import models as omodels  # noqa: E402
from ref_cpu import span as ospan  # noqa: E402


def load_ref(mod_name, rel_path):
    spec = importlib.util.spec_from_file_location(mod_name, os.path.join(SRC, rel_path))
    m = importlib.util.module_from_spec(spec)
    sys.modules[mod_name] = m
    spec.loader.exec_module(m)
    return m


def ref_metrics():
    """Import reference metrics/ops/{span,registry}.py with the reference's ScipyLsqSolver standing in
    for the qpth-backed class that span.py imports unconditionally."""
    for name, path in (
        ("graspqp", SRC),
        ("graspqp.metrics", os.path.join(SRC, "metrics")),
        ("graspqp.metrics.solver", os.path.join(SRC, "metrics", "solver")),
        ("graspqp.metrics.ops", os.path.join(SRC, "metrics", "ops")),
    ):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    scipy_solver = load_ref("graspqp.metrics.solver.scipy_solver", "metrics/solver/scipy_solver.py")
    stub = types.ModuleType("graspqp.metrics.solver.qp_solver")
    stub.SQPLsqSolver = scipy_solver.ScipyLsqSolver
    sys.modules["graspqp.metrics.solver.qp_solver"] = stub
    registry = importlib.import_module("graspqp.metrics.ops.registry")
    return scipy_solver, registry


def to_np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def gen_kat(scipy_solver):
    A = torch.Tensor(
        [
            [-0.0, 0.25819889, 0.25819889, -0.25819889, -0.25819889, -0.25819889],
            [0.31622777, -0.18257419, -0.18257419, -0.18257419, -0.18257419, -0.18257419],
            [0.02810913, -0.09128826, 0.10345754, 0.13997471, -0.05477109, -0.10345754],
        ]
    ).unsqueeze(0)
    b = torch.Tensor([0.25819889, 0.18257419, 0.00608464]).unsqueeze(0)
    solver = scipy_solver.ScipyLsqSolver.from_mat(A, b)
    val, x = solver(A, b, min_bound=-10.0, max_bound=1e3, init=0.1, return_solution=True)
    np.savez(os.path.join(OUT, "kat_solver.npz"), A=A.numpy(), b=b.numpy(), value=val.numpy(), x=x.numpy(),
             min_bound=-10.0, max_bound=1e3)
    print("kat value", val)


def gen_span(registry):
    MT = registry.GraspSpanMetricFactory.MetricType
    for n, k, B, seed in ((4, 4, 16, 0), (12, 4, 32, 1), (16, 4, 16, 2), (12, 8, 16, 3)):
        torch.manual_seed(seed)
        # contacts roughly on a 5 cm object, normals roughly outward + noise (some near the b1 flip zone)
        # the reference metric is float32-only (b1 is built as float32, span.py:271-273)
        d = torch.nn.functional.normalize(torch.randn(B, n, 3), dim=-1)
        pts = d * (0.05 + 0.01 * torch.randn(B, n, 1))
        nrm = torch.nn.functional.normalize(d + 0.3 * torch.randn(B, n, 3), dim=-1)
        nrm[0, 0] = torch.tensor([1.0, 1.0, 1.0]) / 3**0.5  # exercises dot > 0.9 branch
        cog = 0.005 * torch.randn(B, 3)
        fn = registry.GraspSpanMetricFactory.create(
            MT.GRASPQP_SCIPY, solver_kwargs={"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": k}
        )
        pts_r = pts.clone().requires_grad_()
        e, x = fn(contact_pts=pts_r, contact_normals=nrm, sdf=None, cog=cog, with_solution=True, svd_gain=0.1)
        F = fn.metric._cache["F"]
        svd = (torch.linalg.svdvals(F)).prod(-1) ** (1 / 6)
        np.savez(
            os.path.join(OUT, f"span_n{n}_k{k}.npz"),
            **to_np(dict(contact_pts=pts, contact_normals=nrm, cog=cog, F=F, svd=svd, e_fc=e, x_sum=x,
                         friction=0.2, max_limit=20.0, n_cone_vecs=k, svd_gain=0.1)),
        )
        print(f"span n={n} k={k}: F{tuple(F.shape)} e_fc[0:3]={e[:3].tolist()}")


class _FcAdapter:
    """energy_fnc with the reference calling convention, backed by the oracle PDIPM."""

    def __init__(self, **kw):
        self.kw = kw

    def __call__(self, contact_pts, contact_normals, sdf, cog, with_solution=True, svd_gain=0.1):
        return ospan.e_fc(contact_pts, contact_normals, cog, svd_gain=svd_gain, **self.kw)


def make_scene(hand_name, n_obj, batch_each, n_contact, dtype, seed, mesh="sphere", n_surface=2500):
    spec = get_hand_spec(hand_name)
    hand = omodels.OracleHand(spec, dtype=dtype)
    fvs, sps = [], []
    for i in range(n_obj):
        fv = meshes.icosphere(3, 0.05) if mesh == "sphere" else meshes.superquadric(seed + i, 48, 24)
        fvs.append(fv)
        sps.append(meshes.surface_points(fv, n_surface, oversample=8, seed=42))
    obj = omodels.OracleObject(fvs, sps, batch_each, dtype=dtype)
    B = n_obj * batch_each
    g = torch.Generator().manual_seed(seed)
    # hand ~12-16 cm from the object centre, palm roughly facing it, joints = default + jitter
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g, dtype=torch.float64), dim=-1)
    t = dirs * (0.10 + 0.04 * torch.rand(B, 1, generator=g, dtype=torch.float64))
    six = torch.randn(B, 6, generator=g, dtype=torch.float64)
    th = torch.as_tensor(spec.default_state, dtype=torch.float64)[None] + 0.15 * torch.randn(
        B, spec.n_dofs, generator=g, dtype=torch.float64
    )
    hp = torch.cat([t, six, th], dim=1).to(dtype)
    idx = torch.randint(spec.n_contact_candidates, (B, n_contact), generator=g)
    return spec, hand, obj, hp, idx, fvs, sps


def gen_energy(ref_energy):
    for tag, hand_name, n_obj, be, n, mesh, seed in (
        ("allegro_sphere_b4_n4", "allegro", 1, 4, 4, "sphere", 11),
        ("allegro_sq_b6_n12", "allegro", 2, 3, 12, "sq", 12),
    ):
        spec, hand, obj, hp, idx, fvs, sps = make_scene(hand_name, n_obj, be, n, torch.float64, seed, mesh, n_surface=600)
        # push one row into the object so that E_pen / E_dis-inside branches are exercised
        hp[0, :3] *= 0.35
        hp = hp.clone().requires_grad_()
        hand.set_parameters(hp, idx)
        losses = ref_energy.calculate_energy(
            hand, obj, energy_fnc=_FcAdapter(mu=0.2, k=4, max_limit=20.0),
            energy_names=["E_dis", "E_fc", "E_pen", "E_spen", "E_joints"], method="gendexgrasp", svd_gain=0.1,
        )
        w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
        tot = sum(w[k] * v for k, v in losses.items())
        tot.sum().backward()
        out = dict(hand_pose=hp, contact_idx=idx, total=tot, grad=hand.hand_pose.grad,
                   batch_size_each=be, n_obj=n_obj)
        # the optional terms of energy.py:68-78 (E_prior, E_wall) from the same reference file, on seeded surface samples
        rng = np.random.default_rng(5)
        sl = np.sort(rng.integers(0, spec.n_links, 64)).astype(np.int32)
        spts = np.stack([spec.link_faces(int(l))[rng.integers(0, len(spec.link_faces(int(l))))].mean(0) for l in sl]).astype(np.float64)
        hand.surface_points, hand.surface_link = spts, sl
        hp2 = hp.detach().clone()
        hp2[:, 2] -= 0.08  # part of the hand below the table plane z = 0
        hp2.requires_grad_()
        hand.set_parameters(hp2, idx)
        opt_l = ref_energy.calculate_energy(
            hand, obj, energy_fnc=_FcAdapter(mu=0.2, k=4, max_limit=20.0),
            energy_names=["E_dis", "E_fc", "E_pen", "E_spen", "E_joints", "E_prior", "E_wall"], method="gendexgrasp", svd_gain=0.1)
        (2.0 * opt_l["E_prior"] + 3.0 * opt_l["E_wall"]).sum().backward()
        out.update(opt_hand_pose=hp2, opt_surface_points=spts, opt_surface_link=sl, opt_E_prior=opt_l["E_prior"],
                   opt_E_wall=opt_l["E_wall"], opt_grad=hand.hand_pose.grad)
        hand.set_parameters(hp, idx)
        for k, v in losses.items():
            out[k] = v
        for i in range(n_obj):
            out[f"obj{i}_face_verts"] = fvs[i]
            out[f"obj{i}_surface_points"] = sps[i]
        np.savez_compressed(os.path.join(OUT, f"energy_{tag}.npz"), **to_np(out))
        print(f"energy {tag}:", {k: v.detach().numpy().round(5).tolist() for k, v in losses.items()})


class _Recorder:
    """Records every torch.rand / torch.randint result while active."""

    def __init__(self):
        self.log = []

    def __enter__(self):
        self._rand, self._randint = torch.rand, torch.randint

        def rand(*a, **k):
            r = self._rand(*a, **k)
            self.log.append(("rand", r.clone()))
            return r

        def randint(*a, **k):
            r = self._randint(*a, **k)
            self.log.append(("randint", r.clone()))
            return r

        torch.rand, torch.randint = rand, randint
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randint = self._rand, self._randint


def gen_mala(ref_energy, ref_opt):
    tag, hand_name, n_obj, be, n, seed, n_steps = "allegro_sphere_b8_n4", "allegro", 2, 4, 4, 21, 5
    # fp64 models under the reference's (fp32-typed) optimizer: the trajectory is then free of the fp32 noise of a
    # CPU PDIPM and serves as the high-accuracy reference for both the fp64 oracle replay and the fp32 GPU path
    dtype = torch.float64
    spec, hand, obj, hp, idx, fvs, sps = make_scene(hand_name, n_obj, be, n, dtype, seed, "sphere", n_surface=400)
    B = n_obj * be
    hp0 = hp.clone().requires_grad_()
    hand.set_parameters(hp0, idx)
    fc = _FcAdapter(mu=0.2, k=4, max_limit=20.0)
    names = ["E_dis", "E_fc", "E_pen", "E_spen", "E_joints"]
    w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    opt = ref_opt.MalaStar(hand, switch_possibility=0.4, starting_temperature=18, temperature_decay=0.95,
                           annealing_period=30, step_size=0.005, stepsize_period=50, mu=0.98, device="cpu",
                           batch_size=be, clip_grad=False)

    def total(losses):
        return sum(w[k] * v for k, v in losses.items())

    # fit.py:381-396
    losses = ref_energy.calculate_energy(hand, obj, energy_fnc=fc, energy_names=names, method="gendexgrasp", svd_gain=0.1)
    energy = total(losses)
    energy.sum().backward()
    out = dict(hand_pose0=hand.hand_pose.detach().clone(), contact_idx0=idx.clone(), energy0=energy.detach().clone(),
               grad0=hand.hand_pose.grad.detach().clone(), batch_size_each=be, n_obj=n_obj, n_steps=n_steps)
    opt.zero_grad()  # fit.py:396 -- zeroes the initial gradient
    torch.manual_seed(seed)
    for step in range(1, n_steps + 1):
        with _Recorder() as rec:
            s = opt.try_step()  # fit.py:400
        u_switch = rec.log[0][1]
        vals = rec.log[1][1]
        mask = u_switch < 0.4
        new_idx = torch.zeros(B, n, dtype=torch.long)
        new_idx[mask] = vals
        eb = energy.view(-1, be)  # fit.py:403-406
        z = ((eb - eb.mean(-1).unsqueeze(-1)) / eb.std(-1).unsqueeze(-1)).view(-1)
        opt.zero_grad()
        new_losses = ref_energy.calculate_energy(hand, obj, energy_fnc=fc, energy_names=names, method="gendexgrasp", svd_gain=0.1)
        new_energy = total(new_losses)
        new_energy.sum().backward()
        prop_pose = hand.hand_pose.detach().clone()
        prop_grad = hand.hand_pose.grad.detach().clone()
        with torch.no_grad():
            with _Recorder() as rec2:
                accept, T = opt.accept_step(energy, new_energy, None, z, 1.0)
            u_acc = rec2.log[0][1]
            energy[accept] = new_energy[accept]
        out.update({
            f"s{step}_u_switch": u_switch, f"s{step}_new_idx": new_idx, f"s{step}_u_accept": u_acc,
            f"s{step}_step_size": s.detach().clone(), f"s{step}_z": z.detach().clone(),
            f"s{step}_prop_pose": prop_pose, f"s{step}_prop_grad": prop_grad,
            f"s{step}_new_energy": new_energy.detach().clone(), f"s{step}_accept": accept.clone(),
            f"s{step}_temperature": T.detach().clone(),
            f"s{step}_hand_pose": hand.hand_pose.detach().clone(),
            f"s{step}_contact_idx": hand.contact_point_indices.clone(),
            f"s{step}_grad": hand.hand_pose.grad.detach().clone(),
            f"s{step}_energy": energy.detach().clone(),
            f"s{step}_ema": opt.ema_grad_hand_pose.detach().clone(),
        })
        print(f"mala step {step}: accept={accept.tolist()} mean E={energy.mean().item():.4f}")
    for i in range(n_obj):
        out[f"obj{i}_face_verts"] = fvs[i]
        out[f"obj{i}_surface_points"] = sps[i]
    np.savez_compressed(os.path.join(OUT, f"mala_{tag}.npz"), **to_np(out))


def _fit_iteration(ref_energy, opt, hand, obj, fc, names, w, energy, be, B, n, out, tag, reset=None, inject_grad=None):
    """One iteration in scripts/fit.py:399-458 order on the reference's MalaStar; everything observable goes into
    ``out`` under ``<tag>_*``.  reset = (mask, new_pose, new_idx): the re-initialisation branch fit.py:408-422 with the
    poses that initialize_convex_hull would write supplied (trimesh / pytorch3d are not importable); what is pinned is
    the reference's HandModel.set_parameters(env_mask=...) contract (restated by the oracle model),
    MalaStar.reset_envs and accept_step(reset_mask).  inject_grad: overwrite the accepted gradient first (NaN / inf /
    large entries for the clip_grad path, optimizer.py:211-215)."""
    if inject_grad is not None:
        with torch.no_grad():
            hand.hand_pose.grad.copy_(inject_grad)
        out[f"{tag}_grad_in"] = inject_grad.clone()
    with _Recorder() as rec:
        s = opt.try_step()
    u_switch, vals = rec.log[0][1], rec.log[1][1]
    mask = u_switch < 0.4
    new_idx = torch.zeros(B, n, dtype=torch.long)
    new_idx[mask] = vals
    eb = energy.view(-1, be)
    z = ((eb - eb.mean(-1).unsqueeze(-1)) / eb.std(-1).unsqueeze(-1)).view(-1)
    reset_mask = None
    if reset is not None:
        reset_mask, new_pose, new_cidx = reset
        if reset_mask.sum() > 0:  # fit.py:412 -- an all-false mask re-initialises nothing: an ordinary iteration, but
            hp = new_pose.clone()  # accept_step still receives the (all-false) mask (fit.py:447-453)
            hp.requires_grad_()
            hand.set_parameters(hp, new_cidx, env_mask=reset_mask)  # initializations.py:193
            opt.reset_envs(reset_mask)  # fit.py:422
        out.update({f"{tag}_reset_mask": reset_mask.clone(), f"{tag}_reset_pose": new_pose.clone(),
                    f"{tag}_reset_idx": new_cidx.clone()})
    opt.zero_grad()
    new_losses = ref_energy.calculate_energy(hand, obj, energy_fnc=fc, energy_names=names, method="gendexgrasp", svd_gain=0.1)
    new_energy = sum(w[k] * v for k, v in new_losses.items())
    new_energy.sum().backward()
    prop_pose = hand.hand_pose.detach().clone()
    prop_idx = hand.contact_point_indices.clone()
    prop_grad = hand.hand_pose.grad.detach().clone()
    with torch.no_grad():
        with _Recorder() as rec2:
            accept, T = opt.accept_step(energy, new_energy, reset_mask, z, 1.0)
        u_acc = rec2.log[0][1]
        energy[accept] = new_energy[accept]
    out.update({
        f"{tag}_u_switch": u_switch, f"{tag}_new_idx": new_idx, f"{tag}_u_accept": u_acc,
        f"{tag}_step_size": s.detach().clone(), f"{tag}_z": z.detach().clone(),
        f"{tag}_prop_pose": prop_pose, f"{tag}_prop_idx": prop_idx, f"{tag}_prop_grad": prop_grad,
        f"{tag}_new_energy": new_energy.detach().clone(), f"{tag}_accept": accept.clone(),
        f"{tag}_temperature": T.detach().clone(),
        f"{tag}_hand_pose": hand.hand_pose.detach().clone(),
        f"{tag}_contact_idx": hand.contact_point_indices.clone(),
        f"{tag}_grad": hand.hand_pose.grad.detach().clone(),
        f"{tag}_energy": energy.detach().clone(),
        f"{tag}_ema": opt.ema_grad_hand_pose.detach().clone(),
        f"{tag}_step": opt.step.clone(),
    })
    return accept


def gen_mala_ext(ref_energy, ref_opt):
    """Two more runs of the reference's MalaStar that close holes of the 5-step fixture:

    R  started at optimizer.step = 149, so that both decay exponents are non-zero and change inside the run
       (0.95^(step // 50): 2 -> 3, 0.95^(step // 30): 5), with a re-initialisation iteration in the middle
       (MalaStar.reset_envs + accept_step(reset_mask), optimizer.py:275-316; fit.py:408-422);
    C  clip_grad=True with NaN / +-inf / out-of-range entries injected into the accepted gradient
       (optimizer.py:211-215,235-250)."""
    hand_name, n_obj, be, n, seed = "allegro", 2, 4, 4, 23
    dtype = torch.float64
    names = ["E_dis", "E_fc", "E_pen", "E_spen", "E_joints"]
    w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    fc = _FcAdapter(mu=0.2, k=4, max_limit=20.0)
    B = n_obj * be
    out = dict(batch_size_each=be, n_obj=n_obj)
    for run, clip in (("R", False), ("C", True)):
        spec, hand, obj, hp, idx, fvs, sps = make_scene(hand_name, n_obj, be, n, dtype, seed, "sphere", n_surface=400)
        hand.set_parameters(hp.clone().requires_grad_(), idx)
        opt = ref_opt.MalaStar(hand, switch_possibility=0.4, starting_temperature=18, temperature_decay=0.95,
                               annealing_period=30, step_size=0.005, stepsize_period=50, mu=0.98, device="cpu",
                               batch_size=be, clip_grad=clip)
        losses = ref_energy.calculate_energy(hand, obj, energy_fnc=fc, energy_names=names, method="gendexgrasp", svd_gain=0.1)
        energy = sum(w[k] * v for k, v in losses.items())
        energy.sum().backward()
        out.update({f"{run}_hand_pose0": hand.hand_pose.detach().clone(), f"{run}_contact_idx0": idx.clone(),
                    f"{run}_energy0": energy.detach().clone(), f"{run}_grad0": hand.hand_pose.grad.detach().clone()})
        torch.manual_seed(seed + (1 if clip else 0))
        if run == "R":
            opt.zero_grad()
            opt.step += 149
            out["R_step0"] = opt.step.clone()
            # "fresh initial poses" of the re-initialised rows: another seeded scene (stands in for initialize_convex_hull)
            _, _, _, hp_new, idx_new, _, _ = make_scene(hand_name, n_obj, be, n, dtype, seed + 100, "sphere", n_surface=400)
            _fit_iteration(ref_energy, opt, hand, obj, fc, names, w, energy, be, B, n, out, "R_s1")
            eb = energy.view(-1, be)
            z = ((eb - eb.mean(-1).unsqueeze(-1)) / eb.std(-1).unsqueeze(-1)).view(-1)
            reset_mask = z > 0.5  # fit.py:409 with a threshold that selects a few rows of this small batch
            assert 0 < int(reset_mask.sum()) < B
            _fit_iteration(ref_energy, opt, hand, obj, fc, names, w, energy, be, B, n, out, "R_s2",
                           reset=(reset_mask, hp_new, idx_new))
            _fit_iteration(ref_energy, opt, hand, obj, fc, names, w, energy, be, B, n, out, "R_s3")
            # a scheduled re-initialisation whose mask comes out EMPTY (fit.py:408-412: no row has z > threshold -- the
            # threshold is put above the largest z-score a batch of `be` rows can reach, (be - 1) / sqrt(be)), followed by
            # one more ordinary iteration: neither may differ from an unscheduled iteration
            eb = energy.view(-1, be)
            z = ((eb - eb.mean(-1).unsqueeze(-1)) / eb.std(-1).unsqueeze(-1)).view(-1)
            empty = z > 1.6
            assert int(empty.sum()) == 0
            _fit_iteration(ref_energy, opt, hand, obj, fc, names, w, energy, be, B, n, out, "R_s4",
                           reset=(empty, hp_new, idx_new))
            _fit_iteration(ref_energy, opt, hand, obj, fc, names, w, energy, be, B, n, out, "R_s5")
            out["R_n_steps"] = 5
            out["R_empty_mask_threshold"] = 1.6
        else:
            # the first gradient is NOT zeroed here: the clip path needs a non-trivial gradient around the bad entries
            bad = hand.hand_pose.grad.detach().clone()
            bad[2, 0], bad[2, 4], bad[2, 10], bad[2, 11], bad[2, 12] = float("nan"), float("inf"), float("-inf"), 250.0, -1e4
            bad[5, 3], bad[5, 20] = float("nan"), 1e3
            _fit_iteration(ref_energy, opt, hand, obj, fc, names, w, energy, be, B, n, out, "C_s1", inject_grad=bad)
            _fit_iteration(ref_energy, opt, hand, obj, fc, names, w, energy, be, B, n, out, "C_s2")
            out["C_n_steps"] = 2
        if run == "R":
            for i in range(n_obj):
                out[f"obj{i}_face_verts"] = fvs[i]
                out[f"obj{i}_surface_points"] = sps[i]
    np.savez_compressed(os.path.join(OUT, "mala_ext_allegro_sphere_b8_n4.npz"), **to_np(out))
    print("mala_ext: R accepts", [out[f"R_s{i}_accept"].tolist() for i in (1, 2, 3, 4, 5)], "reset", out["R_s2_reset_mask"].tolist())
    print("mala_ext: C accepts", [out[f"C_s{i}_accept"].tolist() for i in (1, 2)])


def gen_init():
    """reference core/initializations.py::initialize_convex_hull executed in place.  trimesh / pytorch3d / transforms3d
    are not importable, so minimal stand-ins backed by the oracle's restatements (oracle/ref_cpu/init.py: hull sampling,
    farthest points, closest point on the hull, euler2mat) are registered under those names -- they stay PARITY
    UNPINNED; what the fixture pins is the reference's OWN arithmetic around them: look_at, distance / angle draws, the
    translation / rotation assembly, the rot6d layout, truncated-normal joints, random contact indices and the
    set_parameters(env_mask=...) hand-over (initializations.py:79-193)."""
    from ref_cpu import init as oinit

    tag, hand_name, n_obj, be, n = "allegro_sq_b12", "allegro", 2, 6, 4
    rec = {"u_face": [], "u_len": [], "p": [], "nrm": [], "trunc": []}

    class _Mesh:
        def __init__(self, vertices=None, faces=None, fv=None):
            if fv is None:
                fv = np.asarray(vertices, dtype=np.float64)[np.asarray(faces)]
            self.fv = np.asarray(fv, dtype=np.float64)
            self.vertices = self.fv.reshape(-1, 3).copy()
            self.faces = np.arange(len(self.vertices)).reshape(-1, 3)
            self.face_normals = oinit.face_normals(self.fv).numpy()
            self.nearest = self

        def remove_degenerate_faces(self):
            return np.ones(len(self.faces), dtype=bool)

        @property
        def convex_hull(self):
            return _Mesh(fv=oinit.convex_hull_faces(self.vertices))

        def on_surface(self, pts):  # trimesh.proximity: (closest, distance, triangle id)
            from ref_cpu import sdf as osdf
            _, _, _, cl = osdf.compute_sdf(torch.as_tensor(pts, dtype=torch.float64), torch.as_tensor(self.fv))
            return cl.numpy(), None, None

    def sample_surface_even(mesh, count):
        u_face, u_len = torch.rand(count, dtype=torch.float64), torch.rand(count, 2, dtype=torch.float64)
        pts, f = oinit.sample_surface(mesh.fv, u_face, u_len)
        rec["u_face"].append(u_face), rec["u_len"].append(u_len)
        return pts.numpy(), f.numpy()

    def sample_farthest_points(points, K, random_start_point=False):
        assert not random_start_point
        idx = oinit.farthest_points(points[0], K)
        return points[:, idx].float(), idx[None]  # float32 like the rest of the reference's pipeline

    def euler2mat(ai, aj, ak, axes="sxyz"):
        assert axes == "rxyz"
        return oinit.euler2mat_rxyz(*(torch.as_tensor([float(v)], dtype=torch.float64) for v in (ai, aj, ak)))[0].numpy()

    tm = types.ModuleType("trimesh")
    tm.Trimesh = _Mesh
    tm.sample = types.SimpleNamespace(sample_surface_even=sample_surface_even)
    p3 = types.ModuleType("pytorch3d")
    p3.ops = types.ModuleType("pytorch3d.ops")
    p3.ops.sample_farthest_points = sample_farthest_points
    p3.structures = types.ModuleType("pytorch3d.structures")
    t3 = types.ModuleType("transforms3d")
    t3.euler = types.SimpleNamespace(euler2mat=euler2mat)
    saved = {k: sys.modules.get(k) for k in ("trimesh", "pytorch3d", "pytorch3d.ops", "pytorch3d.structures", "transforms3d")}
    sys.modules.update({"trimesh": tm, "pytorch3d": p3, "pytorch3d.ops": p3.ops, "pytorch3d.structures": p3.structures,
                        "transforms3d": t3})
    try:
        ref_init = load_ref("_ref_initializations", "core/initializations.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    spec = get_hand_spec(hand_name)
    hand = omodels.OracleHand(spec, dtype=torch.float32)
    hand.up_axis = torch.tensor(spec.up_axis, dtype=torch.float32)
    hand.forward_axis = torch.tensor(spec.forward_axis, dtype=torch.float32)
    fvs = [meshes.superquadric(31 + i, 24, 12) for i in range(n_obj)]
    obj = types.SimpleNamespace(object_mesh_list=[_Mesh(fv=f) for f in fvs], batch_size_each=be, device="cpu",
                                object_scale_tensor=torch.ones(n_obj, be), data_root_path="/data/synthetic")
    args = types.SimpleNamespace(n_contact=n, **oinit.DEFAULT_ARGS)
    torch.manual_seed(7)
    # record torch.rand (distance / rotate / pitch / tilt, in that order per object), torch.randint (contact indices) and
    # the truncated-normal joint columns
    tn = torch.nn.init.trunc_normal_

    def trunc(t, mean, std, a, b):
        r = tn(t, float(mean), float(std), float(a), float(b))
        rec["trunc"].append(r.clone())
        return r

    torch.nn.init.trunc_normal_ = trunc
    try:
        with _Recorder() as r0:
            ref_init.initialize_convex_hull(hand, obj, args)
    finally:
        torch.nn.init.trunc_normal_ = tn
    rands = [v for k, v in r0.log if k == "rand" and v.dim() == 1 and v.shape[0] == be]
    assert len(rands) == 4 * n_obj
    u_pose = torch.cat([torch.stack(rands[4 * i : 4 * i + 4], dim=1) for i in range(n_obj)])  # (B,4)
    idx0 = [v for k, v in r0.log if k == "randint"][0]
    pose0 = hand.hand_pose.detach().clone()
    out = dict(n_obj=n_obj, batch_size_each=be, n_contact=n, hand_pose=pose0, contact_idx=hand.contact_point_indices.clone(),
               randint=idx0, u_pose=u_pose, joints=torch.stack(rec["trunc"], dim=1),
               u_face=torch.stack(rec["u_face"]), u_len=torch.stack(rec["u_len"]))
    for i in range(n_obj):
        out[f"obj{i}_face_verts"] = fvs[i]
        out[f"obj{i}_hull_face_verts"] = obj.object_mesh_list[i].convex_hull.fv
    # second call with an env_mask (fit.py:421): only the masked rows change
    mask = torch.zeros(n_obj * be, dtype=torch.bool)
    mask[[1, 4, 9]] = True
    rec2 = {"u_face": [], "u_len": [], "trunc": []}
    rec.update(rec2)
    torch.nn.init.trunc_normal_ = trunc
    try:
        with _Recorder() as r1:
            ref_init.initialize_convex_hull(hand, obj, args, env_mask=mask)
    finally:
        torch.nn.init.trunc_normal_ = tn
    out.update(env_mask=mask, hand_pose_masked=hand.hand_pose.detach().clone(),
               contact_idx_masked=hand.contact_point_indices.clone())
    assert torch.equal(out["hand_pose_masked"][~mask], pose0[~mask]) and not torch.equal(out["hand_pose_masked"][mask], pose0[mask])
    np.savez_compressed(os.path.join(OUT, f"init_{tag}.npz"), **to_np(out))
    print("init: translation norms", pose0[:, :3].norm(dim=1).numpy().round(3).tolist())


def gen_alt_metrics():
    """reference metrics/ops/dexgrasp.py and tdg.py (plain torch, import here): energies and contact-point gradients."""
    dex = load_ref("_ref_dexgrasp", "metrics/ops/dexgrasp.py")
    tdg = load_ref("_ref_tdg", "metrics/ops/tdg.py")
    out = {}
    for n, B, seed in ((4, 8, 0), (12, 16, 1), (20, 6, 2)):
        torch.manual_seed(seed)
        d = torch.nn.functional.normalize(torch.randn(B, n, 3), dim=-1)
        pts = d * (0.05 + 0.01 * torch.randn(B, n, 1))
        nrm = torch.nn.functional.normalize(-d + 0.4 * torch.randn(B, n, 3), dim=-1)
        nrm[0, 0] = torch.tensor([0.0, 1.0, 0.0])  # the |n.y| > 0.99 branch of utils_1axis_to_3axes
        cog = 0.005 * torch.randn(B, 3)
        tag = f"n{n}"
        out.update({f"{tag}_contact_pts": pts, f"{tag}_contact_normals": nrm, f"{tag}_cog": cog})
        for tw in (0.0, 1.0, 5.0):
            p = pts.clone().requires_grad_()
            e, ones = dex.DexgraspSpanMetric()(p, nrm, cog, torque_weight=tw, with_solution=True)
            (g,) = torch.autograd.grad(e.sum(), p, allow_unused=True)
            out[f"{tag}_dex_tw{int(tw)}_e"] = e.detach()
            out[f"{tag}_dex_tw{int(tw)}_grad"] = torch.zeros_like(pts) if g is None else g
        np.random.seed(100 + seed)
        m = tdg.TDGSpanMetric(device="cpu")
        p = pts.clone().requires_grad_()
        e, _ = m(p, nrm, cog)
        e.sum().backward()
        out.update({f"{tag}_tdg_directions": m.tdg_energy.target_direction_6D[0, :, :3], f"{tag}_tdg_e": e.detach(),
                    f"{tag}_tdg_grad": p.grad})
        assert float(m.tdg_energy.target_direction_6D[0, :, 3:].abs().max()) == 0.0
        print(f"alt metrics n={n}: dexgrasp tw=1 {out[f'{tag}_dex_tw1_e'][:2].tolist()} tdg {e[:2].tolist()}")
    np.savez_compressed(os.path.join(OUT, "alt_metrics.npz"), **to_np(out))


def gen_annealing(ref_energy, ref_opt, ref_dex):
    """reference AnnealingDexGraspNet (core/optimizer.py:11-149) with energy_type dexgrasp (fit.py:337,343-345) in fit.py
    order, every draw recorded; started at step 61 so that both decay exponents are non-zero."""
    hand_name, n_obj, be, n, seed, n_steps = "allegro", 2, 4, 4, 29, 3
    dtype = torch.float32  # the reference's dexgrasp metric is float32-only (dexgrasp.py:19-31)
    spec, hand, obj, hp, idx, fvs, sps = make_scene(hand_name, n_obj, be, n, dtype, seed, "sphere", n_surface=400)
    B = n_obj * be
    hand.set_parameters(hp.clone().requires_grad_(), idx)
    names = ["E_dis", "E_fc", "E_pen", "E_spen", "E_joints"]
    w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    fc = ref_dex.DexgraspSpanMetric()
    opt = ref_opt.AnnealingDexGraspNet(hand, switch_possibility=0.4, starting_temperature=18, temperature_decay=0.95,
                                       annealing_period=30, step_size=0.005, stepsize_period=50, mu=0.98, device="cpu")
    losses = ref_energy.calculate_energy(hand, obj, energy_fnc=fc, energy_names=names, method="gendexgrasp", svd_gain=0.1)
    energy = sum(w[k] * v for k, v in losses.items())
    energy.sum().backward()
    out = dict(hand_pose0=hand.hand_pose.detach().clone(), contact_idx0=idx.clone(), energy0=energy.detach().clone(),
               E_fc0=losses["E_fc"].detach().clone(), batch_size_each=be, n_obj=n_obj, n_steps=n_steps, step0=61)
    opt.zero_grad()
    opt.step = 61
    torch.manual_seed(seed)
    for step in range(1, n_steps + 1):
        with _Recorder() as rec:
            s = opt.try_step()
        u_switch, vals = rec.log[0][1], rec.log[1][1]
        mask = u_switch < 0.4
        new_idx = torch.zeros(B, n, dtype=torch.long)
        new_idx[mask] = vals
        opt.zero_grad()
        new_losses = ref_energy.calculate_energy(hand, obj, energy_fnc=fc, energy_names=names, method="gendexgrasp", svd_gain=0.1)
        new_energy = sum(w[k] * v for k, v in new_losses.items())
        new_energy.sum().backward()
        prop_pose = hand.hand_pose.detach().clone()
        prop_idx = hand.contact_point_indices.clone()
        with torch.no_grad():
            with _Recorder() as rec2:
                accept, T = opt.accept_step(energy, new_energy, None, None, 1.0)
            energy[accept] = new_energy[accept]
        out.update({
            f"s{step}_u_switch": u_switch, f"s{step}_new_idx": new_idx, f"s{step}_u_accept": rec2.log[0][1],
            f"s{step}_step_size": torch.as_tensor(s).detach().clone().expand(B).clone(), f"s{step}_prop_pose": prop_pose,
            f"s{step}_prop_idx": prop_idx,
            f"s{step}_new_energy": new_energy.detach().clone(), f"s{step}_new_E_fc": new_losses["E_fc"].detach().clone(),
            f"s{step}_accept": accept.clone(), f"s{step}_temperature": torch.as_tensor(T).detach().clone().expand(B).clone(),
            f"s{step}_hand_pose": hand.hand_pose.detach().clone(), f"s{step}_contact_idx": hand.contact_point_indices.clone(),
            f"s{step}_grad": hand.hand_pose.grad.detach().clone(), f"s{step}_energy": energy.detach().clone(),
            f"s{step}_ema": opt.ema_grad_hand_pose.detach().clone().unsqueeze(0).expand(B, -1).clone(),
            f"s{step}_step": torch.full((B,), int(opt.step), dtype=torch.long),
        })
        print(f"annealing step {step}: accept={accept.tolist()}")
    for i in range(n_obj):
        out[f"obj{i}_face_verts"] = fvs[i]
        out[f"obj{i}_surface_points"] = sps[i]
    np.savez_compressed(os.path.join(OUT, "annealing_dexgrasp_allegro_sphere_b8_n4.npz"), **to_np(out))


def main():
    os.makedirs(OUT, exist_ok=True)
    scipy_solver, registry = ref_metrics()
    gen_kat(scipy_solver)
    gen_span(registry)
    ref_energy = load_ref("_ref_energy", "core/energy.py")
    ref_opt = load_ref("_ref_optimizer", "core/optimizer.py")
    gen_energy(ref_energy)
    gen_mala(ref_energy, ref_opt)
    gen_mala_ext(ref_energy, ref_opt)
    gen_init()
    gen_alt_metrics()
    gen_annealing(ref_energy, ref_opt, load_ref("_ref_dexgrasp2", "metrics/ops/dexgrasp.py"))


if __name__ == "__main__":
    main()
