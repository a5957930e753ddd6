#!/usr/bin/env python3
"""Throughput of the reference's PLUGIN SURFACE on the HIP back end (what an unchanged scripts/fit.py would call):

  (a) ``graspqp_amd.torchsdf.compute_sdf(points, face_verts)`` at the reference's shapes -- the per-link calls of
      HandModel.cal_distance (reference core/hand_model.py:914-953: one call per link mesh with N = batch * 2500 points
      in the link frame) and the object call of ObjectModel.cal_distance (core/object_model.py:217-220: N = batch *
      n_contact contact points against the object mesh);
  (b) ``QPFunction(maxIter=12, eps=5e-2)(Q, p, G, h)`` / ``SQPLsqSolver.solve`` the way metrics/solver/qp_solver.py:101-125
      drives them;
  (c) a scripts/fit.py:399-458-shaped loop on HandModel / ObjectModel / calculate_energy / MalaStar.

The HBM figure of (a) is physical: N*(12 B in + 32 B out) + 36 B * F bytes really move per call.
Used by bench.py (``plugin_surface`` block, after the timed region) and stand-alone:  python tools/plugin_surface.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0


def _time_us(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def link_frame_points(spec, hand, hp, idx, surf, batch_each):
    """x_local of reference hand_model.py:890,915-917 for every link: (L) tensors of (B*P,3)."""
    from graspqp_amd import ops

    Rg, LT, *_ = ops.fk_contacts(hp, idx, hand)
    x = surf.repeat_interleave(batch_each, dim=0)  # (B,P,3), object_model.py:182-184
    x = (x - hp[:, None, 0:3]) @ Rg
    out = []
    for l in range(spec.n_links):
        xl = (x - LT[:, l, :, 3].unsqueeze(1)) @ LT[:, l, :, :3]
        out.append(xl.reshape(-1, 3).contiguous())
    return out


def sdf_calls(spec, hand, hp, idx, surf, batch_each, obj_fv, n_contact, reps=10):
    from graspqp_amd import ops
    from graspqp_amd.torchsdf import compute_sdf

    pts = link_frame_points(spec, hand, hp, idx, surf, batch_each)
    rows, tot_us, tot_bytes = [], 0.0, 0
    for l in range(spec.n_links):
        fv = torch.tensor(np.ascontiguousarray(spec.link_faces(l)), dtype=torch.float32, device="cuda")
        if fv.shape[0] == 0:
            continue
        x = pts[l]
        med, mn = _time_us(lambda: compute_sdf(x, fv), reps=reps)
        nbytes = x.shape[0] * 44 + fv.shape[0] * 36
        rows.append({"link": spec.link_names[l] if hasattr(spec, "link_names") else l, "N": int(x.shape[0]),
                     "F": int(fv.shape[0]), "us": med, "us_min": mn, "moved_bytes": nbytes,
                     "gbps": nbytes / med / 1e3, "hbm_frac": nbytes / med / 1e3 / HBM_PEAK_GBPS})
        tot_us += med
        tot_bytes += nbytes
    # the object call: contact points of all rows against the object mesh
    _, _, cp, *_ = ops.fk_contacts(hp, idx, hand)
    q = cp.reshape(-1, 3).contiguous()
    ofv = torch.tensor(obj_fv, dtype=torch.float32, device="cuda")
    med, mn = _time_us(lambda: compute_sdf(q, ofv), reps=reps)
    nb = q.shape[0] * 44 + ofv.shape[0] * 36
    obj = {"N": int(q.shape[0]), "F": int(ofv.shape[0]), "us": med, "us_min": mn, "moved_bytes": nb, "gbps": nb / med / 1e3,
           "hbm_frac": nb / med / 1e3 / HBM_PEAK_GBPS}
    # first call on a mesh the back end has not seen (set-up the drop-in pays once per face_verts tensor)
    ofv2 = ofv.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    compute_sdf(q, ofv2)
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - t0) * 1e3
    return {"hand_links": rows, "hand_links_total_us": tot_us, "hand_links_calls": len(rows),
            "hand_links_moved_bytes": tot_bytes, "hand_links_gbps": tot_bytes / tot_us / 1e3,
            "hand_links_hbm_frac": tot_bytes / tot_us / 1e3 / HBM_PEAK_GBPS, "object": obj,
            "object_first_call_ms": first_ms,
            "bytes_rule": "N*(12 in + 4+4+12+12 out) + 36*F per call, all of them really cross HBM"}


def grasp_matrix(pts, nrm, cog, k, mu=0.2, torque_weight=5.0):
    """(B,6,n*k) friction-cone grasp matrix in plain torch -- input generator of the QP timings (the formulas of reference
    metrics/ops/span.py:263-295,362-375: edges mu*(cos a t1 + sin a t2) + sqrt(1-mu^2) n, divided by k)."""
    b1 = torch.full_like(nrm, 3.0 ** -0.5)
    dot = (b1 * nrm).sum(-1) / (nrm.norm(dim=-1) + 1e-6)
    b1[..., 1] = torch.where(dot > 0.9, b1[..., 1] - 2.0, b1[..., 1])
    t1 = torch.linalg.cross(nrm, b1)
    t2 = torch.linalg.cross(nrm, t1)
    ang = torch.arange(k, device=pts.device, dtype=pts.dtype) * (2 * np.pi / k)
    f = (mu * (torch.cos(ang)[None, None, :, None] * t1[:, :, None] + torch.sin(ang)[None, None, :, None] * t2[:, :, None])
         + (1 - mu * mu) ** 0.5 * nrm[:, :, None]) / k
    tau = torque_weight * torch.linalg.cross((pts - cog[:, None])[:, :, None].expand_as(f), f)
    return torch.cat([f, tau], -1).flatten(1, 2).transpose(1, 2).contiguous()


def qp_calls(sizes=((256, 48), (2048, 48), (32768, 48), (256, 96), (2048, 96), (32768, 96)), reps=10):
    """qp_solver.py:101-125: Q = A'A + 1e-4 I, p = -A'b (= 0), G = [I; -I], h = [max; -min], QPFunction(maxIter=12, eps=5e-2)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _scenes import hetero_contacts

    from graspqp_amd import ops
    from graspqp_amd.metrics import QPFunction, SQPLsqSolver

    out = []
    for B, nz in sizes:
        k = 4 if nz == 48 else 8
        n = nz // k
        b0 = min(B, 2048)
        pts, nrm, cog = hetero_contacts(b0, n, 7, torch.float32)
        A = grasp_matrix(pts.cuda(), nrm.cuda(), cog.cuda(), k)
        A = A.repeat((B + b0 - 1) // b0, 1, 1)[:B].contiguous()
        b = torch.zeros(B, 6, device="cuda")
        rec = {"B": B, "nz": nz}
        solver = SQPLsqSolver.from_mat(A, b)
        med, mn = _time_us(lambda: solver.solve(A, b, min_bound=1.0, max_bound=21.0, return_solution=True), reps=reps)
        rec.update(sqplsq_solve_us=med, sqplsq_problems_per_s=B / med * 1e6)
        Ag = A.clone().requires_grad_()

        def fb():
            v, _ = solver.solve(Ag, b, min_bound=1.0, max_bound=21.0, return_solution=True)
            v.sum().backward()
            Ag.grad = None

        med, mn = _time_us(fb, reps=reps)
        rec.update(sqplsq_solve_fwd_bwd_us=med)
        try:
            Q = A.transpose(1, 2) @ A + 1e-4 * torch.eye(nz, device="cuda")
            p = torch.zeros(B, nz, device="cuda")
            G = torch.cat([torch.eye(nz), -torch.eye(nz)]).cuda()
            h = torch.cat([21.0 * torch.ones(B, nz), -torch.ones(B, nz)], 1).cuda()
            qf = QPFunction(verbose=-1, maxIter=12, eps=5e-2)
            med, mn = _time_us(lambda: qf(Q, p, G, h, torch.empty(0, device="cuda"), torch.empty(0, device="cuda")), reps=reps)
            rec.update(qpfunction_us=med, qpfunction_problems_per_s=B / med * 1e6)
            Qg = Q.clone().requires_grad_()

            def qfb():
                x = qf(Qg, p, G, h)
                x.sum().backward()
                Qg.grad = None

            med, mn = _time_us(qfb, reps=reps)
            rec.update(qpfunction_fwd_bwd_us=med)
        except Exception as e:  # e.g. nz above the dense kernels' limit
            rec.update(qpfunction_error=f"{type(e).__name__}: {str(e)[:160]}")
        out.append(rec)
    return out


def class_surface_loop(spec, fv, sp, batch, n_contact, hp, idx, iters=60, warm=None):
    """scripts/fit.py:399-458 on the class surface (autograd route): evals/s, no resets, no export.  The loop is host-bound and
    keeps the GPU ~40 % busy: the first ~100 iterations after a pause run at lower clocks (0.85 instead of 0.56 ms each), so
    the warm-up is a third of the timed count (at least 10)."""
    warm = max(10, iters // 3) if warm is None else warm
    from graspqp_amd.core.energy import calculate_energy
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.core.object_model import ObjectModel
    from graspqp_amd.core.optimizer import MalaStar
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF

    hm = HandModel(spec, "cuda")
    om = ObjectModel(batch_size_each=batch, num_samples=sp.shape[0])
    om.initialize_from_meshes([fv], surface_points_list=[sp])
    hm.set_parameters(hp.clone().requires_grad_(), idx.clone())
    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
    w = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}
    names = list(w)

    def total():
        losses = calculate_energy(hm, om, energy_fnc=fn, energy_names=names, svd_gain=0.1)
        return sum(w[k] * v for k, v in losses.items())

    opt = MalaStar(hm, switch_possibility=0.4, device="cuda", batch_size=batch)
    energy = total()
    energy.sum().backward()
    opt.zero_grad()
    energy = energy.detach().clone()

    def one():
        opt.try_step()
        eb = energy.view(-1, batch)
        z = ((eb - eb.mean(-1, keepdim=True)) / eb.std(-1, keepdim=True)).view(-1)
        opt.zero_grad()
        new_energy = total()
        new_energy.sum().backward()
        with torch.no_grad():
            opt.accept_step(energy, new_energy, None, z, 1.0)

    for _ in range(warm):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"evals_per_s": batch * iters / dt, "ms_per_iteration": dt / iters * 1e3, "iterations": iters,
            "warm_up_iterations": warm,
            "route": "HandModel / ObjectModel / calculate_energy / MalaStar (autograd, eager launches, host-driven)",
            "mean_energy": float(energy.mean())}


def measure(spec, fv, sp, batch, n_contact, hp, idx, qp_sizes=None, loop_iters=60, reps=10):
    from graspqp_amd import ops

    hand = ops.HandHandle(spec)
    surf = torch.tensor(sp, dtype=torch.float32, device="cuda")[None].contiguous()
    hp, idx = hp.cuda().float().contiguous(), idx.cuda().contiguous()
    res = {"compute_sdf": sdf_calls(spec, hand, hp, idx, surf, batch, fv, n_contact, reps=reps)}
    res["qp"] = qp_calls(**({"sizes": qp_sizes} if qp_sizes else {}), reps=reps)
    res["fit_loop_class_surface"] = class_surface_loop(spec, fv, sp, batch, n_contact, hp, idx, iters=loop_iters)
    return res


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    import bench
    from graspqp_amd.hands import get_hand_spec
    from graspqp_amd.utils import meshes

    spec = get_hand_spec("allegro")
    fv = meshes.superquadric(0)
    sp = meshes.surface_points(fv, 2500, oversample=4, seed=42)
    hp, idx = bench.make_initial_state(spec, fv, 256, 12, 1000)
    if len(sys.argv) > 1 and sys.argv[1] == "sdf":  # A/B runs of the SDF paths only
        from graspqp_amd import ops

        hand = ops.HandHandle(spec)
        surf = torch.tensor(sp, dtype=torch.float32, device="cuda")[None].contiguous()
        r = sdf_calls(spec, hand, hp.cuda().float().contiguous(), idx.cuda().contiguous(), surf, 256, fv, 12)
        print(json.dumps({"lib": os.environ.get("GRASPQP_HIP_LIB", "default"), "hand_links_total_us": r["hand_links_total_us"],
                          "per_link_us": [round(x["us"], 1) for x in r["hand_links"]], "object_us": r["object"]["us"]}), flush=True)
    elif len(sys.argv) > 1 and sys.argv[1] == "loop":  # the class-surface loop only (under rocprofv3: its kernel list)
        print(json.dumps(class_surface_loop(spec, fv, sp, 256, 12, hp.cuda().float().contiguous(), idx.cuda().contiguous(),
                                            iters=int(sys.argv[2]) if len(sys.argv) > 2 else 200)), flush=True)
    else:
        print(json.dumps(measure(spec, fv, sp, 256, 12, hp, idx)), flush=True)
