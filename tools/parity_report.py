#!/usr/bin/env python3
"""Parity of the HIP path against the fp64 oracle IN NUMBERS, per BASELINE configuration, with the oracle's own
fp32-vs-fp64 error beside every figure (SURVEY 8c contract: total energy <= 1e-4 rel per row; E_fc <= 1e-4 at the median and
<= 2x the oracle's fp32 noise at p99 / max; gradient <= 1e-3 norm-wise through the kinematics, looser through the QP).

Per configuration (per-rank size), on the accepted state after four MALA* iterations with a few rows pushed into the object:
  * E_fc and its contact-point gradient of ALL rows, HIP vs oracle fp64 run on the same B x n contacts (same batch
    composition, so qpth's batch-global stop rule sees the same input); oracle fp32 (reduced form and qpth's block form)
    vs oracle fp64 = the noise floor of the algorithm in the reference's precision;
  * total energy and d total / d hand_pose of a row sample (all rows when B <= 256): the oracle evaluates the kinematics
    and E_fc of the whole batch but E_pen / E_spen / E_dis / E_joints only on the sample (E_pen is the expensive term).

    python tools/parity_report.py [--configs 0,1,2,3,4a,4b] [--sample 32] > profiles/r03_parity_report.json
TEST INFRASTRUCTURE (imports oracle/); runs on the GPU box.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np
import torch

CONFIGS = {
    "0": dict(label="configs[0] Allegro, sphere, batch 4, n_contact 4", hand="allegro", mesh="sphere", n_obj=1, be=4, n=4, k=4),
    "1": dict(label="configs[1] Allegro, 1 mesh, batch 256, n_contact 12", hand="allegro", mesh="sq", n_obj=1, be=256, n=12, k=4),
    "2": dict(label="configs[2] Shadow Hand, 8 meshes x 512, n_contact 16", hand="shadow_hand", mesh="sq", n_obj=8, be=512, n=16, k=4),
    "3": dict(label="configs[3] per rank: Allegro, 8 meshes x 256, n_contact 12", hand="allegro", mesh="sq", n_obj=8, be=256, n=12, k=4),
    "4a": dict(label="configs[4] per rank, Allegro: 32 meshes x 1024, 8-edge cones", hand="allegro", mesh="sq", n_obj=32, be=1024, n=12, k=8),
    "4b": dict(label="configs[4] per rank, Robotiq-3F: 32 meshes x 1024, 8-edge cones", hand="robotiq3", mesh="sq", n_obj=32, be=1024, n=12, k=8),
}
W = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}


def stats(rel):
    rel = np.asarray(rel, dtype=np.float64)
    return {"p50": float(np.percentile(rel, 50)), "p99": float(np.percentile(rel, 99)), "max": float(rel.max()), "n": int(rel.size)}


def rel_err(a, b, floor=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def row_norm_err(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(len(a), -1), np.asarray(b, dtype=np.float64).reshape(len(b), -1)
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)


def oracle_fc(cpts, onrm, cog, k, dtype, box_form):
    from ref_cpu import qp as oqp
    from ref_cpu import span as ospan

    po = cpts.detach().to(dtype).clone().requires_grad_()
    e, _ = ospan.e_fc(po, onrm.to(dtype), cog.to(dtype), k=k, box_form=box_form)
    nit = oqp.LAST["n_iter"]
    e.sum().backward()
    return e.detach().double().numpy(), po.grad.double().numpy(), int(nit)


def oracle_rows(spec, fvs, sps, be, pose, cidx, rows, k, dtype):
    """Energies (5 terms) of ``rows`` and d total / d hand_pose[rows]: kinematics + E_fc on the whole batch, the other
    terms on the sample (models.OracleHand twice over the same leaf)."""
    import ref_cpu
    from ref_cpu import models as omodels
    from ref_cpu import span as ospan

    B = pose.shape[0]
    leaf = pose.to(dtype).clone().requires_grad_()
    full = omodels.OracleHand(spec, dtype)
    full.set_parameters(leaf, cidx)
    obj_full = omodels.OracleObject(fvs, sps, be, dtype)
    _, cn = obj_full.cal_distance(full.contact_points)
    e_fc_all, _ = ospan.e_fc(full.contact_points, cn, obj_full.cog, k=k, box_form=True)
    r = torch.as_tensor(rows)
    terms = {"E_fc": e_fc_all[r]}
    tot = W["E_fc"] * e_fc_all[r]
    parts = {kk: [] for kk in ("E_dis", "E_pen", "E_spen", "E_joints")}
    sub = omodels.OracleHand(spec, dtype)
    for i, row in enumerate(rows):  # one row at a time: every row has its own object
        if i % 4 == 0:
            print(f"[parity]   oracle rows {i}/{len(rows)} ({dtype})", file=sys.stderr, flush=True)
        o = row // be
        oo = omodels.OracleObject([fvs[o]], [sps[o]], 1, dtype)
        sub.set_parameters(leaf[row : row + 1], cidx[row : row + 1])
        lo = ref_cpu.calculate_energy(sub, oo, box_form=True, k=k, e_fc_fn=lambda *a: torch.zeros(1, dtype=dtype))
        for kk in parts:
            parts[kk].append(lo[kk])
    for kk in parts:
        terms[kk] = torch.cat(parts[kk])
        tot = tot + W[kk] * terms[kk]
    tot.sum().backward()
    return ({kk: v.detach().double().numpy() for kk, v in terms.items()}, tot.detach().double().numpy(),
            leaf.grad[r].double().numpy())


def run_config(key, sample):
    from bench import make_initial_state
    from graspqp_amd import ops
    from graspqp_amd.hands import get_hand_spec
    from graspqp_amd.stepper import GraspStepper
    from graspqp_amd.utils import meshes

    c = CONFIGS[key]
    t0 = time.time()
    spec = get_hand_spec(c["hand"])
    n_obj, be, n, k = c["n_obj"], c["be"], c["n"], c["k"]
    if c["mesh"] == "sphere":
        fvs = [meshes.icosphere(3, 0.05)]
    else:
        fvs = [meshes.superquadric(o) for o in range(n_obj)]
    sps = [meshes.surface_points(f, 2500, oversample=4, seed=42) for f in fvs]
    B = n_obj * be
    hand = ops.HandHandle(spec)
    ms = ops.MeshSet(fvs)
    surf = torch.tensor(np.stack(sps))
    hps, idxs = zip(*[make_initial_state(spec, f, be, n, 1000 + o) for o, f in enumerate(fvs)])
    hp, idx = torch.cat(hps).cuda(), torch.cat(idxs).cuda()
    hp[:: max(B // 21, 3), :3] *= 0.45  # some rows inside their object: E_pen and the inside branch of E_dis non-trivial
    fc_cfg = {"n_cone_vecs": k}
    st = GraspStepper(hand, ms, surf, be, n, fc_cfg=fc_cfg, seed=5)
    st.reset(hp, idx)
    for _ in range(4):
        st.step()
    st.flush()
    torch.cuda.synchronize()
    pose, cidx = st.hand_pose.clone(), st.contact_idx.clone()
    terms, total, grad = st.evaluate(pose, cidx)
    torch.cuda.synchronize()
    print(f"[parity] {key}: state ready after {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    out = {"label": c["label"], "rows": B, "n_contact": n, "n_cone_vecs": k, "hip_n_iter": int(st.n_iter.item())}
    # ---- E_fc of all rows -------------------------------------------------------------------------------------------
    s_fc = GraspStepper(hand, ms, surf, be, n, fc_cfg=fc_cfg, weights={"E_dis": 0.0, "E_fc": 1.0, "E_pen": 0.0, "E_spen": 0.0, "E_joints": 0.0})
    t_fc, _, _ = s_fc.evaluate(pose, cidx)
    torch.cuda.synchronize()
    cp, on, cg = s_fc.cpts.cpu().double(), s_fc.obj_normal.cpu().double(), s_fc.cog.cpu().double()
    e64, g64, it64 = oracle_fc(cp, on, cg, k, torch.float64, True)
    e32, g32, it32 = oracle_fc(cp, on, cg, k, torch.float32, True)
    fc = {"hip_vs_fp64": stats(rel_err(t_fc["E_fc"].cpu().numpy(), e64)),
          "oracle_fp32_vs_fp64": stats(rel_err(e32, e64)),
          "grad_hip_vs_fp64_rowwise": stats(row_norm_err(s_fc.g_cpts.cpu().numpy(), g64)),
          "grad_oracle_fp32_vs_fp64_rowwise": stats(row_norm_err(g32, g64)),
          "grad_hip_vs_fp64_batch_norm": float(np.linalg.norm(s_fc.g_cpts.cpu().numpy() - g64) / np.linalg.norm(g64)),
          "grad_oracle_fp32_vs_fp64_batch_norm": float(np.linalg.norm(g32 - g64) / np.linalg.norm(g64)),
          "n_iter": {"hip": int(s_fc.n_iter.item()), "oracle_fp64": it64, "oracle_fp32": it32}}
    print(f"[parity] {key}: E_fc of all rows done after {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    if B <= 4096:  # qpth's own 2nz x 2nz block form in fp32: what the reference runs
        e32q, g32q, it32q = oracle_fc(cp, on, cg, k, torch.float32, False)
        fc["oracle_fp32_qpth_form_vs_fp64"] = stats(rel_err(e32q, e64))
        fc["grad_oracle_fp32_qpth_form_vs_fp64_rowwise"] = stats(row_norm_err(g32q, g64))
        fc["n_iter"]["oracle_fp32_qpth_form"] = it32q
    out["E_fc"] = fc
    # ---- total energy / gradient of a row sample --------------------------------------------------------------------
    if B <= 256:
        rows = list(range(B))
    else:
        pen = torch.nonzero(terms["E_pen"] > 1e-4).flatten().tolist()
        rng = np.random.default_rng(0)
        rows = sorted(set(rng.choice(B, sample, replace=False).tolist()) | set(pen[:4]) | {0, B - 1})
    t64, tot64, gr64 = oracle_rows(spec, fvs, sps, be, pose.cpu(), cidx.cpu(), rows, k, torch.float64)
    t32, tot32, gr32 = oracle_rows(spec, fvs, sps, be, pose.cpu(), cidx.cpu(), rows, k, torch.float32)
    r = torch.as_tensor(rows)
    out["sample_rows"] = len(rows)
    out["rows_with_penetration_in_sample"] = int((t64["E_pen"] > 1e-6).sum())
    out["total"] = {"hip_vs_fp64": stats(rel_err(total.cpu().numpy()[rows], tot64)), "oracle_fp32_vs_fp64": stats(rel_err(tot32, tot64))}
    out["terms"] = {}
    for kk in ("E_dis", "E_fc", "E_pen", "E_spen", "E_joints"):
        a64 = t64[kk]
        scale = np.maximum(np.abs(a64), 1e-6)  # terms that are exactly zero: absolute error against 1e-6
        out["terms"][kk] = {"hip_vs_fp64": stats(np.abs(terms[kk].cpu().numpy()[rows] - a64) / scale),
                            "oracle_fp32_vs_fp64": stats(np.abs(t32[kk] - a64) / scale)}
    out["grad_hand_pose"] = {"hip_vs_fp64_rowwise": stats(row_norm_err(grad.cpu().numpy()[rows], gr64)),
                             "oracle_fp32_vs_fp64_rowwise": stats(row_norm_err(gr32, gr64)),
                             "hip_vs_fp64_sample_norm": float(np.linalg.norm(grad.cpu().numpy()[rows] - gr64) / np.linalg.norm(gr64)),
                             "oracle_fp32_vs_fp64_sample_norm": float(np.linalg.norm(gr32 - gr64) / np.linalg.norm(gr64))}
    out["seconds"] = round(time.time() - t0, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="0,1,2,3,4a,4b")
    ap.add_argument("--sample", type=int, default=32)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    res = {"contract": "SURVEY 8c: total <= 1e-4 rel per row; E_fc <= 1e-4 median, <= 2x oracle-fp32 noise at p99/max; gradient <= 1e-3 "
                       "(kinematics) / 2e-2 (through the QP) norm-wise", "weights": W, "configs": {}}
    for key in args.configs.split(","):
        res["configs"][key] = run_config(key, args.sample)
        print(f"[parity] {key}: {json.dumps(res['configs'][key])[:400]}", file=sys.stderr, flush=True)
        if args.out:
            json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
