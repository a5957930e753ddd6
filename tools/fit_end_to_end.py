"""The reference's whole fit.py flow on the bench scene, on the device: initialize_convex_hull (fit.py:315) -> n_iter
MALA* iterations with z-score resets every `reset_epochs` (fit.py:399-458) -> export_poses (.dexgrasp.pt, fit.py:224-300),
then three independent checks of what came out:

  * the final energies re-evaluated through the class surface (HandModel / ObjectModel / calculate_energy: the autograd
    route, separate launches) must equal the stepper's accepted energies;
  * a sample of rows re-evaluated by the CPU oracle (fp64);
  * the exported files reloaded with torch.load(weights_only=True) and assembled the way the reference's consumer does
    (graspqp_isaaclab/.../utils/data.py:105-140).

Evidence run, not a test: writes one JSON record (default gpurun_out/fit_end_to_end.json).

usage: python tools/fit_end_to_end.py [--n_objects 1] [--batch_size 256] [--n_iter 7000] [--out file.json]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--hand", default="allegro")
ap.add_argument("--n_objects", type=int, default=1)
ap.add_argument("--batch_size", type=int, default=256)
ap.add_argument("--n_contact", type=int, default=12)
ap.add_argument("--n_iter", type=int, default=7000)
ap.add_argument("--reset_epochs", type=int, default=600)
ap.add_argument("--oracle_rows", type=int, default=6)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fit_end_to_end.json"))
args = ap.parse_args()

from graspqp_amd import ops
from graspqp_amd.core.energy import calculate_energy
from graspqp_amd.core.hand_model import HandModel
from graspqp_amd.core.object_model import ObjectModel
from graspqp_amd.export import export_poses
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.metrics import GraspSpanMetricFactory
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes

W = {"E_dis": 100.0, "E_fc": 1.0, "E_pen": 100.0, "E_spen": 10.0, "E_joints": 1.0}  # fit.py:51-55
spec = get_hand_spec(args.hand)
n_obj, be, n = args.n_objects, args.batch_size, args.n_contact
B = n_obj * be
codes = [f"superquadric_{o}" for o in range(n_obj)]
fvs = [meshes.superquadric(o) for o in range(n_obj)]
sps = [meshes.surface_points(f, 2500, oversample=4, seed=42) for f in fvs]
om = ObjectModel(batch_size_each=be, num_samples=2500)
om.initialize_from_meshes(fvs, codes, surface_points_list=sps)
hand = ops.HandHandle(spec)
st = GraspStepper(hand, ops.MeshSet(fvs), torch.tensor(np.stack(sps)), be, n, seed=1)
st.set_hulls(om.convex_hulls())
st.initialize()  # on-device initialize_convex_hull + the first evaluation
e0, t0_terms = st.energy.clone(), st.terms.clone()
st.capture(iters=8)
trace = []


def cb(step):
    if step % 500 == 0:
        e = st.energy
        trace.append({"step": int(step), "mean": float(e.mean()), "min": float(e.min()), "max": float(e.max()),
                      "accept": float(st.accept.float().mean())})


torch.cuda.synchronize()
t0 = time.time()
st.run(args.n_iter, reset_epochs=args.reset_epochs, z_score_threshold=1.0, callback=cb)
torch.cuda.synchronize()
dt = time.time() - t0
assert torch.isfinite(st.energy).all() and torch.isfinite(st.hand_pose).all()

# ---- (1) class surface (autograd route, separate launches) on the final state -------------------------------------------
hm = HandModel(spec, "cuda")
hp = st.hand_pose.clone().requires_grad_()
hm.set_parameters(hp, st.contact_idx.clone())
metric = GraspSpanMetricFactory.create(GraspSpanMetricFactory.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
losses = calculate_energy(hm, om, energy_fnc=metric, method="gendexgrasp", svd_gain=0.1)
total_cls = sum(W[k] * losses[k] for k in W)
rel_cls = ((total_cls.detach() - st.energy).abs() / st.energy.abs().clamp_min(1e-6))

# ---- (2) the CPU oracle (fp64) on a few rows ------------------------------------------------------------------------------
import ref_cpu
from ref_cpu import models as omodels

rows = torch.linspace(0, be - 1, args.oracle_rows).long().tolist()
oh = omodels.OracleHand(spec, torch.float64)
oo = omodels.OracleObject([fvs[0]], [sps[0]], len(rows), torch.float64)
oh.set_parameters(st.hand_pose[rows].double().cpu(), st.contact_idx[rows].cpu())
lo = ref_cpu.calculate_energy(oh, oo, box_form=True)
# every term but E_fc row by row (E_fc's stop rule is batch-global: it is compared on whole batches in tests/)
rel_oracle = {}
names = ["E_dis", "E_fc", "E_pen", "E_spen", "E_joints"]
for k in ("E_dis", "E_pen", "E_spen", "E_joints"):
    got = st.terms[names.index(k)][rows].double().cpu()
    rel_oracle[k] = float(((got - lo[k]).abs() / lo[k].abs().clamp_min(1e-4)).max())

# ---- (3) export + reload the way the consumer does ----------------------------------------------------------------------
with tempfile.TemporaryDirectory() as tmp:
    hm.set_parameters(st.hand_pose.clone(), st.contact_idx.clone())
    files = export_poses(hm, st.energy.clone(), om, codes, be, tmp, args.hand, n, "graspqp", suffix="")  # fit.py:521
    data = torch.load(files[0], weights_only=True)
    jn = list(spec.joint_names)
    params = torch.cat([data["parameters"]["root_pose"], torch.stack([data["parameters"][k] for k in jn], -1)], -1)
    vel = torch.stack([data["grasp_velocities_off"][k] + 0.1 * data["grasp_velocities"][k] for k in jn], -1)
    export = {"files": [os.path.relpath(f, tmp) for f in files], "keys": sorted(data), "params_shape": list(params.shape),
              "velocities_finite": bool(torch.isfinite(vel).all()), "quaternion_norm_max_dev": float((params[:, 3:7].norm(dim=-1) - 1).abs().max()),
              "bytes": os.path.getsize(files[0])}

best = torch.topk(-st.energy, min(5, B)).indices
rec = {
    "what": "reference fit.py flow on the device: initialize_convex_hull -> MALA* schedule with z-score resets -> export_poses",
    "hand": args.hand, "n_objects": n_obj, "batch_size": be, "n_contact": n, "n_iter": args.n_iter, "reset_epochs": args.reset_epochs,
    "graph_mode": st.graph_mode, "wall_s": dt, "evals_per_s_including_host_callbacks_and_resets": B * args.n_iter / dt,
    "energy_mean_initial": float(e0.mean()), "energy_mean_final": float(st.energy.mean()), "energy_min_final": float(st.energy.min()),
    "terms_mean_initial": {k: float(t0_terms[i].mean()) for i, k in enumerate(names)},
    "terms_mean_final": {k: float(st.terms[i].mean()) for i, k in enumerate(names)},
    "best_rows": [{"row": int(i), "E": float(st.energy[i]), **{k: float(st.terms[j][i]) for j, k in enumerate(names)}} for i in best],
    "trace": trace,
    "check_class_surface_rel_err": {"median": float(rel_cls.median()), "max": float(rel_cls.max())},
    "check_oracle_fp64_rel_err_max_over_rows": rel_oracle, "oracle_rows": rows,
    "export": export,
}
assert rec["energy_mean_final"] < rec["energy_mean_initial"]
assert rec["check_class_surface_rel_err"]["median"] < 1e-4
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
json.dump(rec, open(args.out, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("wall_s", "energy_mean_initial", "energy_mean_final", "energy_min_final",
                                      "check_class_surface_rel_err", "check_oracle_fp64_rel_err_max_over_rows", "export")}))
