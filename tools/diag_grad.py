"""Diagnostic (GPU): per-term gradient error of the fused stepper vs the fp64 / fp32 oracle on a golden state."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import ref_cpu
from ref_cpu import models as omodels
from graspqp_amd import ops
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper

g = np.load(os.path.join(ROOT, "tests/golden/mala_allegro_sphere_b8_n4.npz"))
spec = get_hand_spec("allegro")
n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
fvs = [g[f"obj{i}_face_verts"] for i in range(n_obj)]
sps = [g[f"obj{i}_surface_points"] for i in range(n_obj)]
hand = ops.HandHandle(spec)
ms = ops.MeshSet(fvs)
for key_pose, key_idx in (("s1_hand_pose", "s1_contact_idx"), ("s2_prop_pose", None), ("s3_hand_pose", "s3_contact_idx")):
    hp = torch.tensor(g[key_pose]); idx = torch.tensor(g[key_idx] if key_idx else g["s2_contact_idx"])
    print("==== state", key_pose)
    for term in ("E_dis", "E_fc", "E_pen", "E_spen", "E_joints", "all"):
        w = {k: 0.0 for k in ("E_dis", "E_fc", "E_pen", "E_spen", "E_joints")}
        if term == "all":
            w = dict(ref_cpu.energy.DEFAULT_WEIGHTS)
        else:
            w[term] = ref_cpu.energy.DEFAULT_WEIGHTS[term]
        st = GraspStepper(hand, ms, torch.tensor(np.stack(sps)), be, 4, weights=w)
        terms, total, grad = st.evaluate(hp.cuda(), idx.cuda())
        res = {}
        for dt in (torch.float64, torch.float32):
            oh = omodels.OracleHand(spec, dt); oo = omodels.OracleObject(fvs, sps, be, dt)
            hpo = hp.to(dt).requires_grad_(); oh.set_parameters(hpo, idx)
            lo = ref_cpu.calculate_energy(oh, oo, box_form=True)
            tot = ref_cpu.total_energy(lo, w); 
            if tot.requires_grad:
                tot.sum().backward(); res[dt] = (tot.detach().double(), oh.hand_pose.grad.double())
            else:
                res[dt] = (tot.detach().double(), torch.zeros_like(hpo).double())
        t64, g64 = res[torch.float64]; t32, g32 = res[torch.float32]
        gg = grad.cpu().double()
        den = g64.norm().item() + 1e-30
        print(f"{term:9s} |g64|={den:9.3e}  gpu-vs-64 {((gg-g64).norm()/den):.2e}  cpu32-vs-64 {((g32-g64).norm()/den):.2e}  "
              f"E rel gpu {((total.cpu().double()-t64).abs()/(t64.abs()+1e-9)).max():.2e} cpu32 {((t32-t64).abs()/(t64.abs()+1e-9)).max():.2e}  "
              f"maxabs comp gpu {(gg-g64).abs().max():.2e} cpu32 {(g32-g64).abs().max():.2e}")
