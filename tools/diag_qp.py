"""Diagnostic (GPU): E_fc / QP stop iteration of the GPU path vs the fp64 oracle on the MALA fixture proposals."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import ref_cpu
from ref_cpu import models as omodels, span as ospan, qp as oqp
from graspqp_amd import ops
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper

g = np.load(os.path.join(ROOT, "tests/golden/mala_allegro_sphere_b8_n4.npz"))
spec = get_hand_spec("allegro")
n_obj, be = int(g["n_obj"]), int(g["batch_size_each"])
fvs = [g[f"obj{i}_face_verts"] for i in range(n_obj)]
sps = [g[f"obj{i}_surface_points"] for i in range(n_obj)]
hand = ops.HandHandle(spec); ms = ops.MeshSet(fvs)
st = GraspStepper(hand, ms, torch.tensor(np.stack(sps)), be, 4)
for s in range(1, 6):
    hp = torch.tensor(g[f"s{s}_prop_pose"]); 
    # proposal indices: idx where accepted else? use recorded new idx logic: reconstruct from fixture
    prev_idx = torch.tensor(g["contact_idx0"] if s == 1 else g[f"s{s-1}_contact_idx"])
    mask = torch.tensor(g[f"s{s}_u_switch"]) < 0.4
    idx = torch.where(mask, torch.tensor(g[f"s{s}_new_idx"]), prev_idx)
    terms, total, grad = st.evaluate(hp.float().cuda(), idx.cuda())
    nit_gpu = int(st.n_iter.item())
    oh = omodels.OracleHand(spec, torch.float64); oo = omodels.OracleObject(fvs, sps, be, torch.float64)
    oh.set_parameters(hp.double(), idx)
    dist, cn = oo.cal_distance(oh.contact_points)
    F = ospan.grasp_matrix(oh.contact_points, cn, oo.cog)
    B, _, nz = F.shape
    Q = F.transpose(1, 2) @ F + 1e-4 * torch.eye(nz, dtype=torch.float64)
    for dt in (torch.float64, torch.float32):
        hist = []
        x, lam, sl, nit = oqp.pdipm_forward_box(Q.to(dt), torch.zeros(B, nz, dtype=dt), torch.ones(B, nz, dtype=dt), 21 * torch.ones(B, nz, dtype=dt), history=hist)
        val = 0.5 * ((F.to(dt) @ x.unsqueeze(-1)).squeeze(-1) ** 2).sum(-1)
        res = torch.stack([h["resids"] for h in hist], 1)
        print(f"step {s} {str(dt):14s} n_iter {nit} val {val.double().numpy().round(4)}")
        if dt == torch.float64:
            print("   resid table (rows x iters):\n", res.numpy().round(4))
    # GPU val from fc workspace
    import ctypes
    outs = ops.fc_peek(st.fc_ws, st.B, 4, 4)
    valp = (ctypes.c_float * st.B).from_address(0)  # placeholder
    e_fc = terms["E_fc"].cpu().double().numpy()
    print(f"step {s} GPU            n_iter {nit_gpu} E_fc {e_fc.round(4)} total {total.cpu().numpy().round(3)}")
    print(f"        fixture new_energy {g[f's{s}_new_energy'].round(3)}")
