"""Which surface points of one row differ between the penetration-only E_pen query (what the stepper runs), the full
query (penetration_only=0, TorchSDF sign rule for every link) and the fp64 oracle.  Development aid.
usage: python tools/diag_pen_row.py  (scene of tests/test_gpu_alt_energies.py[shadow_hand-pinch], row 5)"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

from graspqp_amd import ops, stepper
from graspqp_amd.core.hand_model import HandModel
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.utils import meshes
from ref_cpu import models as omodels, sdf

spec = get_hand_spec("shadow_hand", grasp_type="pinch")
n_obj, be, n = 2, 5, 12
B = n_obj * be
fvs = [meshes.superquadric(9 + i, 32, 16) for i in range(n_obj)]
sps = [meshes.surface_points(f, 600, oversample=4, seed=3 + i) for i, f in enumerate(fvs)]
g0 = torch.Generator().manual_seed(17)
t = torch.nn.functional.normalize(torch.randn(B, 3, generator=g0, dtype=torch.float64), dim=-1) * 0.09
th = torch.tensor(spec.default_state, dtype=torch.float64)[None] + 0.2 * torch.randn(B, spec.n_dofs, generator=g0, dtype=torch.float64)
th[0] = torch.tensor(spec.joints_upper, dtype=torch.float64) + 0.05
hp = torch.cat([t, torch.randn(B, 6, generator=g0, dtype=torch.float64), th], 1)
idx = torch.randint(spec.n_contact_candidates, (B, n), generator=g0)
r = int(os.environ.get("ROW", 5))

hand = ops.HandHandle(spec)
st = stepper.GraspStepper(hand, ops.MeshSet(fvs), torch.tensor(np.stack(sps)), be, n, seed=5)
terms, total, grad = st.evaluate(hp.float().cuda(), idx.cuda())
torch.cuda.synchronize()
d1 = st.pen_dis[r].cpu().double()
l1 = st.pen_link[r].cpu()
hm = HandModel(spec, "cuda")
hm.set_parameters(hp.float().cuda(), idx.cuda())
surf = torch.tensor(np.stack(sps)).cuda()
d0 = hm.cal_distance(surf, penetration_only=False)[r].cpu().double()
dp = hm.cal_distance(surf, penetration_only=True)[r].cpu().double()

oh = omodels.OracleHand(spec, torch.float64)
oh.set_parameters(hp, idx)
x = torch.as_tensor(sps[r // be], dtype=torch.float64)[None]
xh = (x - oh.global_translation[r:r + 1].unsqueeze(1)) @ oh.global_rotation[r:r + 1]
D, XL = [], []
for l, fv in enumerate(oh.link_faces):
    T = oh.current_status[r:r + 1, l]
    xl = ((xh - T[:, :3, 3].unsqueeze(1)) @ T[:, :3, :3]).reshape(-1, 3)
    d2, sgn, _, _ = sdf.compute_sdf(xl, fv)
    D.append(torch.sqrt(d2 + 1e-8) * (-sgn))
    XL.append(xl)
D = torch.stack(D)
mx, arg = D.max(0)
print("E_pen: stepper", float(terms["E_pen"][r]), " relu-sum pen-only", float(d1.clamp_min(0).sum()), " class pen-only", float(dp.clamp_min(0).sum()),
      " class full", float(d0.clamp_min(0).sum()), " oracle", float(mx.clamp_min(0).sum()))
for name, d in (("pen-only(stepper)", d1), ("pen-only(class)", dp), ("full(class)", d0)):
    bad = torch.nonzero((d.clamp_min(0) - mx.clamp_min(0)).abs() > 1e-5).flatten().tolist()
    print(name, "points that differ from the oracle:", bad)
    for j in bad:
        l = int(arg[j])
        fv = oh.link_faces[l]
        lo, hi = fv.reshape(-1, 3).min(0)[0], fv.reshape(-1, 3).max(0)[0]
        xl = XL[l][j]
        vox = ((xl - lo) / (hi - lo) * 32).tolist()
        pos_links = [(spec.link_names[int(k)], round(float(D[k, j]), 6)) for k in torch.nonzero(D[:, j] > 0).flatten()]
        print(f"   point {j}: got {float(d[j]):.6f} (link {int(l1[j]) if name.startswith('pen-only(st') else '-'}) oracle {float(mx[j]):.6f} on "
              f"{spec.link_names[l]}; oracle-positive links {pos_links}; voxel coords in that link {['%.2f' % v for v in vox]}, "
              f"voxel size mm {[round(float(s) * 1e3 / 32, 2) for s in (hi - lo)]}")
