"""Penetration-only E_pen query (occupancy grid shortcut) against the full query (TorchSDF sign rule at every point) for
every hand on deep-penetration scenes: how many points and how much E_pen the shortcut changes.  Development aid."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch

from graspqp_amd.core.hand_model import HandModel
from graspqp_amd.hands import AVAILABLE_HANDS, get_hand_spec
from graspqp_amd.utils import meshes

be, P = int(os.environ.get("ROWS", 48)), 2500
fvs = [meshes.superquadric(3), meshes.superquadric(4)]
surf = torch.tensor(np.stack([meshes.surface_points(f, P, oversample=4, seed=42) for f in fvs])).cuda()
for h in AVAILABLE_HANDS:
    spec = get_hand_spec(h)
    B = 2 * be
    g = torch.Generator().manual_seed(7)
    t = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * (0.02 + 0.08 * torch.rand(B, 1, generator=g))
    lo, hi = torch.tensor(spec.joints_lower), torch.tensor(spec.joints_upper)
    th = lo + (hi - lo) * torch.rand(B, spec.n_dofs, generator=g)  # anywhere inside the limits: folded fingers included
    hp = torch.cat([t, torch.randn(B, 6, generator=g), th], 1).cuda()
    idx = torch.randint(spec.n_contact_candidates, (B, 4), generator=g).cuda()
    hm = HandModel(spec, "cuda")
    hm.set_parameters(hp, idx)
    d0 = torch.relu(hm.cal_distance(surf, penetration_only=0))
    d1 = torch.relu(hm.cal_distance(surf, penetration_only=1))
    torch.cuda.synchronize()
    diff = (d0 - d1).abs()
    bad = diff > 3e-6
    e0, e1 = d0.sum(-1), d1.sum(-1)
    rel = ((e0 - e1).abs() / e0.clamp_min(1e-6))
    print(f"{h:13s} penetrating points {int((d0 > 0).sum()):7d}  differing {int(bad.sum()):5d} ({float(bad.sum()) / max(1, int((d0 > 0).sum())):.2e})  "
          f"shortcut low {int(((d0 - d1) > 3e-6).sum())} high {int(((d1 - d0) > 3e-6).sum())}  max |diff| {float(diff.max()):.2e}  "
          f"E_pen rel diff per row: median {float(rel.median()):.1e} max {float(rel.max()):.1e}  rows > 1e-3: {int((rel > 1e-3).sum())}/{B}", flush=True)
    if bad.any() and os.environ.get("DETAIL", "1") == "1":
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from ref_cpu import models as omodels, sdf as osdf

        o0 = torch.ops.graspqp_amd.hand_pen(hp, surf, be, hm._hand.hid, hm.global_rotation.detach().contiguous(), hm.current_status.detach().contiguous(), 0)
        o1 = torch.ops.graspqp_amd.hand_pen(hp, surf, be, hm._hand.hid, hm.global_rotation.detach().contiguous(), hm.current_status.detach().contiguous(), 1)
        oh = omodels.OracleHand(spec, torch.float64)
        oh.set_parameters(hp.double().cpu(), idx.cpu())
        for r, j in torch.nonzero(bad).tolist()[:8]:
            x = surf[r // be, j].double().cpu()[None, None]
            xh = (x - oh.global_translation[r:r + 1].unsqueeze(1)) @ oh.global_rotation[r:r + 1]
            per = []
            for l, fv in enumerate(oh.link_faces):
                T = oh.current_status[r:r + 1, l]
                xl = ((xh - T[:, :3, 3].unsqueeze(1)) @ T[:, :3, :3]).reshape(-1, 3)
                d2, sgn, _, _ = osdf.compute_sdf(xl, fv)
                lo_, hi_ = fv.reshape(-1, 3).min(0)[0], fv.reshape(-1, 3).max(0)[0]
                per.append((float(torch.sqrt(d2 + 1e-8) * (-sgn)), bool(((xl[0] >= lo_) & (xl[0] <= hi_)).all())))
            lo_best = max(range(len(per)), key=lambda k: per[k][0])
            print(f"   row {r} point {j}: full {float(o0[0][r, j]):+.5f} (link {int(o0[1][r, j])})  shortcut {float(o1[0][r, j]):+.5f} (link {int(o1[1][r, j])})  "
                  f"oracle {per[lo_best][0]:+.5f} on link {lo_best} (point in that link's AABB: {per[lo_best][1]}); "
                  f"oracle-positive links {[(k, round(v, 5), inb) for k, (v, inb) in enumerate(per) if v > 0]}", flush=True)
