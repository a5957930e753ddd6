"""Which quantity differs between the fused (graph) and the unfused (eager) iteration? (development aid)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from graspqp_amd import ops, _C
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes
from bench import make_initial_state

spec = get_hand_spec("allegro")
fv = meshes.superquadric(0); sp = meshes.surface_points(fv, 2500, oversample=4, seed=42)
hand = ops.HandHandle(spec); ms = ops.MeshSet([fv])
hp, idx = make_initial_state(spec, fv, 256, 12, 1000)
res = []
for fused in (False, True):
    st = GraspStepper(hand, ms, torch.tensor(sp)[None], 256, 12, seed=1)
    st.reset(hp.cuda(), idx.cuda())
    st.draw()
    st._iteration(_C.stream_ptr(), fused=fused)
    torch.cuda.synchronize()
    res.append({k: getattr(st, k).clone() for k in ("terms_new", "total_new", "grad_new", "spheres", "g_sph_w", "d2", "cpts",
                                                       "pen_dis", "wrench", "g_cpts", "energy", "hand_pose")})
for k in res[0]:
    a, b = res[0][k], res[1][k]
    print(k, "equal" if torch.equal(a, b) else f"max abs diff {(a - b).abs().max().item():.3e}")
t0, t1 = res[0]["terms_new"], res[1]["terms_new"]
for i, n in enumerate(("E_dis", "E_fc", "E_pen", "E_spen", "E_joints")):
    print(n, (t0[i] - t1[i]).abs().max().item())
