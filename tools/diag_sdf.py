import os, sys, ctypes
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT)
import torch
from graspqp_amd import ops, _C
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes
from bench import make_initial_state
spec = get_hand_spec("allegro")
fv = meshes.superquadric(0); sp = meshes.surface_points(fv, 2500, oversample=4, seed=42)
hand = ops.HandHandle(spec)
st = GraspStepper(hand, ops.MeshSet([fv]), torch.tensor(sp)[None], 256, 12, seed=1)
hp, idx = make_initial_state(spec, fv, 256, 12, 1000)
st.reset(hp.cuda(), idx.cuda())
for _ in range(100): st.step()
N = 256 * 12
cnt = torch.zeros(16 + N * 8, dtype=torch.int64, device="cuda")
_C.call("gq_debug_set_pen_counters", ctypes.c_void_p(cnt.data_ptr()))
_C.call("gq_sdf_forward_meshset", st.objs.handle, _C.f32(st.cpts), N, N, _C.f32(st.d2), _C.i32(st.sgn), _C.f32(st.onrm), _C.f32(st.closest), _C.stream_ptr())
torch.cuda.synchronize()
_C.call("gq_debug_set_pen_counters", None)
d = cnt[16:].view(N, 8).cpu().double()
for i, n in enumerate(["visits", "rounds", "select cycles", "load+rank cycles", "search total cycles"]):
    print(f"{n:22s} mean {d[:, i].mean().item():9.1f}  p90 {d[:, i].quantile(0.9).item():9.1f}  max {d[:, i].max().item():9.1f}")
print("per round: select %.0f, load+rank %.0f cycles" % ((d[:, 2].sum() / d[:, 1].sum()).item(), (d[:, 3].sum() / d[:, 1].sum()).item()))
