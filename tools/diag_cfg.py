"""Step-by-step run of one non-default configuration with a synchronisation after every phase (development aid)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from graspqp_amd import ops, _C
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes

hand_name, n, k, n_obj, be = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
def say(*a):
    print(*a, flush=True)
spec = get_hand_spec(hand_name)
fvs = [meshes.superquadric(5 + i, n_u=32, n_v=16) for i in range(n_obj)]
sps = [meshes.surface_points(f, 600, oversample=4, seed=3 + i) for i, f in enumerate(fvs)]
B = n_obj * be
g = torch.Generator().manual_seed(31)
t = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 0.1
hp = torch.cat([t, torch.randn(B, 6, generator=g), torch.tensor(spec.default_state)[None].float() + 0.3 * torch.randn(B, spec.n_dofs, generator=g)], 1).cuda()
idx = torch.randint(spec.n_contact_candidates, (B, n), generator=g).cuda()
say("hand handle"); hand = ops.HandHandle(spec); torch.cuda.synchronize()
say("meshset"); ms = ops.MeshSet(fvs); torch.cuda.synchronize()
say("stepper"); st = GraspStepper(hand, ms, torch.tensor(np.stack(sps)), be, n, fc_cfg={"n_cone_vecs": k}, seed=5); torch.cuda.synchronize()
s = _C.stream_ptr()
st.pose_new.copy_(hp); st.idx_new.copy_(idx)
say("fk"); st._eval_fk(st.pose_new, st.idx_new, s); torch.cuda.synchronize()
say("contacts"); st._eval_contacts(s); torch.cuda.synchronize()
say("pen"); st._eval_pen(st.pose_new, s); torch.cuda.synchronize()
say("tail"); st._eval_tail(st.pose_new, st.idx_new, s); torch.cuda.synchronize()
say("reset"); st.reset(hp, idx); torch.cuda.synchronize()
say("eager step"); st.step(); torch.cuda.synchronize()
say("fused eager"); st.draw(); st._iteration(s, fused=True); torch.cuda.synchronize()
say("capture"); st.capture(iters=2); torch.cuda.synchronize()
for i in range(4):
    st.step()
st.flush(); torch.cuda.synchronize()
say("done", float(st.energy.mean()))
