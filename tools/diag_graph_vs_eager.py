"""One MALA* iteration from the same state, eager unfused launches vs the captured graph: which buffers differ (development aid)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from bench import make_initial_state
from graspqp_amd import ops
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes

spec = get_hand_spec("allegro")
fv = meshes.superquadric(0)
sp = meshes.surface_points(fv, 2500, oversample=4, seed=42)
hand = ops.HandHandle(spec)
hp, idx = make_initial_state(spec, fv, 256, 12, 7)
outs = []
for mode in ("eager", "graph"):
    st = GraspStepper(hand, ops.MeshSet([fv]), torch.tensor(sp[None]), 256, 12, seed=5)
    st.reset(hp.cuda(), idx.cuda())
    if mode == "graph":
        st.capture()
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
        st.step()
    st.flush()
    torch.cuda.synchronize()
    from graspqp_amd import _C
    saved = st.cpts.clone()
    st._eval_fk(st.pose_new, st.idx_new, _C.stream_ptr(), loop=False)  # plain FK of the proposal: ground truth of cpts
    torch.cuda.synchronize()
    print(mode, "cpts of the iteration == plain FK of (pose_new, idx_new):", torch.equal(saved, st.cpts),
          "differing rows", torch.nonzero((saved != st.cpts).reshape(256, -1).any(1)).flatten()[:8].tolist())
    st.cpts.copy_(saved)
    outs.append({k: getattr(st, k).clone() for k in ("pose_new", "idx_new", "cpts", "cnrm", "Rg", "link_T", "d2", "g_cpts", "grad_new", "total_new",
                                                     "energy", "hand_pose", "contact_idx", "ema", "g2", "accept")})
for k in outs[0]:
    a, b = outs[0][k], outs[1][k]
    same = torch.equal(a, b)
    print(f"{k:12s} equal={same}" + ("" if same else f"  differing elements {(a != b).sum().item()} / {a.numel()}  rows {torch.nonzero((a != b).reshape(a.shape[0], -1).any(1)).flatten()[:8].tolist()}"))
