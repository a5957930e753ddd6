"""How much of the stand-alone hand-penetration query's time is cache state?  The same launch (same inputs) timed by its
in-kernel span (first block start -> last block end, 100 MHz) (a) right after itself, (b) after one whole eager iteration
of the other kernels, (c) after 256 MB of unrelated traffic.  Development aid."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from graspqp_amd import ops, _C
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes
from bench import make_initial_state

n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 1
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 600
spec = get_hand_spec("allegro")
fvs = [meshes.superquadric(o) for o in range(n_obj)]
sps = [meshes.surface_points(f, 2500, oversample=4, seed=42) for f in fvs]
hand = ops.HandHandle(spec)
st = GraspStepper(hand, ops.MeshSet(fvs), torch.tensor(np.stack(sps)), 256, 12, seed=1)
hps, idxs = zip(*[make_initial_state(spec, f, 256, 12, 1000 + o) for o, f in enumerate(fvs)])
st.reset(torch.cat(hps).cuda(), torch.cat(idxs).cuda())
for _ in range(warm):
    st.step()
torch.cuda.synchronize()
junk = torch.empty(64 * 1024 * 1024, device="cuda")
pose, idx = st.hand_pose.clone(), st.contact_idx.clone()


def pen(ppt=1):
    _C.call("gq_debug_set_pen_ppt", ppt)
    span = torch.zeros(64, 2, dtype=torch.int64, device="cuda")
    span[:, 0] = torch.iinfo(torch.int64).max
    _C.call("gq_hand_pen_forward", hand.links.handle, _C.f32(st.surf), st.n_obj, st.P, st.be, _C.f32(pose), st.D,
            _C.f32(st.Rg), _C.f32(st.link_T), 1, _C.f32(st.pen_dis), _C.i32(st.pen_link), _C.f32(st.pen_gvec), None, 0, None,
            ctypes.c_void_p(span.data_ptr()), _C.f32(st.patch), _C.stream_ptr())
    torch.cuda.synchronize()
    _C.call("gq_debug_set_pen_ppt", 0)
    s = span.cpu().numpy()
    return (s[:, 1].max() - s[:, 0].min()) / 100.0


st._eval_fk(pose, idx, _C.stream_ptr())
torch.cuda.synchronize()
for ppt in (1, 2):
    a = [pen(ppt) for _ in range(6)][1:]
    b = []
    for _ in range(5):
        st._iteration(_C.stream_ptr())  # one whole eager iteration of every kernel (state may move: re-evaluate FK at the frozen pose)
        st._eval_fk(pose, idx, _C.stream_ptr())
        b.append(pen(ppt))
    c = []
    for _ in range(5):
        junk.fill_(1.0)
        c.append(pen(ppt))
    d = []
    for _ in range(5):
        st._eval_fk(pose, idx, _C.stream_ptr())
        d.append(pen(ppt))
    f = lambda v: f"{np.median(v):5.1f} (min {min(v):5.1f} max {max(v):5.1f})"
    print(f"{n_obj} x 256 rows after {warm} iterations, {ppt} point(s) per thread, span us: back to back {f(a)} | after an eager iteration "
          f"{f(b)} | after 256 MB of other traffic {f(c)} | after the FK forward launch only {f(d)}", flush=True)
