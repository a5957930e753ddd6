"""When do the blocks of the stand-alone hand-penetration query start and end?  Needs a development build of the library
with -DGQ_BLOCK_TIMES (graspqp_amd/lib/libgraspqp_hip_A.so, see tools/block_timeline.sh) through GRASPQP_HIP_LIB.
Prints start-offset and duration percentiles and the number of blocks in flight over time (100 MHz clock)."""
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
spec = get_hand_spec("allegro")
fvs = [meshes.superquadric(o) for o in range(n_obj)]
sps = [meshes.surface_points(f, 2500, oversample=4, seed=42) for f in fvs]
hand = ops.HandHandle(spec)
st = GraspStepper(hand, ops.MeshSet(fvs), torch.tensor(np.stack(sps)), 256, 12, seed=1)
# the instrumented library writes one (start, end) pair per block BEHIND the 64 span pairs: the stepper's own span buffer
# has to be that large too, or its launches write out of bounds
assert "libgraspqp_hip_A" in os.environ.get("GRASPQP_HIP_LIB", ""), "run with the -DGQ_BLOCK_TIMES build (tools/block_timeline.sh)"
st._span = torch.zeros(64 + 4 * (st.B * 10 + 16), 2, dtype=torch.int64, device="cuda")
st._span[:64, 0] = -1
st._pen_desc.span = st._span.data_ptr()
hps, idxs = zip(*[make_initial_state(spec, f, 256, 12, 1000 + o) for o, f in enumerate(fvs)])
st.reset(torch.cat(hps).cuda(), torch.cat(idxs).cuda())
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 600):
    st.step()
torch.cuda.synchronize()
st._eval_fk(st.hand_pose, st.contact_idx, _C.stream_ptr())
for ppt in (1, 2):
    _C.call("gq_debug_set_pen_ppt", ppt)
    nblk = st.B * ((10 + ppt - 1) // ppt)
    for rep in range(3):
        span = torch.zeros(128 + 8 * nblk, dtype=torch.int64, device="cuda")
        span[0:128:2] = torch.iinfo(torch.int64).max
        _C.call("gq_hand_pen_forward", hand.links.handle, _C.f32(st.surf), st.n_obj, st.P, st.be, _C.f32(st.hand_pose), st.D,
                _C.f32(st.Rg), _C.f32(st.link_T), 1, _C.f32(st.pen_dis), _C.i32(st.pen_link), _C.f32(st.pen_gvec), None, 0, None,
                ctypes.c_void_p(span.data_ptr()), _C.f32(st.patch), _C.stream_ptr())
        torch.cuda.synchronize()
    t = span[128:].view(-1, 8).cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    start, dur = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0  # us
    cnt = span[128:].view(-1, 8)[:, 5].cpu().numpy()
    cnt = cnt[: len(t)] if len(cnt) >= len(t) else cnt
    n_ent, n_item = (cnt & 0xffffffff), (cnt >> 32)
    pc = lambda a: " ".join(f"{np.percentile(a, q):6.1f}" for q in (0, 10, 50, 90, 99, 100))
    print(f"{n_obj} x 256 rows, {ppt} point(s) per thread: {len(t)} blocks recorded of {nblk}, kernel span {(t[:, 1].max() - t0) / 100.0:.1f} us")
    print(f"   block start offset us (min p10 p50 p90 p99 max): {pc(start)}")
    print(f"   block duration     us (min p10 p50 p90 p99 max): {pc(dur)}")
    full = t[:, 2] > 0
    if full.any():
        ph = lambda a, b: (t[full, a] - t[full, b]) / 100.0
        print(f"   blocks that reach the list phases: {int(full.sum())}; prologue+scan {pc(ph(2, 0))} | ranking {pc(ph(3, 2))} | finish {pc(ph(4, 3))} | output {pc(ph(1, 4))}")
    order = np.argsort(-dur)[:8]
    print("   longest blocks (duration us, start us, entries, items):", [(round(float(dur[i]), 1), round(float(start[i]), 1), int(n_ent[i]) if i < len(n_ent) else -1, int(n_item[i]) if i < len(n_item) else -1) for i in order])
    print(f"   entries per block (min p10 p50 p90 p99 max): {pc(n_ent)}   items: {pc(n_item)}")
    grid = np.arange(0, (t[:, 1].max() - t0) / 100.0, 2.0)
    infl = [(int(((start <= g) & (start + dur > g)).sum())) for g in grid]
    print("   blocks in flight every 2 us:", infl)
