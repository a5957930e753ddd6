"""GPU micro-benchmark / diagnostics of the hand-penetration query on a frozen chain state (development aid)."""
import ctypes, os, sys
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
hand = ops.HandHandle(spec)
st = GraspStepper(hand, ops.MeshSet([fv]), torch.tensor(sp)[None], 256, 12, seed=1)
hp, idx = make_initial_state(spec, fv, 256, 12, 1000)
st.reset(hp.cuda(), idx.cuda())
done = 0
for warm in [int(a) for a in sys.argv[1:]] or [20, 200, 1000]:
    while done < warm:
        st.step(); done += 1
    torch.cuda.synchronize()
    s = _C.stream_ptr()
    s_side = torch.cuda.Stream()
    st._eval_fk(st.hand_pose, st.contact_idx, s)
    def fwd(mode, ws):
        _C.call("gq_hand_pen_forward", hand.links.handle, _C.f32(st.surf), st.n_obj, st.P, st.be, _C.f32(st.hand_pose), st.D,
                _C.f32(st.Rg), _C.f32(st.link_T), mode, _C.f32(st.pen_dis), _C.i32(st.pen_link), _C.f32(st.pen_gvec),
                _C.ptr(ws_t) if ws else None, nb if ws else 0, None, None, None, _C.stream_ptr())
    res = {}
    nb = ops._size_call("gq_hand_pen_workspace_bytes", ctypes.c_int64(st.B), ctypes.c_int64(st.P), st.L)
    ws_t = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    ref = None
    for name, mode, ws in (("grid", 1, False), ("queue", 3, True), ("scanonly", 9, False)):
        for _ in range(3): fwd(mode, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g = torch.cuda.CUDAGraph()   # replay from a graph: the ctypes call costs more host time than the kernel
        with torch.cuda.graph(g):
            for _ in range(20): fwd(mode, ws)
        g.replay(); torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
        if name == "scanonly": continue
        if ref is None: ref = (st.pen_dis.clone(), st.pen_link.clone(), st.pen_gvec.clone())
        else:
            pos = ref[0] > 1e-6
            print(f"   {name}: max |dis - grid| on positives {(st.pen_dis[pos] - ref[0][pos]).abs().max().item():.2e}, "
                  f"positives {int(pos.sum())} vs {int((st.pen_dis > 1e-6).sum())}")
    cnt = torch.zeros(12, dtype=torch.int64, device="cuda")
    _C.call("gq_debug_set_pen_counters", ctypes.c_void_p(cnt.data_ptr()))
    fwd(3, False); torch.cuda.synchronize()
    _C.call("gq_debug_set_pen_counters", None)
    npos = int((st.pen_dis > 0).sum())
    print(f"after {done:5d} steps: grid {res['grid']:.1f} us | queue {res['queue']:.1f} us | scan only {res['scanonly']:.1f} us | "
          f"(point,link) pairs in occupied voxels {int(cnt[0])} of {256*2500*14}, penetrating points {npos}, mean E {st.energy.mean().item():.2f}")
