"""GPU micro-benchmark of the hand-penetration kernel variants on a frozen chain state (A/B in one process)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
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
    # freeze: evaluate pre-stage on current accepted state
    st._eval_pre(st.hand_pose, st.contact_idx, _C.stream_ptr())
    res = {}
    for mode in (2, 3, 1, 0):
        st.penetration_only = 1 if mode == 3 else mode
        ws_keep = st.pen_ws
        if mode == 3: st.pen_ws = None  # single-kernel occupancy path
        for _ in range(3): st._eval_pen(st.hand_pose, _C.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): st._eval_pen(st.hand_pose, _C.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 20 * 1e3
        st.pen_ws = ws_keep
        if mode == 2: ref = st.pen_dis.clone()
        else:
            pos = ref > 1e-6
            err = (st.pen_dis[pos]-ref[pos]).abs().max().item(); assert err < 3e-5, (mode, err)
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    import ctypes
    for mode in (2, 1):
        cnt.zero_(); st.penetration_only = mode
        _C.call("gq_debug_set_pen_counters", ctypes.c_void_p(cnt.data_ptr()))
        st._eval_pen(st.hand_pose, _C.stream_ptr()); torch.cuda.synchronize()
        _C.call("gq_debug_set_pen_counters", None)
        print(f"   mode {mode}: needing pairs {int(cnt[0])}, (wave,link) evals {int(cnt[1])} of {10240*14}, (wave,sub) evals {int(cnt[2])} -> faces/wave-eval {16*int(cnt[2])/max(int(cnt[1]),1):.0f}")
    npos = int((ref > 0).sum())
    print(f"after {done:5d} steps: AABB only {res[2]:7.1f} us | occupancy grid {res[3]:7.1f} us | +queue {res[1]:7.1f} us | exact {res[0]:7.1f} us | penetrating points {npos} / {ref.numel()}  mean E {st.energy.mean().item():.2f}")
    st.penetration_only = 1
