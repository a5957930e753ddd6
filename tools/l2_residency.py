#!/usr/bin/env python3
"""L2 residency of the FK-forward launch (kinematics + the row's 12 contact queries of the object SDF) at BASELINE
configs[1]: the same launch on the same state (a) back to back, (b) after the other three launches of an iteration,
(c) after 256 MB of unrelated traffic.  Two uses:

  python tools/l2_residency.py time             -> JSON line with HIP-event durations of the three cases
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  -- python3 tools/l2_residency.py pmc      (then)
  python tools/l2_residency.py fold <counter_collection.csv> [...more passes]   -> per-case counter values per launch

The cases are told apart in the counter file by marker launches (gq_fill with distinct grid sizes) dispatched before each
group.  Development / profiling aid (run on the GPU box).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

REPS = 12


def setup():
    import numpy as np
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
    st = GraspStepper(hand, ops.MeshSet([fv]), torch.tensor(sp[None]), 256, 12, seed=1)
    hp, idx = make_initial_state(spec, fv, 256, 12, 1000)
    st.reset(hp.cuda(), idx.cuda())
    for _ in range(200):
        st.step()
    torch.cuda.synchronize()
    return st


def run(mode):
    import numpy as np
    import torch

    from graspqp_amd import _C

    st = setup()
    pose, idx = st.hand_pose.clone(), st.contact_idx.clone()
    junk = torch.empty(64 * 1024 * 1024, device="cuda")
    mark = [torch.empty(256 * (10 + i), device="cuda") for i in range(4)]  # gq_fill launches of distinct grid sizes
    stream = _C.stream_ptr()

    def fk():
        st._eval_fk(pose, idx, stream, False, sdf=True, spheres=False)

    def others():  # the other three launches of the fused iteration on the same state
        st._pen_desc.hand_pose = pose.data_ptr()
        _C.call("gq_fc_pen_step", __import__("ctypes").byref(st._fc_desc), __import__("ctypes").byref(st._pen_desc), stream)
        st._eval_tail(pose, idx, stream, False)

    def timed(pre):
        ts = []
        for _ in range(REPS):
            pre()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fk()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(ts[2:])), float(np.min(ts[2:]))

    fk()
    torch.cuda.synchronize()
    out = {}
    for i, (name, pre) in enumerate((("back_to_back", lambda: None), ("after_the_other_launches_of_an_iteration", others),
                                     ("after_256MB_of_unrelated_traffic", lambda: junk.fill_(1.0)))):
        _C.call("gq_fill", _C.f32(mark[i]), float(i), mark[i].numel(), stream)  # marker dispatch: tells the groups apart
        torch.cuda.synchronize()
        out[name] = dict(zip(("us_median", "us_min"), timed(pre)))
    _C.call("gq_fill", _C.f32(mark[3]), 3.0, mark[3].numel(), stream)
    torch.cuda.synchronize()
    if mode == "time":
        print(json.dumps({"fk_forward_with_contact_queries_us": out, "reps": REPS - 2}))


def fold(paths):
    """Counter values of gq_fk_forward_kernel per launch, grouped by the marker fills (element counts 1000, 1007, 1014)."""
    res = {}
    for p in paths:
        rows = list(csv.DictReader(open(p)))
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", r.get("Dispatch_ID", 0))))
        group, names = -1, ["warm_up", "back_to_back", "after_the_other_launches_of_an_iteration", "after_256MB_of_unrelated_traffic", "end"]
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        seen = set()
        for r in rows:
            k = r["Kernel_Name"]
            if "gq_fill" in k:  # markers: grid sizes 2560 + 256 i
                g = int(float(r.get("Grid_Size", 0) or 0))
                if g in (2560, 2816, 3072, 3328) and r.get("Dispatch_Id") not in seen:
                    seen.add(r.get("Dispatch_Id"))
                    group = (g - 2560) // 256
                continue
            if "gq_fk_forward_kernel" in k and 0 <= group < 3:
                acc[names[group + 1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for gname, cs in acc.items():
            for cname, v in cs.items():
                res.setdefault(gname, {})[cname] = {"per_launch_mean": sum(v[2:]) / max(len(v[2:]), 1), "launches": len(v)}
    for gname, cs in res.items():
        if "FETCH_SIZE" in cs:
            cs["fabric_read_MB_per_launch"] = 2.0 * cs["FETCH_SIZE"]["per_launch_mean"] * 1024 / 1e6  # gfx950: 128-B requests tallied at 64 B
        if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
            h, m = cs["TCC_HIT_sum"]["per_launch_mean"], cs["TCC_MISS_sum"]["per_launch_mean"]
            cs["l2_hit_rate"] = h / max(h + m, 1.0)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "time"
    if mode == "fold":
        fold(sys.argv[2:])
    else:
        run(mode)
