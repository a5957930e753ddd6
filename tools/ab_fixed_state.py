#!/usr/bin/env python3
"""A/B of library builds on ONE fixed optimizer state (the MALA* trajectories of two builds drift apart after a few hundred
iterations, and the cost of the penetration query depends on the state, so end-of-run timings of different builds are not
comparable to the per cent).

  python tools/ab_fixed_state.py save <state.pt> [n_objects]      run 400 iterations with the default build, keep the state
  GRASPQP_HIP_LIB=... python tools/ab_fixed_state.py time <state.pt> [n_objects]
        -> HIP-event medians of an energy+gradient evaluation of exactly that state (fused and per-role form)

Development aid (GPU box).
"""
import ctypes
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def stepper(n_obj):
    import torch

    from bench import make_initial_state
    from graspqp_amd import ops
    from graspqp_amd.hands import get_hand_spec
    from graspqp_amd.stepper import GraspStepper
    from graspqp_amd.utils import meshes

    spec = get_hand_spec("allegro")
    fvs = [meshes.superquadric(i) for i in range(n_obj)]
    sps = [meshes.surface_points(fv, 2500, oversample=4, seed=42 + i) for i, fv in enumerate(fvs)]
    st = GraspStepper(ops.HandHandle(spec), ops.MeshSet(fvs), torch.tensor(__import__("numpy").stack(sps)), 256, 12, seed=1)
    hps, idxs = zip(*[make_initial_state(spec, fv, 256, 12, 1000 + i) for i, fv in enumerate(fvs)])
    return st, torch.cat(hps).cuda(), torch.cat(idxs).cuda()


def main():
    import numpy as np
    import torch

    from graspqp_amd import _C

    mode, path = sys.argv[1], sys.argv[2]
    n_obj = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    st, hp, idx = stepper(n_obj)
    if mode == "save":
        st.reset(hp, idx)
        for _ in range(400):
            st.step()
        torch.cuda.synchronize()
        torch.save({"hand_pose": st.hand_pose.cpu(), "contact_idx": st.contact_idx.cpu()}, path)
        print("saved", path, "mean energy", float(st.energy.mean()))
        return
    d = torch.load(path, weights_only=True)
    st.reset(d["hand_pose"].cuda(), d["contact_idx"].cuda())
    pose, ix = st.hand_pose.clone(), st.contact_idx.clone()
    stream = _C.stream_ptr()
    out = {}
    for name, fused in (("fused_four_launches_us", True), ("per_role_launches_one_stream_us", False)):
        ts = []
        for rep in range(60):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st._evaluate(pose, ix, stream, fused=fused)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        out[name] = round(float(np.median(ts[5:])), 2)
    st.start_kernel_timing()  # the hand-penetration query alone (per-role form)
    for rep in range(30):
        st._evaluate(pose, ix, stream)
    torch.cuda.synchronize()
    st.kernel_events = None
    out["pen_query_kernel_span_us"] = round(st.kernel_times_ms()[1] * 1e3, 2)  # in-kernel s_memrealtime span
    out["lib"] = os.environ.get("GRASPQP_HIP_LIB", "default")[-40:]
    out["energy_mean"] = float(st.energy.mean())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
