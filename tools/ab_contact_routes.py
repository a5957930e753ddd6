#!/usr/bin/env python3
"""Which compute_sdf route for which (N queries, F faces): the cluster search (one wavefront per query, coalesced), the box
hierarchy (one query per lane; LDS-resident up to ~450 faces, global memory above) and the face loop, timed on the same
queries.  Development aid (GPU box):  python tools/ab_contact_routes.py [hand ...]   (no argument: synthetic meshes;
with hand names: the reference's per-link call shapes of those hands)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from graspqp_amd import ops
from graspqp_amd.utils import meshes


def t(fn, reps=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return round(float(np.median(ts)), 1)


def synthetic():
    rng = np.random.default_rng(0)
    ico3 = meshes.icosphere(3, 0.05).astype(np.float32)
    cases = {"superquadric_9024": meshes.superquadric(0).astype(np.float32), "icosphere_1280": ico3, "open_720": ico3[:720],
             "open_480": ico3[:480], "open_400": ico3[:400]}
    out = {}
    for name, fv in cases.items():
        ctr = fv.mean(1)
        ms, bvh = ops.MeshSet([fv]), ops.Bvh(fv)
        f = torch.tensor(fv, device="cuda")
        for N in (3072, 49152, 640000):
            pts = (ctr[rng.integers(0, len(ctr), N)] + rng.normal(size=(N, 3)) * 0.02).astype(np.float32)
            p = torch.tensor(pts, device="cuda")
            r = {"cluster_us": t(lambda: ops._sdf_meshset_op._init_fn(p, ms.hid, N)),
                 "hierarchy_us": t(lambda: ops._sdf_bvh_op._init_fn(p, bvh.hid))}
            if N * len(fv) <= 3e9:
                r["face_loop_us"] = t(lambda: ops._compute_sdf_op._init_fn(p, f))
            out[f"{name} N={N}"] = r
            print(name, N, r, flush=True)
    print(json.dumps(out))



def links(hand_name, batch=256):
    """The reference's per-link calls (N = batch x 2500 object surface points in the link frame) of another hand, per route."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import plugin_surface as ps
    from bench import make_initial_state
    from graspqp_amd.hands import get_hand_spec

    spec = get_hand_spec(hand_name)
    fvo = meshes.superquadric(0)
    sp = meshes.surface_points(fvo, 2500, oversample=4, seed=42)
    hp, idx = make_initial_state(spec, fvo, batch, 12, 1000)
    hand = ops.HandHandle(spec)
    surf = torch.tensor(sp, dtype=torch.float32, device="cuda")[None].contiguous()
    pts = ps.link_frame_points(spec, hand, hp.cuda().float().contiguous(), idx.cuda().contiguous(), surf, batch)
    tot = {"cluster_us": 0.0, "hierarchy_us": 0.0, "face_loop_us": 0.0, "routed_us": 0.0}
    for l in range(spec.n_links):
        fv = np.ascontiguousarray(spec.link_faces(l), dtype=np.float32)
        if len(fv) == 0:
            continue
        ms, bvh, f, p = ops.MeshSet([fv]), ops.Bvh(fv), torch.tensor(fv, device="cuda"), pts[l]
        N = p.shape[0]
        r = {"F": len(fv), "cluster_us": t(lambda: ops._sdf_meshset_op._init_fn(p, ms.hid, N), 5),
             "hierarchy_us": t(lambda: ops._sdf_bvh_op._init_fn(p, bvh.hid), 5),
             "face_loop_us": t(lambda: ops._compute_sdf_op._init_fn(p, f), 5), "routed_us": t(lambda: ops.compute_sdf(p, f), 5)}
        for k in tot:
            tot[k] += r[k]
        print(hand_name, l, r, flush=True)
    print(json.dumps({"hand": hand_name, "total_us": {k: round(v, 1) for k, v in tot.items()}}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for h in sys.argv[1:]:
            links(h)
    else:
        synthetic()
