"""Soak run of the reference's schedule (fit.py:399-458: n_iter iterations, re-initialisation of the z-score outliers
every `reset_epochs`) on the bench scene, at one or several batch sizes: energies must stay finite and fall, the debug
counters must not show inline-ranked overflow growing, the stop iteration is histogrammed.  Development aid, not a test.

usage: python tools/soak.py [n_objects ...]        (256 rows each; default 1 8)
"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch

from graspqp_amd import ops
from graspqp_amd.core.object_model import ObjectModel
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes

N_ITER, RESET = int(os.environ.get("SOAK_ITERS", 7000)), 600
spec = get_hand_spec("allegro")
hand = ops.HandHandle(spec)
for n_obj in [int(a) for a in sys.argv[1:]] or [1, 8]:
    fvs = [meshes.superquadric(o) for o in range(n_obj)]
    sps = [meshes.surface_points(f, 2500, oversample=4, seed=42) for f in fvs]
    om = ObjectModel(batch_size_each=256, num_samples=2500)
    om.initialize_from_meshes(fvs, surface_points_list=sps)
    st = GraspStepper(hand, ops.MeshSet(fvs), torch.tensor(np.stack(sps)), 256, 12, seed=3)
    st.set_hulls(om.convex_hulls())
    st.initialize()
    e0 = st.energy.clone()
    st.capture(iters=8)
    hist = {}
    acc = []

    def cb(step):
        if step % 50 == 0:
            k = int(st.n_iter.item())
            hist[k] = hist.get(k, 0) + 1
            acc.append(float(st.accept.float().mean()))
        if step % 1000 == 0:
            e = st.energy
            print(f"  [{n_obj} x 256] step {step:5d}  E mean {float(e.mean()):9.3f}  min {float(e.min()):8.3f}  max {float(e.max()):9.3f}  "
                  f"finite {bool(torch.isfinite(e).all())}  accept {np.mean(acc[-20:]):.3f}", flush=True)

    torch.cuda.synchronize()
    t0 = time.time()
    st.run(N_ITER, reset_epochs=RESET, z_score_threshold=1.0, callback=cb)
    torch.cuda.synchronize()
    dt = time.time() - t0
    e1 = st.energy
    assert torch.isfinite(e1).all() and torch.isfinite(st.hand_pose).all() and torch.isfinite(st.grad).all()
    assert float(e1.mean()) < float(e0.mean()), "the chain must lower the mean energy"
    assert int(st.contact_idx.max()) < spec.n_contact_candidates and int(st.contact_idx.min()) >= 0
    print(f"{n_obj} x 256 rows: {N_ITER} iterations in {dt:.2f} s (callback every step: host-bound), mean E {float(e0.mean()):.2f} -> "
          f"{float(e1.mean()):.2f}, stop-iteration histogram {dict(sorted(hist.items()))}, graph mode {st.graph_mode}", flush=True)
