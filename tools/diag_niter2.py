"""Histogram of qpth's batch-global stop iteration over a bench-like run (development aid)."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from graspqp_amd import ops
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes
from bench import make_initial_state
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 1
spec = get_hand_spec("allegro")
fvs = [meshes.superquadric(o) for o in range(n_obj)]
sps = [meshes.surface_points(f, 2500, oversample=4, seed=42) for f in fvs]
st = GraspStepper(ops.HandHandle(spec), ops.MeshSet(fvs), torch.tensor(np.stack(sps)), 256, 12, seed=1)
hps, idxs = zip(*[make_initial_state(spec, f, 256, 12, 1000 + o) for o, f in enumerate(fvs)])
st.reset(torch.cat(hps).cuda(), torch.cat(idxs).cuda())
h = collections.Counter()
for i in range(600):
    st.step()
    if i % 3 == 0:
        h[int(st.n_iter.item())] += 1
print("n_iter histogram (every 3rd of 600 iterations):", sorted(h.items()))
