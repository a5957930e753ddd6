"""Distribution of qpth's batch-global stop iteration over a bench-like run (development aid)."""
import os, sys, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from graspqp_amd import ops
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
hist = collections.Counter()
for i in range(1200):
    st.step()
    hist[int(st.n_iter.item())] += 1
    if i in (99, 299, 599, 1199):
        print(i + 1, dict(sorted(hist.items())))
