import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from graspqp_amd import ops, stepper
from graspqp_amd.hands import get_hand_spec
g = np.load("/root/repo/tests/golden/mala_allegro_sphere_b8_n4.npz")
spec = get_hand_spec("allegro")
hand = ops.HandHandle(spec)
n_obj = int(g["n_obj"]); be = int(g["batch_size_each"])
fvs = [g[f"obj{i}_face_verts"] for i in range(n_obj)]
sps = np.stack([g[f"obj{i}_surface_points"] for i in range(n_obj)])
f32 = lambda k: torch.tensor(g[k], dtype=torch.float32).cuda()
for mode in (1, 3, 2, 0):
    st = stepper.GraspStepper(hand, ops.MeshSet(fvs), torch.tensor(sps), be, 4, penetration_only=mode)
    st.reset(f32("hand_pose0"), torch.tensor(g["contact_idx0"]).cuda())
    out = []
    for s in range(1, int(g["n_steps"]) + 1):
        st.step(draws=(f32(f"s{s}_u_switch"), torch.tensor(g[f"s{s}_new_idx"]).cuda(), f32(f"s{s}_u_accept")))
        torch.cuda.synchronize()
        acc = st.accept.cpu().bool().tolist() == g[f"s{s}_accept"].tolist()
        dp = np.abs(st.hand_pose.cpu().numpy() - g[f"s{s}_hand_pose"]).max()
        de = np.abs(st.energy.cpu().numpy() / g[f"s{s}_energy"] - 1).max()
        out.append((acc, round(float(dp), 5), round(float(de), 5)))
    print("mode", mode, out)
