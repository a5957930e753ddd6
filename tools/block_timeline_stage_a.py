"""Block timeline of the penetration query as the ROLE of stage A (config 2), with the CU every block ran on: are the query
blocks that share a CU with force-closure rows the slow ones?  Needs the -DGQ_BLOCK_TIMES build (tools/block_timeline.sh)
through GRASPQP_HIP_LIB."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from graspqp_amd import ops, _C
from graspqp_amd.hands import get_hand_spec
from graspqp_amd.stepper import GraspStepper
from graspqp_amd.utils import meshes
from bench import make_initial_state

assert "libgraspqp_hip_A" in os.environ.get("GRASPQP_HIP_LIB", ""), "run with the -DGQ_BLOCK_TIMES build (tools/block_timeline.sh)"
spec = get_hand_spec("allegro")
fv = meshes.superquadric(0)
sp = meshes.surface_points(fv, 2500, oversample=4, seed=42)
hand = ops.HandHandle(spec)
st = GraspStepper(hand, ops.MeshSet([fv]), torch.tensor(sp)[None], 256, 12, seed=1)
B = st.B
nq, nfc = B * 5, B // 4  # two points per thread: five query blocks per row; four fc rows per block
st._span = torch.zeros(64 + 4 * (B * 10 + nfc + 16), 2, dtype=torch.int64, device="cuda")  # room for either block count
st._span[:64, 0] = -1
st._pen_desc.span = st._span.data_ptr()
hp, idx = make_initial_state(spec, fv, 256, 12, 1000)
st.reset(hp.cuda(), idx.cuda())
st.capture(iters=8)
assert st.graph_mode == "one grid"
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 600):
    st.step()
st.flush()
torch.cuda.synchronize()
rec = st._span.view(-1)[128:].view(-1, 8).cpu().numpy()
q, fc = rec[:nq], rec[nq:nq + nfc]
cu = lambda r: ((r[:, 6] >> 32) & 0xf) * 4096 + ((r[:, 6] >> 8) & 0xff)  # (XCC, SE | SH | CU) -> one id per CU
t0 = min(q[:, 0].min(), fc[:, 0].min())
start, dur = (q[:, 0] - t0) / 100.0, (q[:, 1] - q[:, 0]) / 100.0
fc_cus = set(cu(fc).tolist())
on_fc = np.array([c in fc_cus for c in cu(q).tolist()])
pc = lambda a: " ".join(f"{np.percentile(a, p):6.1f}" for p in (0, 10, 50, 90, 99, 100)) if len(a) else "-"
print(f"stage A, last captured iteration: {nq} query blocks, {nfc} fc blocks on {len(fc_cus)} CUs; fc rows {((fc[:, 1] - fc[:, 0]) / 100.0).min():.1f}..{((fc[:, 1] - fc[:, 0]) / 100.0).max():.1f} us")
print(f"   query role: first start -> last end {(q[:, 1].max() - q[:, 0].min()) / 100.0:.1f} us")
print(f"   blocks on a CU with fc rows: {int(on_fc.sum())}  duration us (min p10 p50 p90 p99 max): {pc(dur[on_fc])}   end us: {pc((start + dur)[on_fc])}")
print(f"   blocks on other CUs:         {int((~on_fc).sum())}  duration us (min p10 p50 p90 p99 max): {pc(dur[~on_fc])}   end us: {pc((start + dur)[~on_fc])}")
print(f"   start offsets us (min p10 p50 p90 p99 max): {pc(start)}")
bx = np.arange(nq) % 5  # block_id = bx + 5 row
for b in range(5):
    m = bx == b
    print(f"   slice block {b}: duration {pc(dur[m])}  start {pc(start[m])}")
row = np.arange(nq) // 5
qpos = bx * B + row  # dispatch position of the block among the query blocks (slice-major)
print("   dispatch position mod 256 of the blocks that share a CU with fc rows, histogram over [0,64) [64,128) [128,192) [192,256):",
      np.bincount((qpos[on_fc] % 256) // 64, minlength=4).tolist(), " per slice block:", np.bincount(bx[on_fc], minlength=5).tolist())
early = start < 2.0
print("   blocks resident from the start:", int(early.sum()), " of them on fc CUs:", int((early & on_fc).sum()),
      "; late starters on fc CUs:", int((~early & on_fc).sum()), "of", int((~early).sum()))
