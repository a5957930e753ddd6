"""Time the contact-point object SDF (gq_sdf_forward_meshset) for query points at several distances from the surface
(development aid; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graspqp_amd import _C, ops
from graspqp_amd.utils import meshes

fv = meshes.superquadric(0)
ms = ops.MeshSet([fv])
sp = meshes.surface_points(fv, 3072, oversample=2)
cen = sp.mean(0)
N = 3072
for dist in (0.0005, 0.005, 0.02, 0.05, 0.15):
    d = sp - cen
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = torch.tensor(sp + d * dist, dtype=torch.float32).cuda().contiguous()
    d2 = torch.empty(N, device="cuda"); sg = torch.empty(N, dtype=torch.int32, device="cuda")
    nr = torch.empty(N, 3, device="cuda"); cl = torch.empty(N, 3, device="cuda")
    def run():
        _C.call("gq_sdf_forward_meshset", ms.handle, _C.f32(pts), N, N, _C.f32(d2), _C.i32(sg), _C.f32(nr), _C.f32(cl), _C.stream_ptr())
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    cnt = torch.zeros(12, dtype=torch.int64, device="cuda")
    _C.call("gq_debug_set_pen_counters", _C.ptr(cnt)); run(); torch.cuda.synchronize(); _C.call("gq_debug_set_pen_counters", None)
    print("  visits per query:", float(cnt[0]) / float(cnt[1]), "max", int(cnt[2]), "queries with > 16 visits", int(cnt[3]))
    print(f"offset {dist*1e3:6.1f} mm: {e0.elapsed_time(e1)*10:.1f} us per launch; mean dist {float(d2.sqrt().mean())*1e3:.2f} mm")
