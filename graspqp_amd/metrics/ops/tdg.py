"""TDG (task-oriented grasp-wrench-space) metric with the reference's surface (metrics/ops/tdg.py:219-239) on the HIP
kernel.  Configuration as in the reference: friction 0.2, no soft-finger term, object box length 0.2, contact density
weighting on, 1000 target directions drawn once with ``np.random.randn`` at construction (tdg.py:147-160)."""

import numpy as np
import torch

from ... import ops


def random_sample_points_on_sphere(dim_num, point_num):
    """reference tdg.py:48-52 (same use of the global numpy stream)."""
    points = np.random.randn(point_num, dim_num)
    return points / (np.linalg.norm(points, axis=-1)[:, None] + 1e-8)


class TDGSpanMetric(torch.nn.Module):
    def __init__(self, device="cuda", directions=None):
        super().__init__()
        self.miu_coef = [0.2, 0.0]
        self.obj_obb_length = 0.2
        self.enable_density = True
        self.direct_num = 1000
        d = random_sample_points_on_sphere(3, self.direct_num) if directions is None else np.asarray(directions)
        self.target_direction = torch.tensor(d, dtype=torch.float32, device=device).contiguous()  # (P,3); torque part is 0

    def forward(self, contact_pts, contact_normals, cog, torque_weight=0.0, with_solution=False, **kwargs):
        e = ops.tdg_energy(contact_pts, contact_normals, cog, self.target_direction, self.miu_coef[0], self.obj_obb_length,
                           self.enable_density, 100.0)
        return e, None
