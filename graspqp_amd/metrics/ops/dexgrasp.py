"""DexGraspNet force-closure metric with the reference's surface (metrics/ops/dexgrasp.py) on the HIP kernel."""

import torch

from ... import ops


def calc_e_fc(contact_pts: torch.Tensor, contact_normals: torch.Tensor, torque_weight: float = 1.0) -> torch.Tensor:
    """reference dexgrasp.py:4-34 -- contact_pts are taken relative to the origin there (the caller subtracts cog)."""
    zero = torch.zeros(contact_pts.shape[0], 3, device=contact_pts.device)
    return ops.dexgrasp_energy(contact_pts, contact_normals, zero, torque_weight)


class DexgraspSpanMetric(torch.nn.Module):
    def __init__(self) -> None:
        super().__init__()

    def forward(self, contact_pts, contact_normals, cog, torque_weight: float = 0.0, with_solution: bool = False, **kwargs):
        e = ops.dexgrasp_energy(contact_pts, contact_normals, cog, torque_weight)  # cog subtraction inside the kernel
        if with_solution:
            return e, torch.ones_like(contact_pts[..., 0])
        return e
