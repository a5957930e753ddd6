"""Shared synthetic inputs of the CPU and GPU tests (plain torch, no product / oracle imports)."""
import torch


def hetero_contacts(B, n, seed, dtype=torch.float64):
    """Heterogeneous batch of contact sets (points (B,n,3), object normals (B,n,3), cog (B,3)): contacts on spheres of
    different radii, from well spread (easy rows: the force-closure QP converges in a few iterations) to bunched on one
    side of the object (hard rows: far from force closure, slow PDIPM convergence)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.nn.functional.normalize(torch.randn(B, n, 3, generator=g, dtype=torch.float64), dim=-1)
    bunch = torch.rand(B, 1, 1, generator=g, dtype=torch.float64) ** 2  # 0 = spread, 1 = all contacts on one side
    pole = torch.nn.functional.normalize(torch.randn(B, 1, 3, generator=g, dtype=torch.float64), dim=-1)
    d = torch.nn.functional.normalize(d + 3.0 * bunch * pole, dim=-1)
    pts = d * (0.03 + 0.05 * torch.rand(B, 1, 1, generator=g, dtype=torch.float64))
    nrm = torch.nn.functional.normalize(-d + 0.3 * torch.randn(B, n, 3, generator=g, dtype=torch.float64), dim=-1)
    cog = 0.005 * torch.randn(B, 3, generator=g, dtype=torch.float64)
    return pts.to(dtype), nrm.to(dtype), cog.to(dtype)
