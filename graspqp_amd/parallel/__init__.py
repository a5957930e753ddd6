"""Multi-GPU layout of the grasp loop: one process per GPU, whole objects per rank, no collective in the loop.

This is the reference's own scale-out model (one ``fit.py`` process per batch of assets, scripts/fit_all.sh:181-208):
the three cross-row statistics of an iteration -- the RMS gradient mean over all rows (core/optimizer.py:231), the
per-object z-score (scripts/fit.py:403-406) and the QP's batch-global stopping rule -- are per-process in the
reference, so "per rank == per process" keeps parity without any exchange.  The only collective is an optional
gather of the final (hand_pose, energy, contact_idx) over RCCL.
"""

from typing import List

import torch


def shard_objects(n_objects: int, world_size: int, rank: int) -> List[int]:
    """Contiguous block of object ids owned by ``rank`` (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    q, r = divmod(n_objects, world_size)
    start = rank * q + min(rank, r)
    return list(range(start, start + q + (1 if rank < r else 0)))


def gather_results(hand_pose: torch.Tensor, energy: torch.Tensor, contact_idx: torch.Tensor, dst: int = 0):
    """Gather the per-rank final state on ``dst`` (ragged shards allowed).  Returns lists on ``dst``, None elsewhere.

    Works with the ``nccl`` (= RCCL) backend on GPU tensors and with ``gloo`` on CPU tensors (tests)."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [hand_pose], [energy], [contact_idx]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([hand_pose.shape[0]], device=hand_pose.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = int(max(int(s) for s in sizes))

    def pad(t):
        if t.shape[0] == mx:
            return t.contiguous()
        p = torch.zeros(mx - t.shape[0], *t.shape[1:], dtype=t.dtype, device=t.device)
        return torch.cat([t, p]).contiguous()

    outs = []
    for t in (hand_pose, energy, contact_idx):
        buf = [torch.empty_like(pad(t)) for _ in range(world)]
        dist.all_gather(buf, pad(t))
        outs.append([b[: int(s)] for b, s in zip(buf, sizes)])
    return tuple(outs) if rank == dst else (None, None, None)
