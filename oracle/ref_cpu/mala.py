"""Oracle MALA* optimiser step with the random draws injected.  TEST INFRASTRUCTURE.

Restates ``MalaStar.try_step / accept_step`` (reference ``core/optimizer.py:199-273, 289-340``) and
the loop body of ``scripts/fit.py:399-458`` as pure functions of explicit state, so that the HIP
path and the reference class (run in the build container on the oracle models, fixtures
``tests/golden/mala_*.npz``) can be compared on identical random numbers.

State: hand_pose (B,D), contact idx (B,n) int64, grad (B,D), ema (B,D), step (B,) int64, energy (B,).
"""

import math

import torch


def propose(hand_pose, grad, ema, step, idx, u_switch, new_idx, n_candidates=None, step_size=0.005,
            stepsize_period=50, decay=0.95, mu=0.98, switch_possibility=0.4, clip_grad=False):
    """optimizer.py:199-273.  ``u_switch`` (B,n) uniform draws, ``new_idx`` (B,n) candidate draws
    (only entries where u_switch < switch_possibility are used).  Returns hp', idx', ema', step', s."""
    dt = hand_pose.dtype
    s = step_size * decay ** torch.div(step, stepsize_period, rounding_mode="floor").to(dt)
    g = grad
    if clip_grad:
        g = g.clip(min=-100, max=100)
        g = torch.where(torch.isnan(g), torch.zeros_like(g), g)
    g2 = (g**2).mean(0)  # mean over ALL rows in the process -> (D,)
    ema = mu * g2 + (1 - mu) * ema  # (D,) broadcast into (B,D)
    ema = torch.where(torch.isnan(ema), torch.zeros_like(ema), ema)
    hp = hand_pose - s[:, None] * g / (torch.sqrt(ema) + 1e-6)
    nan_rows = hp.isnan().any(dim=-1)
    hp = torch.where(nan_rows[:, None], torch.zeros_like(hp), hp)
    mask = u_switch < switch_possibility
    idx2 = torch.where(mask, new_idx, idx)
    return hp, idx2, ema, step + 1, s


def z_score(energy, batch_size_each):
    """fit.py:403-406 (per-object mean / unbiased std of the CURRENT accepted energies)."""
    e = energy.view(-1, batch_size_each)
    return ((e - e.mean(-1, keepdim=True)) / e.std(-1, keepdim=True)).view(-1)


def accept(energy, new_energy, step, u_accept, z=None, reset_mask=None, starting_temperature=18.0,
           decay=0.95, annealing_period=30):
    """optimizer.py:289-316 -> (accept (B,) bool, temperature (B,)).  ``step`` is the post-propose counter."""
    dt = energy.dtype
    T = starting_temperature * decay ** torch.div(step, annealing_period, rounding_mode="floor").to(dt)
    if z is not None:
        proba = 0.5 * (1 + torch.erf(z / math.sqrt(2.0)))
        T = T * (1 + proba)
    acc = u_accept < torch.exp((energy - new_energy) / T)
    if reset_mask is not None:
        acc = acc | reset_mask
    return acc, T


def merge(acc, new, old):
    """Rows that are rejected keep the old value (optimizer.py:325-338, fit.py:454-458)."""
    m = acc.view(-1, *([1] * (new.dim() - 1)))
    return torch.where(m, new, old)
