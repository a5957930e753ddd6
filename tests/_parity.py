"""Tail assertions of the parity contract, shared by the GPU tests.

E_fc: <= 1e-4 rel at the median and, at p99 / max, <= 2x the error the ORACLE ITSELF makes when it runs the same algorithm
in fp32 instead of fp64 on the same inputs (the PDIPM iterate path has an fp32 noise tail of 1e-4 .. 1e-3 on ~1 % of the
rows; measured per configuration in profiles/r03_parity_report.json).  Errors below the contract's per-row figure of 1e-4
need no such excuse, hence the floor.  Total energy: <= 1e-4 rel per row with the same noise allowance.
"""
import numpy as np


def rel_err(a, b, floor=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def assert_tail_within_fp32_noise(rel_hip, rel_noise, what, median_cap=1e-4, floor=1e-4):
    rel_hip, rel_noise = np.asarray(rel_hip), np.asarray(rel_noise)
    assert np.median(rel_hip) < median_cap, f"{what}: median {np.median(rel_hip):.3g}"
    for q in (99, 100):
        h, o = np.percentile(rel_hip, q), np.percentile(rel_noise, q)
        assert h <= max(2.0 * o, floor), f"{what}: p{q} {h:.3g} > 2 x oracle fp32 noise {o:.3g} (floor {floor:g})"


def oracle_fc_fp32_noise(ospan, cpts, onrm, cog, k, e64=None):
    """rel error of the oracle's E_fc in fp32 against its own fp64 result on the same batch (reduced nz x nz form: the
    tighter of the two oracle forms; qpth's 2nz x 2nz block form in fp32 is 4-100x noisier, profiles/r03_parity_report.json)."""
    import torch

    c, o, g = (torch.as_tensor(t).detach() for t in (cpts, onrm, cog))
    if e64 is None:
        e64, _ = ospan.e_fc(c.double(), o.double(), g.double(), k=k, box_form=True)
    e32, _ = ospan.e_fc(c.float(), o.float(), g.float(), k=k, box_form=True)
    return rel_err(e32.detach().double().numpy(), torch.as_tensor(e64).detach().double().numpy())
