"""Register / scratch budgets of the hot kernels, read from the built library's code-object metadata (no GPU needed).

On gfx950 the VGPR count fixes the wavefronts per SIMD (<= 64 -> 8, <= 128 -> 4, <= 168 -> 3).  The kernels below are
latency-bound and were tuned at a specific occupancy; a change that silently pushes one of them over its boundary costs
5-15 % (round 2: the stop-rule epilogue took gq_stage_a_kernel from 126 to 166 registers and the penetration role from 4
to 3 wavefronts per SIMD until it was split off into its own instantiation)."""
import os
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))

LIB = os.path.join(ROOT, "graspqp_amd", "lib", "libgraspqp_hip.so")

BUDGET = {  # kernel -> (max VGPRs, max scratch bytes per lane)
    "gq_stage_a_kernel<1, false, 2>": (128, 0),   # config 2: fc head + penetration query, 4 wavefronts per SIMD
    "gq_stage_a_kernel<1, true, 2>": (128, 32),    # 257..511 rows in one grid: with the stop-rule epilogue, capped
    "gq_sdf_wave_kernel<4>": (128, 16),            # capped by __launch_bounds__(256, 4)
    "gq_pen_grid_kernel<true, 512, 4096, 1>": (64, 0),
    "gq_fc_head_kernel<1>": (128, 0),
    "gq_fc_head_stop_kernel<1>": (128, 32),     # large batches: capped (4 wavefronts per SIMD beside the other branch); 5 words spill
    "gq_fk_backward_kernel": (128, 0),
    "gq_fk_forward_row_kernel": (128, 0),       # kinematics without contact queries (large batches): nothing spills
    "gq_fc_tail_kernel<1, 0>": (128, 16),
    "gq_fk_forward_kernel": (160, 0),           # 12 wavefronts per block (hardware limit 170); since round 3 two records in
                                                # flight per query round: 156 VGPRs, nothing spills
    "gq_stage_b_kernel<1, 4>": (168, 0),
    "gq_hand_pen_bwd_kernel<10>": (128, 0),     # <= 2560 surface points: ten slices per round, 4 wavefronts per SIMD
    "gq_stage_alt_kernel<1, 2>": (128, 0),      # dexgrasp role + penetration query
    "gq_stage_alt_kernel<2, 2>": (128, 32),     # tdg role (49 accumulators per thread) capped to the query's occupancy: 5 words spill
}


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_hot_kernels_stay_within_their_register_budget():
    from kernel_resources import kernel_resources

    res = kernel_resources(LIB)
    assert len(res) > 40, "metadata of the code objects not found"
    for name, (vmax, smax) in BUDGET.items():
        assert name in res, f"{name} is missing from the library (renamed? update the budget table)"
        r = res[name]
        assert r["vgpr"] + r["agpr"] <= vmax, (name, r)
        assert r["scratch"] <= smax, (name, r)
    spilled = {k: v["scratch"] for k, v in res.items() if v["scratch"] > 0}
    assert set(spilled) <= {"gq_sdf_wave_kernel<4>", "gq_fc_head_stop_kernel<1>",
                            "gq_stage_a_kernel<1, true, 2>", "gq_stage_a_kernel<1, true, 1>", "gq_fc_tail_kernel<1, 0>",
                            "gq_stage_alt_kernel<2, 2>", "gq_stage_alt_kernel<2, 1>"}, f"new register spills: {spilled}"
