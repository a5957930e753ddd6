"""Fold rocprofv3 SQ instruction counters (tools/pmc_instr.sh) and the kernel durations of a --stats run into
per-kernel executed work: instructions per wavefront by class and the VALU issue-slot utilisation.

A gfx950 SIMD issues one wave64 VALU instruction per 4 cycles (16 lanes per cycle); 256 CUs x 4 SIMDs at 2.4 GHz give
6.1e11 x 1/4 wave-instructions per second.  issue_frac = VALU instructions x 4 / (duration x 2.4e9 x 1024): the share of
the chip's VALU issue slots the launch used -- the roofline of a kernel that is neither HBM- nor MFMA-bound.

usage: python tools/pmc_instr.py <kernel_stats.csv> <pass1 counter_collection.csv> <pass2 ...> <out.json>
"""
import collections
import csv
import json
import sys

CLOCK_HZ, SIMDS = 2.4e9, 256 * 4


def short(name):
    return name.split("(")[0].replace("void ", "")


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "gq_" in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


dur = {short(r["Name"]): float(r["AverageNs"]) * 1e-9 for r in csv.DictReader(open(sys.argv[1])) if "gq_" in r["Name"]}
cnt = collections.defaultdict(dict)
for p in sys.argv[2:-1]:
    for k, d in counters(p).items():
        cnt[k].update(d)
out = {}
for k, d in sorted(cnt.items(), key=lambda kv: -dur.get(kv[0], 0.0)):
    w = d.get("SQ_WAVES", 0.0)
    if not w or k not in dur:
        continue
    per = lambda c: d.get(c, 0.0) / w
    out[k] = {"duration_us": dur[k] * 1e6, "wavefronts": w, "valu_per_wave": per("SQ_INSTS_VALU"), "salu_per_wave": per("SQ_INSTS_SALU"),
              "lds_per_wave": per("SQ_INSTS_LDS"), "vmem_rd_per_wave": per("SQ_INSTS_VMEM_RD"), "vmem_wr_per_wave": per("SQ_INSTS_VMEM_WR"),
              "smem_per_wave": per("SQ_INSTS_SMEM"), "fma_f64_per_wave": per("SQ_INSTS_VALU_FMA_F64"),
              "valu_issue_frac": d.get("SQ_INSTS_VALU", 0.0) * 4.0 / (dur[k] * CLOCK_HZ * SIMDS)}
json.dump({"clock_hz": CLOCK_HZ, "simds": SIMDS, "kernels": out}, open(sys.argv[-1], "w"), indent=1)
for k, v in out.items():
    print(f"{k[:44]:44s} {v['duration_us']:8.1f} us  waves {v['wavefronts']:9.0f}  VALU/wave {v['valu_per_wave']:8.0f}  SALU {v['salu_per_wave']:6.0f}  "
          f"LDS {v['lds_per_wave']:5.0f}  VMEM {v['vmem_rd_per_wave'] + v['vmem_wr_per_wave']:5.0f}  issue {100 * v['valu_issue_frac']:5.1f} %")
