"""Fold a rocprofv3 kernel-stats CSV and the PMC traffic summary (tools/pmc_traffic.py) of the same bench command into
the per-launch table bench.py attaches to its line: duration, HBM bytes per launch (counters), physical HBM rate and
fraction of the 8 TB/s peak.  Only the kernels of the MALA* iteration (gq_*) are listed.

usage: python tools/launch_table.py <kernel_stats.csv> <pmc_traffic.json> <out.json>"""
import csv
import json
import sys

stats, traffic, out = sys.argv[1:4]
pm = json.load(open(traffic))
rows = []
for r in csv.DictReader(open(stats)):
    name = r["Name"].replace("void ", "")
    if not name.startswith("gq_"):
        continue
    short = name.split("(")[0]
    t_us = float(r["AverageNs"]) / 1e3
    e = {"name": short, "calls": int(r["Calls"]), "avg_us": t_us, "pct_of_gpu_time": float(r["Percentage"])}
    p = pm.get(short)
    if p:
        b = p["hbm_bytes_per_launch"]
        e.update(hbm_bytes_per_launch=b, hbm_gbps=b / (t_us * 1e-6) / 1e9, hbm_frac=b / (t_us * 1e-6) / 8e12)
    rows.append(e)
json.dump({"source": "rocprofv3 --kernel-trace --stats + separate --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py "
                     "--no_cpu_baseline` (tools/profile_round.sh); FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md",
           "hbm_peak_gbps": 8000.0, "kernels": rows}, open(out, "w"), indent=1)
for e in rows:
    print(f"{e['name'][:44]:44s} {e['calls']:6d} x {e['avg_us']:8.2f} us   "
          + (f"{e['hbm_bytes_per_launch']/1e6:8.2f} MB  {e['hbm_gbps']:8.1f} GB/s  {100*e['hbm_frac']:5.2f} % of HBM peak" if "hbm_frac" in e else ""))
