"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into per-kernel HBM
traffic per launch.  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are
in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled (calibrated there for wide streaming
loads only -- narrower accesses are uncalibrated, so read the result as an estimate); WRITE_SIZE is exact.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "gq_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    out[k] = {"launches": max(nf, nw), "fetch_size_kib_raw": f, "write_size_kib": w,
              "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:36s} {v['launches']:5d} launches  fetch(raw) {v['fetch_size_kib_raw']:9.1f} KiB  write {v['write_size_kib']:9.1f} KiB"
          f"  -> {v['hbm_bytes_per_launch']/1e6:8.2f} MB / launch")
