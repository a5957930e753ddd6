#!/bin/bash
# Round-end validation on the GPU box, one gpurun call: the whole GPU suite, then the profile round (kernel stats, PMC
# traffic, launch table), the instruction counters, the default bench line and the other-configuration lines.
# usage (through gpurun): bash tools/final_round.sh <tag>      -> gpurun_out/<tag>_*; copy them into profiles/ afterwards
TAG=${1:-r02}
python -m pytest tests -m gpu -x -q > gpurun_out/gputest_final.log 2>&1; rc=$?
tail -3 gpurun_out/gputest_final.log
[ $rc = 0 ] || exit $rc
set -e
bash tools/profile_round.sh $TAG > gpurun_out/prof_round.log 2>&1
cp gpurun_out/${TAG}_launches.json gpurun_out/${TAG}_pmc_traffic.json profiles/
bash tools/pmc_instr.sh ${TAG}_cfg2 > gpurun_out/pmc_instr.log 2>&1
cp gpurun_out/${TAG}_cfg2_instr.json profiles/
python bench.py > gpurun_out/${TAG}_default_bench.json 2> gpurun_out/${TAG}_default_bench.err
bash tools/other_configs.sh > gpurun_out/${TAG}_other_configs_bench.json 2> gpurun_out/other.err
python - <<EOF
import json,csv
d=json.loads(open("gpurun_out/${TAG}_default_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], r["kernel_ms"], r["utilisation"]["kernel_ms_isolated"], r["utilisation"]["kernel_ms_isolated_hip_events"], r["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline_config0"]["value"])
for row in csv.DictReader(open("gpurun_out/${TAG}_default_bench_kernel_stats.csv")):
    if "gq_" in row["Name"] and int(row["Calls"])>40: print(row["Name"][:50], row["Calls"], round(float(row["AverageNs"])/1e3,2), round(float(row["MinNs"])/1e3,2), round(float(row["MaxNs"])/1e3,2))
for l in open("gpurun_out/${TAG}_other_configs_bench.json"):
    l=l.strip()
    if l.startswith("{"):
        e=json.loads(l); print(e["config"]["workload"][:50], round(e["value"]), round(e["ms_per_step"],4))
EOF
