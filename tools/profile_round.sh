#!/bin/bash
# Per-round profile of the default bench command on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/<tag>_default_bench_kernel_stats.csv (+ the bench line)
#   2. two separate --pmc passes (FETCH_SIZE / WRITE_SIZE, kernel trace only; MI355X_MICROARCH.md, HBM section)
#                                                  -> gpurun_out/<tag>_pmc_traffic.json
#   3. both folded into the per-launch table       -> gpurun_out/<tag>_launches.json (bench.py reads profiles/<tag>_launches.json)
# usage: bash tools/profile_round.sh <tag> [extra bench.py flags]      (copy the outputs into profiles/ afterwards)
set -e
TAG=${1:-r02}; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o run -- python3 $R/bench.py --no_cpu_baseline "$@" > $OUT/${TAG}_default_bench_profiled.json 2> /tmp/prof_stats.err
cp /tmp/prof_stats/run_kernel_stats.csv $OUT/${TAG}_default_bench_kernel_stats.csv
echo "stats pass done"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python3 $R/bench.py --no_cpu_baseline --steps 40 --windows 1 --warmup 8 "$@" > /tmp/pmc_$c.log 2>&1
  echo "pass $c done"
done
python3 $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE/run_counter_collection.csv /tmp/pmc_WRITE_SIZE/run_counter_collection.csv $OUT/${TAG}_pmc_traffic.json > /tmp/pmc_summary.txt
python3 $R/tools/launch_table.py $OUT/${TAG}_default_bench_kernel_stats.csv $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_launches.json
