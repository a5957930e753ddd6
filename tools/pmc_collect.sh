#!/bin/bash
# HBM traffic per launch of the stepper kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the
# default bench command, summarised by tools/pmc_traffic.py (run on the GPU box; writes gpurun_out/pmc_traffic.json)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python $R/bench.py --no_cpu_baseline --steps 40 --warmup 8 > /tmp/pmc_$c.log 2>&1
  echo "pass $c done"
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE/run_counter_collection.csv /tmp/pmc_WRITE_SIZE/run_counter_collection.csv $R/gpurun_out/pmc_traffic.json
