#!/bin/bash
# L2 residency of the FK-forward launch at BASELINE configs[1] (tools/l2_residency.py): durations, then separate rocprofv3
# --pmc passes (kernel trace only) for the fabric reads and the L2 hit / miss counts of the same three cases.
# usage (GPU box): bash tools/l2_residency.sh <tag>      -> gpurun_out/<tag>_l2_residency.json
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/l2_residency.py time > $OUT/${TAG}_l2_residency_times.json 2> /tmp/l2_time.err
echo "timing pass done"
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/l2p_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/l2p_$i -o run -- python3 $R/tools/l2_residency.py pmc > /tmp/l2p_$i.log 2>&1 || { tail -5 /tmp/l2p_$i.log; }
  echo "pass $i ($set) done"
done
python3 $R/tools/l2_residency.py fold /tmp/l2p_1/run_counter_collection.csv /tmp/l2p_2/run_counter_collection.csv /tmp/l2p_3/run_counter_collection.csv > $OUT/${TAG}_l2_residency_counters.json
head -3 /tmp/l2p_1/run_counter_collection.csv > $OUT/${TAG}_l2_counter_csv_head.txt
cat $OUT/${TAG}_l2_residency_times.json; cat $OUT/${TAG}_l2_residency_counters.json
