#!/bin/bash
# Executed-instruction profile of every kernel (rocprofv3 --pmc, kernel trace only, two passes of SQ counters):
#   pass 1: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS        pass 2: SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64
# -> gpurun_out/<tag>_instr.json (tools/pmc_instr.py): instructions per wavefront and the VALU issue-slot utilisation
#    (VALU instructions x 4 cycles / (duration x clock x SIMDs)) next to each kernel's duration.
# usage: bash tools/pmc_instr.sh <tag> [bench.py flags]      (a --stats pass of the same command comes first)
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmci_0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmci_0 -o run -- python3 $R/bench.py --no_cpu_baseline --steps 40 --windows 1 --warmup 8 --event_steps 0 "$@" > /tmp/pmci_0.log 2>&1
STATS=/tmp/pmci_0/run_kernel_stats.csv
echo "stats pass done"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64"; do
  i=$((i+1)); rm -rf /tmp/pmci_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmci_$i -o run -- python3 $R/bench.py --no_cpu_baseline --steps 40 --windows 1 --warmup 8 --event_steps 0 "$@" > /tmp/pmci_$i.log 2>&1
  echo "pass $i done"
done
python3 $R/tools/pmc_instr.py $STATS /tmp/pmci_1/run_counter_collection.csv /tmp/pmci_2/run_counter_collection.csv $R/gpurun_out/${TAG}_instr.json
