#!/bin/bash
# kernel-level A/B on one box: rocprofv3 stats of the stepper kernels for lib A and the current lib
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  if [ $v = A ]; then export GRASPQP_HIP_LIB=$GRAFT_REPO_ROOT/graspqp_amd/lib/libgraspqp_hip_A.so; else unset GRASPQP_HIP_LIB; fi
  rm -rf /tmp/abp_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$v -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 40 --no_cpu_baseline --event_steps 100 > /tmp/abp_$v.log 2>&1
  echo "== $v"; python - <<PY
import csv
rows=list(csv.reader(open('/tmp/abp_$v/run_kernel_stats.csv')))
for r in rows[1:12]: print(r[0][:44].ljust(44), r[1], r[3][:8])
PY
done
