#!/bin/bash
# kernel-level A/B of a bench.py flag on one box: rocprofv3 stats of the stepper kernels (graph replay only)
cd /tmp && export TMPDIR=/tmp
i=0
for fl in "$@"; do
  i=$((i+1)); rm -rf /tmp/pf_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$i -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no_cpu_baseline --event_steps 0 $fl > /tmp/pf_$i.log 2>&1
  echo "== $fl"; tail -1 /tmp/pf_$i.log | cut -c1-160
  python - <<PY
import csv
rows=list(csv.reader(open('/tmp/pf_$i/run_kernel_stats.csv')))
for r in rows[1:9]: print(r[0][:60].ljust(60), r[1], r[3][:8])
PY
done
