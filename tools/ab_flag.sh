#!/bin/bash
# A/B of a bench.py flag in ONE gpurun call (same box): tools/ab_flag.sh "<flags A>" "<flags B>"
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then fl="$1"; else fl="$2"; fi
    python bench.py --steps 400 --warmup 40 --no_cpu_baseline --event_steps 0 $fl | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value']), d['ms_per_step'], d['mean_energy'])"
  done
done
