#!/bin/bash
# A/B of two builds of the library in ONE gpurun call (same box): graspqp_amd/lib/libgraspqp_hip_A.so vs the current one
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export GRASPQP_HIP_LIB=$PWD/graspqp_amd/lib/libgraspqp_hip_A.so; else unset GRASPQP_HIP_LIB; fi
    python bench.py --steps 400 --warmup 40 --no_cpu_baseline --event_steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step'],5), 'query_ms', round(d['roofline']['kernel_ms'],5))"
  done
done
