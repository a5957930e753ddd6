#!/bin/bash
# A/B of two builds of the library at large batches in ONE gpurun call: graspqp_amd/lib/libgraspqp_hip_A.so vs the current one
for no in ${AB_OBJECTS:-2 8}; do
  for rep in 1 2; do
    for v in A B; do
      if [ $v = A ]; then export GRASPQP_HIP_LIB=$PWD/graspqp_amd/lib/libgraspqp_hip_A.so; else unset GRASPQP_HIP_LIB; fi
      python bench.py --no_cpu_baseline --event_steps 0 --n_objects $no | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$no x 256', '$v', round(d['value']), round(d['ms_per_step'],5), 'query_ms', round(d['roofline']['kernel_ms'],5))"
    done
  done
done
