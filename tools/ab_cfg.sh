#!/bin/bash
# A/B of lib A vs current lib on the large-batch configurations (one box)
for cfg in "--hand allegro --n_objects 8 --batch_size 256 --steps 200 --warmup 24" "--hand shadow_hand --n_objects 8 --batch_size 512 --n_contact 16 --steps 100 --warmup 16" "--hand robotiq3 --n_objects 32 --batch_size 1024 --n_cone_vecs 8 --steps 24 --warmup 8"; do
  for v in A B A B; do
    if [ $v = A ]; then export GRASPQP_HIP_LIB=$PWD/graspqp_amd/lib/libgraspqp_hip_A.so; else unset GRASPQP_HIP_LIB; fi
    python bench.py --no_cpu_baseline --event_steps 0 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step'],4), '$cfg')"
  done
done
