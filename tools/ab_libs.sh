#!/bin/bash
# A/B of several builds of the library in ONE gpurun call (same box): tools/ab_libs.sh <dir1> <dir2> ... ("lib" = the default build)
# every dir is graspqp_amd/<dir>/libgraspqp_hip.so (variant builds: GQ_OUT_DIR=... GQ_EXTRA_FLAGS=... bash graspqp_amd/csrc/build.sh)
for rep in 1 2 3; do
  for v in "$@"; do
    export GRASPQP_HIP_LIB=$PWD/graspqp_amd/$v/libgraspqp_hip.so
    python bench.py --steps 400 --warmup 40 --no_cpu_baseline --plugin_surface 0 --event_steps 0 $AB_FLAGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step'],5), 'query_ms', round(d['roofline']['kernel_ms'],5), 'E', round(d['mean_energy'],4))"
  done
done
