#!/bin/bash
# A/B of the compute_sdf paths between builds of the library in ONE gpurun call: tools/ab_sdf_libs.sh <dir1> <dir2> ...
for rep in 1 2; do
  for v in "$@"; do
    GRASPQP_HIP_LIB=$PWD/graspqp_amd/$v/libgraspqp_hip.so python tools/plugin_surface.py sdf 2>/dev/null | tail -1
  done
done
