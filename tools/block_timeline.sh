#!/bin/bash
# development build with per-block time stamps (sdf.hip only) -> graspqp_amd/lib/libgraspqp_hip_A.so, then tools/block_timeline.py
set -e
cd "$(dirname "$0")/../graspqp_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -DGQ_BLOCK_TIMES -c csrc/sdf.hip -o /tmp/sdf_bt.o
objs=$(ls lib/*.o | grep -v "/sdf.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/sdf_bt.o -o lib/libgraspqp_hip_A.so
echo "built lib/libgraspqp_hip_A.so"
