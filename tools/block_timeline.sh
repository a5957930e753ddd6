#!/bin/bash
# development build with per-block time stamps (sdf.hip, stage.hip) -> graspqp_amd/lib/libgraspqp_hip_A.so, then
# tools/block_timeline.py (stand-alone query) or tools/block_timeline_stage_a.py (the query as the role of stage A)
set -e
cd "$(dirname "$0")/../graspqp_amd"
for f in sdf stage; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -DGQ_BLOCK_TIMES -c csrc/$f.hip -o /tmp/${f}_bt.o &
done
wait
objs=$(ls lib/*.o | grep -v "/sdf.o\|/stage.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/sdf_bt.o /tmp/stage_bt.o -o lib/libgraspqp_hip_A.so
echo "built lib/libgraspqp_hip_A.so"
