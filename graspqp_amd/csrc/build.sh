#!/bin/bash
# Build libgraspqp_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${GQ_OUT_DIR:-$HERE/../lib}"  # GQ_OUT_DIR + GQ_EXTRA_FLAGS: variant builds for A/B runs (GRASPQP_HIP_LIB picks one)
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function"
OBJS=()
PIDS=()
# a changed flag set invalidates every object (hash of the flags kept beside them)
SIG="$(echo "$FLAGS ${GQ_EXTRA_FLAGS:-}" | md5sum | cut -d' ' -f1)"
[ "$(cat "$OUT/.flags" 2>/dev/null || true)" = "$SIG" ] || rm -f "$OUT"/*.o
echo "$SIG" > "$OUT/.flags"
deps() {  # headers each translation unit includes
  case "$1" in
    qp_lr|fcstep) echo "common.h qp_core.h qp_kernels.h qp_lr.h wave.h fc_dev.h fcstep_dev.h" ;;
    qp_dense) echo "common.h qp_core.h qp_kernels.h qp_lr.h wave.h" ;;
    stage) echo "common.h qp_core.h qp_kernels.h qp_lr.h wave.h fc_dev.h fcstep_dev.h pen_dev.h tri.h kin_dev.h metric_dev.h" ;;
    qp|qp_nz*) echo "common.h qp_core.h qp_kernels.h" ;;
    sdf) echo "common.h tri.h pen_dev.h sdf_dev.h wave.h" ;;
    bvh) echo "common.h tri.h" ;;
    fc) echo "common.h fc_dev.h wave.h" ;;
    loop) echo "common.h fc_dev.h loop_dev.h wave.h" ;;
    metric) echo "common.h wave.h metric_dev.h" ;;
    init) echo "common.h wave.h" ;;
    export) echo "common.h kin_dev.h wave.h" ;;
    kin) echo "common.h loop_dev.h sdf_dev.h tri.h kin_dev.h wave.h" ;;
    *) echo "common.h" ;;
  esac
}
for f in api qp qp_lr qp_dense qp_nz16 qp_nz32 qp_nz48 qp_nz64 sdf bvh kin fc fcstep stage loop export init metric terms; do
  stale=0
  [ -f "$OUT/$f.o" ] || stale=1
  for d in $f.hip $(deps $f); do [ "$HERE/$d" -nt "$OUT/$f.o" ] && stale=1; done
  if [ $stale = 1 ]; then
    echo "[build] hipcc $f.hip"
    rm -f "$OUT/$f.o"  # a failed compile must not leave a stale object for the link
    "$HIPCC" $FLAGS ${GQ_EXTRA_FLAGS:-} -c "$HERE/$f.hip" -o "$OUT/$f.o" &
    PIDS+=($!)
  fi
  OBJS+=("$OUT/$f.o")
done
for p in "${PIDS[@]:-}"; do [ -z "$p" ] || wait "$p" || { echo "[build] a compile failed" >&2; exit 1; }; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o "$OUT/libgraspqp_hip.so"
echo "[build] $OUT/libgraspqp_hip.so"
