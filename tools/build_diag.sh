#!/bin/bash
# Diagnostic build of the HIP library with in-kernel phase timers (never shipped / never loaded by default).
# usage: tools/build_diag.sh [extra -D flags]   ->  attpc_engine_amd/_lib/libattpc_hip_timers.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/attpc_engine_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DATTPC_PHASE_TIMERS "$@" -I"$ROOT/include" -I"$C" \
  -o "$ROOT/attpc_engine_amd/_lib/libattpc_hip_timers.so" "$C/abi.hip" "$C/kinematics.hip" "$C/tracks.hip" "$C/scatter.hip" "$C/scatter_small.hip" "$C/spyral.hip"
echo "built $ROOT/attpc_engine_amd/_lib/libattpc_hip_timers.so"
