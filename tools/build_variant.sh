#!/bin/bash
# Experimental build of the HIP library with other build-time constants (never shipped / never loaded by default):
#   tools/build_variant.sh NAME [-D flags]   ->  attpc_engine_amd/_lib/libattpc_NAME.so
# e.g. tools/build_variant.sh timers -DATTPC_PHASE_TIMERS ; compare builds on the GPU box with tools/ab_scatter.py
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
C="$ROOT/attpc_engine_amd/csrc"
NAME=$1
shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden "$@" -I"$ROOT/include" -I"$C" \
  -o "$ROOT/attpc_engine_amd/_lib/libattpc_$NAME.so" "$C/abi.hip" "$C/kinematics.hip" "$C/tracks.hip" "$C/scatter.hip" "$C/scatter_small.hip" "$C/scatter_wide.hip" "$C/lone.hip" "$C/spyral.hip" "$C/unpack_host.cpp"
echo "built $ROOT/attpc_engine_amd/_lib/libattpc_$NAME.so"
