#!/bin/bash
# Measurement aid: build libgrl with -DGRL_TILE_TRACE (every workgroup of the weight-gradient launch records start / end /
# CU / XCD) next to the product library.  Run HERE (hipcc cross-compiles); then on the GPU box:
#   GRL_LIBRARY=deep-rl-grasping_amd/grasp_rl/libgrl_trace.so python scripts/tile_trace.py
# Optional: scripts/tile_trace.sh <suffix> <extra hipcc flags>   ->  libgrl_trace<suffix>.so  (e.g. _a4 -DI2_ABLATE=4)
set -e
cd "$(dirname "$0")/.."
sfx="$1"; shift || true
objs=""
for u in deep-rl-grasping_amd/csrc/*.hip; do
  o=build/obj/$(basename ${u%.hip}).trace${sfx}.o
  mkdir -p build/obj
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 -DGRL_TILE_TRACE "$@" -c $u -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o deep-rl-grasping_amd/grasp_rl/libgrl_trace${sfx}.so
