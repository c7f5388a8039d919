#!/bin/bash
# Measurement aid: build libgrl with -DGRL_TILE_TRACE (every workgroup of the weight-gradient launch records start / end /
# CU / XCD) next to the product library.  Run HERE (hipcc cross-compiles); then on the GPU box:
#   GRL_LIBRARY=deep-rl-grasping_amd/grasp_rl/libgrl_trace.so python scripts/tile_trace.py
# Optional: scripts/tile_trace.sh <suffix> <extra hipcc flags>   ->  libgrl_trace<suffix>.so  (e.g. _a4 -DI2_ABLATE=4)
set -e
cd "$(dirname "$0")/.."
sfx="$1"; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 -DGRL_TILE_TRACE "$@" \
  deep-rl-grasping_amd/csrc/engine.hip -o deep-rl-grasping_amd/grasp_rl/libgrl_trace${sfx}.so
