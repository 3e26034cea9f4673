#!/bin/bash
# usage (GPU box): bash tools/ab_layers.sh libA.so libB.so   - per-layer launch times of the benchmark's ZR layer shapes for two builds
cd ${GRAFT_REPO_ROOT:-.}
for lib in "$@"; do
  for shape in "32 32 32 160 48 48" "32 32 32 160 48 48 32" "32 64 64 80 24 24" "32 64 64 80 24 24 64"; do
    FNN_LIB=$PWD/fast-nnunet_amd/csrc/$lib timeout 300 python tools/layer_time.py $shape 2>&1 | grep "op time"
  done
done
