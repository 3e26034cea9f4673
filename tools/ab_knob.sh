#!/bin/bash
# usage (GPU box): bash tools/ab_knob.sh "ENV=1 ENV2=x" "" ...   - per-layer launch times under knob settings (one quoted string per arm)
cd ${GRAFT_REPO_ROOT:-.}
for arm in "$@"; do
  echo "== arm: $arm"
  for shape in "32 32 32 160 48 48" "32 32 32 160 48 48 32" "32 64 64 80 24 24" "32 64 64 80 24 24 64"; do
    env $arm timeout 300 python tools/layer_time.py $shape 2>&1 | grep "op time"
  done
done
