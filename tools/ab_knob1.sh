#!/bin/bash
# usage (GPU box): bash tools/ab_knob1.sh "N CIN COUT D H W [cin2]" "ENV=1" "ENV=.." ...  - one layer shape under several knob settings
cd ${GRAFT_REPO_ROOT:-.}
shape=$1; shift
for arm in "$@"; do
  echo -n "[$arm] "
  env $arm timeout 300 python tools/layer_time.py $shape 2>&1 | grep "op time"
done
