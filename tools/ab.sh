#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/ab.sh libA.so libB.so [rounds]
# Alternates two builds of libfnn_hip.so (paths relative to fast-nnunet_amd/csrc) inside one session: boxes of the pool
# differ by +-3 %, so two builds can only be compared back to back on the same box.
root=${GRAFT_REPO_ROOT:-$(pwd)}
export FNN_KNOBS=1                     # honour FNN_* A-B switches given on the command line
a=$1; b=$2; n=${3:-3}
for i in $(seq 1 $n); do
  for lib in $a $b; do
    FNN_LIB=$root/fast-nnunet_amd/csrc/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
  done
done
