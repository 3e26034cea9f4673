#!/bin/bash
cd $GRAFT_REPO_ROOT
export FNN_KNOBS=1
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for b in 8 12 16 24 32; do run --batch $b; done
