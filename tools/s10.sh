#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FNN_KNOBS=1
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for b in 32 40 60 100; do run --batch $b; done
for pz in 1 2 4; do FNN_PIPES=$pz run --batch 32; done
FNN_PIPES=2 run --batch 60
FNN_PIPES=4 run --batch 40
run --batch 32
