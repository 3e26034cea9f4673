#!/bin/bash
cd $GRAFT_REPO_ROOT
export FNN_KNOBS=1
for m in 768 480 768 480; do
FNN_ZR_MIN_WGS=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MIN_WGS=$m', d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done
FNN_ZR_MIN_WGS=480 bash tools/layers.sh s18 | grep -v "stats_final" | sed -n 10,20p
