#!/bin/bash
cd $GRAFT_REPO_ROOT
export FNN_KNOBS=1
for tg in 4 8 4 8; do
FNN_TCONV_TG=$tg python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TG=$tg', d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done
