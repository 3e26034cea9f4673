#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FNN_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "zr" > gpurun_out/s13_ops.log 2>&1; echo "zr tests rc=$?"; tail -3 gpurun_out/s13_ops.log
for nr in 0 1 0 1; do
  if [ $nr = 1 ]; then export FNN_NO_ZRP=1; else unset FNN_NO_ZRP; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_ZRP=$nr', d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done
unset FNN_NO_ZRP
bash tools/layers.sh s13 | grep -v "stats_final" | grep "zr\|sum"
