#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FNN_KNOBS=1
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "row" > gpurun_out/s2_row.log 2>&1; echo "row tests rc=$?"; tail -15 gpurun_out/s2_row.log
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "fused" -s > gpurun_out/s2_fused.log 2>&1; echo "fused tests rc=$?"; tail -15 gpurun_out/s2_fused.log
for nr in 0 1; do
  if [ $nr = 1 ]; then export FNN_NO_ROW=1; else unset FNN_NO_ROW; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_ROW=$nr', d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done
unset FNN_NO_ROW
bash tools/layers.sh s2 | grep -v "stats_final"
