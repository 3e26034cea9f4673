#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FNN_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stride2 or strided" > gpurun_out/s3_ops.log 2>&1; echo "s2 tests rc=$?"; tail -15 gpurun_out/s3_ops.log
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "fused" > gpurun_out/s3_fused.log 2>&1; echo "fused tests rc=$?"; tail -5 gpurun_out/s3_fused.log
for nr in 0 1 0 1; do
  if [ $nr = 1 ]; then export FNN_NO_S2=1; else unset FNN_NO_S2; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_S2=$nr', d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done
unset FNN_NO_S2
bash tools/layers.sh s3 | grep -v "stats_final"
