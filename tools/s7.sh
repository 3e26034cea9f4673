#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FNN_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "row_streaming or fused" -s > gpurun_out/s7_cfg.log 2>&1; echo "cfg tests rc=$?"; grep -E "row kernels|passed|failed|Error" gpurun_out/s7_cfg.log | head -20
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_predictor.py -m gpu -x -q > gpurun_out/s7_full.log 2>&1; echo "fullsize+predictor rc=$?"; tail -3 gpurun_out/s7_full.log
for nr in 0 1 0 1; do
  if [ $nr = 1 ]; then export FNN_NO_STEM_ROW=1; else unset FNN_NO_STEM_ROW; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_STEM_ROW=$nr', d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done
