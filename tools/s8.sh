#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FNN_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "row_streaming" -s > gpurun_out/s8_cfg.log 2>&1; echo "cfg tests rc=$?"; grep -E "row kernels|passed|failed|Error" gpurun_out/s8_cfg.log | head -20
bash tools/layers.sh s8 | grep -v "stats_final" | head -8
