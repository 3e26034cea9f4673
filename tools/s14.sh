#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FNN_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -m gpu -x -q -k "transpose or fused or c4 or c5" > gpurun_out/s14.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s14.log
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done
