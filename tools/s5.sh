#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FNN_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stride2" > gpurun_out/s5_ops.log 2>&1; echo "s2 tests rc=$?"; tail -3 gpurun_out/s5_ops.log
STRIDE=2 FNN_OP_TIME=1 python tools/stamps.py 32 32 64 160 48 48 2>&1 | grep -i "op time" | head
STRIDE=2 FNN_OP_TIME=1 python tools/stamps.py 32 64 128 80 24 24 2>&1 | grep -i "op time" | head
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
