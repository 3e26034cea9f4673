#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s12_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error" gpurun_out/s12_pytest.log | tail -5
export FNN_KNOBS=1
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done
