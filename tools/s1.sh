#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/s1_pytest.log
bash tools/layers.sh s1
export FNN_KNOBS=1
for i in 1 2; do
for td in 8 4; do
  FNN_ZR_TD=$td python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TD=$td', d['value'], d['ms_per_step'], d['roofline']['time_share_ms'])"
done; done
