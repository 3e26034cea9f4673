#!/bin/bash
# usage (GPU box): bash tools/pipes_sweep.sh <rounds>   - batches in flight (FNN_NO_PIPELINE, FNN_PIPES = 2 / 3 / 4) per BASELINE workload,
# arms alternating on one box (the hidden share is 5.2 / 2.9 / 2.3 / 0.7 % for C2 / C5 / C1 / C4)
cd ${GRAFT_REPO_ROOT:-.}
rounds=${1:-2}
for wl in "--workload bone_turbo_r2" "--workload iso128_r2" "--workload iso128_teacher" "--workload resenc160_r2"; do
  for r in $(seq 1 $rounds); do
    for arm in "FNN_NO_PIPELINE=1" "FNN_PIPES=2" "FNN_PIPES=3" "FNN_PIPES=4"; do
      line=$(env FNN_KNOBS=1 $arm timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-also --no-from-host --no-clock-probe --no-roofline $wl 2>/dev/null | grep "^{" | tail -1)
      echo "$wl $arm $(echo "$line" | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], 'patches/s', j['ms_per_step'], 'ms')")"
    done
  done
done
