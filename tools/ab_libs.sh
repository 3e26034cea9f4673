#!/bin/bash
# usage (GPU box): bash tools/ab_libs.sh <rounds> <workload flags...>
# Same-box A-B of library builds on one workload: the round-4 and round-5 libraries (built from commits
# 3128df3 and 16db403 into fast-nnunet_amd/csrc/ab/, git-ignored, travelling with gpurun) against the tree's, arms alternating,
# <rounds> times; bench.py's driver-shaped line per arm (patches/s).  FNN_LIB is honoured next to FNN_KNOBS=1.
cd ${GRAFT_REPO_ROOT:-.}
rounds=$1; shift
ab=fast-nnunet_amd/csrc/ab
for r in $(seq 1 $rounds); do
  for arm in r04 r05 head; do
    lib=$ab/libfnn_$arm.so; [ $arm = head ] && lib=fast-nnunet_amd/csrc/libfnn_hip.so
    [ -f $lib ] || continue
    line=$(FNN_KNOBS=1 FNN_LIB=$(pwd)/$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-from-host --no-clock-probe "$@" 2>/dev/null | grep "^{" | tail -1)
    echo "$arm $(echo "$line" | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], 'patches/s', j['ms_per_step'], 'ms, family frac', j['roofline']['frac'], 'hidden', j['roofline']['schedules']['hidden_by_batches_in_flight'])")  [$*]"
  done
done
