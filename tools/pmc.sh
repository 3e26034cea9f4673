#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/pmc.sh <tag> "<COUNTER ...>" ["<COUNTER ...>" ...]
# One rocprofv3 --pmc pass per counter group over a one-volume bench (pipelining off), mean per dispatch per kernel.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export FNN_KNOBS=1 FNN_NO_PIPELINE=1
i=0
for grp in "$@"; do
  d=$root/gpurun_out/pmc_${tag}_$i
  rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-also --no-from-host > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 $root/tools/pmc_summary.py $f > $root/gpurun_out/pmc_${tag}_$i.txt 2>&1
  rm -rf $d
  i=$((i+1))
done
