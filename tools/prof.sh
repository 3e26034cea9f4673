#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args...]
# kernel trace of bench.py + per-kernel summary under gpurun_out/prof_<tag>*
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
export FNN_KNOBS=1                     # honour FNN_* A-B switches given on the command line
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py --no-cpu-baseline --no-clock-probe --no-also --no-from-host --steps 3 --warmup 1 "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
f=$(ls gpurun_out/prof_$tag/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -z "$f" ] && f=$(ls gpurun_out/prof_$tag/*kernel_trace.csv | head -1)
python tools/trace_summary.py $f > gpurun_out/prof_${tag}_summary.txt 2>&1
grep -h metric gpurun_out/prof_$tag.log | cut -c1-180
find gpurun_out/prof_$tag -name "*.csv" ! -name "*kernel_stats.csv" -delete
