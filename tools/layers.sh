#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/layers.sh <tag> [bench args...]
# One-stream kernel trace of bench.py, summarised per (kernel, grid) and per LAYER (tools/trace_layers.py).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
export FNN_KNOBS=1 FNN_NO_PIPELINE=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/lay_$tag -- python3 $root/bench.py --no-cpu-baseline --no-roofline --no-also --no-from-host --steps 2 --warmup 1 "$@" > $root/gpurun_out/lay_$tag.log 2>&1
cd $root
f=$(find gpurun_out/lay_$tag -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f > gpurun_out/lay_${tag}_summary.txt 2>&1
python tools/trace_layers.py $f > gpurun_out/lay_${tag}_layers.txt 2>&1
python tools/trace_gaps.py $f >> gpurun_out/lay_${tag}_layers.txt 2>&1
rm -rf gpurun_out/lay_$tag
cat gpurun_out/lay_${tag}_layers.txt
