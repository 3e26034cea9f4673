#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/capture.sh <tag>
# Collects everything profiles/ holds for a round: kernel trace + stats, the two PMC passes for HBM traffic
# (separate runs, no trace domains next to --pmc), and the plain bench line.
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/cap_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# per-kernel durations and counters are taken with the two-batch pipelining off: kernels of the two internal streams
# otherwise overlap and the trace reports their stretched durations.  The headline bench line below has it on.
export FNN_KNOBS=1 FNN_NO_PIPELINE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/trace.log 2>&1
# MFMA busy / clock: SQ and GRBM counters in a pass of their own
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $out/pmc_mfma -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $out/pmc_mfma.log 2>&1
m=$(find $out/pmc_mfma -name "*counter_collection.csv" | head -1)
python3 $root/tools/pmc_summary.py $m > $out/pmc_mfma_summary.txt 2>&1
rm -rf $out/pmc_mfma
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $out/pmc_$c.log 2>&1
done
cd $root
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python tools/trace_summary.py $kt > $out/trace_summary.txt 2>&1
cp $ks $out/kernel_stats.csv
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic.py $f $w $out/traffic.json > $out/traffic.txt 2>&1
cp $out/traffic.json profiles/${FNN_ROUND:-r02}_traffic.json   # bench.py reads the traffic figure from here
unset FNN_NO_PIPELINE
python bench.py > $out/bench.json 2> $out/bench.err
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
tail -1 $out/bench.json | cut -c1-1500
head -12 $out/trace_summary.txt
cat $out/traffic.txt | head -20
