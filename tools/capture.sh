#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/capture.sh <tag> [bench.py workload flags, e.g. --workload iso128_r2]
# Collects what profiles/ holds per workload: kernel trace + stats + per-layer medians, the two PMC passes for HBM traffic
# (separate runs, no trace domains next to --pmc), the MFMA-busy pass, and the plain bench line.  Every pass under `timeout`.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
round=${FNN_ROUND:-r06}
out=$root/gpurun_out/cap_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
key=$(cd $root && python3 -c "import bench, sys, argparse
ap = argparse.ArgumentParser(); ap.add_argument('--workload', default='bone_turbo_r2'); ap.add_argument('--dtype', default='f16'); ap.add_argument('--mirror', action='store_true'); ap.add_argument('--accum', default='fp16'); ap.add_argument('--volume', type=int, default=512); ap.add_argument('--batch', type=int, default=32)
print(bench.traffic_key(ap.parse_known_args()[0]))" "$@")
echo "workload key: $key"
# per-kernel durations and counters are taken with the pipelining off: kernels of the internal streams otherwise overlap
# and the trace reports their stretched durations.  The headline bench line below has it on.
export FNN_KNOBS=1 FNN_NO_PIPELINE=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu-baseline --no-clock-probe --no-also --no-from-host --steps 3 --warmup 1 "$@" > $out/trace.log 2>&1
if [ -z "$SKIP_PMC" ]; then
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $out/pmc_mfma -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-also --no-from-host "$@" > $out/pmc_mfma.log 2>&1
m=$(find $out/pmc_mfma -name "*counter_collection.csv" | head -1)
mk=$(find $out/pmc_mfma -name "*kernel_trace.csv" | head -1)
python3 $root/tools/pmc_summary.py $m > $out/pmc_mfma_summary.txt 2>&1
python3 $root/tools/clock_summary.py $mk $m > $out/clock_summary.txt 2>&1
rm -rf $out/pmc_mfma
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-also --no-from-host "$@" > $out/pmc_$c.log 2>&1
done
fi
cd $root
kt=$(find $out/trace -name "*kernel_trace.csv" | head -1)
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python tools/trace_summary.py $kt > $out/trace_summary.txt 2>&1
python tools/trace_layers.py $kt > $out/layers.txt 2>&1
cp $ks $out/kernel_stats.csv
if [ -z "$SKIP_PMC" ]; then
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic.py $f $w $root/gpurun_out/${round}_traffic.json "$key" > $out/traffic.txt 2>&1
fi
unset FNN_NO_PIPELINE
cp $root/gpurun_out/${round}_traffic.json $root/profiles/${round}_traffic.json 2>/dev/null   # so that the bench line below quotes it
extra=""; [[ "$*" == *--plan* ]] && extra="--no-cpu-baseline"          # (a plan line: no CPU leg - its oracle forward of a full-width teacher patch takes minutes)
timeout 900 python bench.py $extra "$@" 2> $out/bench.err | grep "^{" > $out/bench.json
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
tail -1 $out/bench.json | cut -c1-1200
head -14 $out/trace_summary.txt
cat $out/clock_summary.txt | head -20
