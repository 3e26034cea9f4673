#!/bin/bash
# usage (GPU box): bash tools/pmc_kernel.sh <kernel-name-substring> [bench flags]  - SQ counters of one kernel of the bench step (two passes, each under timeout)
kn=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmck
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --output-format csv -d $out/p$i -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-also --no-from-host "$@" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $root/tools/pmc_avg.py $f "$kn"
  rm -rf $out/p$i
done <<'EOF'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE
EOF
