#!/bin/bash
# usage (GPU box): bash tools/pmc_zp.sh   - SQ counters of the plane kernels (conv2d_zp / conv2d_zps) inside the 512^2 2-D plan's step:
# MFMA busy, instruction mix, LDS bank conflicts (the image's "conflict free for every tap" claim), three passes each under timeout
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmczp
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $line --output-format csv -d $out/p$i -- python3 $root/bench.py --plan $root/tools/plans/plane2d_512_r1.json --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  for kn in "conv2d_zp_kernel<8, 4>" "conv2d_zp_kernel<8, 2>" "conv2d_zps_kernel<4, 2>"; do
    echo "== $kn: $line"
    [ -n "$f" ] && python3 $root/tools/pmc_avg.py $f "$kn"
  done
  rm -rf $out/p$i
done <<'EOF2'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVES
EOF2
