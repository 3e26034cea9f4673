#!/bin/bash
# usage (this container, after `gpurun -- 'FNN_ROUND=r06 bash tools/capture_all.sh'` has merged its output into gpurun_out/):
#   bash tools/install_capture.sh      - copies the per-workload capture files and the counter-traffic file into profiles/ under the round's names
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${FNN_ROUND:-r06}
for t in bone iso128_r2 iso128_teacher resenc160_r2 resenc160_r2_f8 bone_autocast bone_mirror plan_plane2d_512_r1 plan_brats_128_c4_r1 plan_prostate_20x320x256_r1; do
  d=gpurun_out/cap_$t
  [ -d $d ] || { echo "missing $d"; continue; }
  for f in bench.json kernel_stats.csv trace_summary.txt layers.txt clock_summary.txt pmc_mfma_summary.txt; do
    [ -f $d/$f ] && cp $d/$f profiles/${R}_${t}_$f
  done
done
ls gpurun_out/cap_bone/
cp gpurun_out/final/${R}_* profiles/ 2>/dev/null
ls gpurun_out/final/
cp gpurun_out/${R}_traffic.json profiles/${R}_traffic.json
