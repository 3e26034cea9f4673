# scratch: copy the final capture from gpurun_out/ into profiles/ (round r05)
cd /root/repo
for t in bone iso128_r2 iso128_teacher resenc160_r2 resenc160_r2_f8 bone_autocast bone_mirror; do
  d=gpurun_out/cap_$t
  [ -d $d ] || { echo "missing $d"; continue; }
  for f in bench.json kernel_stats.csv trace_summary.txt layers.txt clock_summary.txt pmc_mfma_summary.txt; do
    [ -f $d/$f ] && cp $d/$f profiles/r05_${t}_$f
  done
done
ls gpurun_out/cap_bone/
cp gpurun_out/final/r05_* profiles/ 2>/dev/null
ls gpurun_out/final/
cp gpurun_out/r05_traffic.json profiles/r05_traffic.json
