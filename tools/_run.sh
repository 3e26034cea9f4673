cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export FNN_KNOBS=1
timeout 600 python tools/gather_ab.py "FNN_GATHER_WAVE_ROWS=1" "" --check > gpurun_out/r4/gather_ab.txt 2>&1
timeout 600 python tools/gather_ab.py "FNN_GATHER_WAVE_ROWS=1" "" --accum fp16_autocast --check > gpurun_out/r4/gather_ab_autocast.txt 2>&1
for shape in "32 32 32 160 48 48" "32 32 32 160 48 48 32" "32 64 64 80 24 24"; do
  echo "== shape $shape" >> gpurun_out/r4/tmode.txt
  FNN_OP_TIME=1 FNN_LIB=$GRAFT_REPO_ROOT/fast-nnunet_amd/csrc/libfnn_ab.so timeout 300 python tools/zr_tmode.py $shape >> gpurun_out/r4/tmode.txt 2>&1
done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/pytest.txt 2>&1
for arm in "FNN_NO_ZSW=1" "" "FNN_NO_ZSW=1" ""; do
  echo "== arm [$arm]" >> gpurun_out/r4/bench_ab.txt
  env $arm timeout 300 python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['time_share_ms'], j['roofline']['frac'])" >> gpurun_out/r4/bench_ab.txt 2>&1
done
grep -v amdgpu gpurun_out/r4/gather_ab.txt gpurun_out/r4/gather_ab_autocast.txt; grep -v amdgpu gpurun_out/r4/tmode.txt; tail -5 gpurun_out/r4/pytest.txt; cat gpurun_out/r4/bench_ab.txt
