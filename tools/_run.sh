cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 600 python tools/gather_ab.py "FNN_GATHER_V1=1" "" "FNN_GATHER_PF=0" "FNN_GATHER_K32=1" "FNN_GATHER_K32=1 FNN_GATHER_PF=0" --check > gpurun_out/r2/gather_ab.txt 2>&1
timeout 600 python tools/gather_ab.py "FNN_GATHER_V1=1" "" "FNN_GATHER_PF=0" "FNN_GATHER_K32=1" "FNN_GATHER_K32=1 FNN_GATHER_PF=0" --accum fp16_autocast --check > gpurun_out/r2/gather_ab_autocast.txt 2>&1
timeout 600 python tools/gather_ab.py "FNN_GATHER_V1=1" "" "FNN_GATHER_PF=0" --labels --check > gpurun_out/r2/gather_ab_labels.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2/bench.txt 2>&1
cat gpurun_out/r2/gather_ab.txt gpurun_out/r2/gather_ab_autocast.txt gpurun_out/r2/gather_ab_labels.txt | grep -v amdgpu.ids; tail -5 gpurun_out/r2/pytest.txt; tail -1 gpurun_out/r2/bench.txt | cut -c1-300
