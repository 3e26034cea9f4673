#!/bin/bash
# usage: gpurun --timeout 3600 -- 'bash tools/final_capture.sh'   - what a round's final capture runs on ONE box (about 30 GPU-minutes):
# the GPU suite, tools/capture_all.sh for every workload (bench line, kernel trace, per-layer table, clocks, MFMA and traffic counters), the five-fold,
# sharded-on-one-rank and driver-shaped bench lines, and the kernel coverage of the suite under rocprofv3.  Output: gpurun_out/final/ and gpurun_out/cap_*/
# (tools/install_capture.sh copies them into profiles/).
cd $GRAFT_REPO_ROOT
export FNN_KNOBS=1
R=${FNN_ROUND:-r06}
mkdir -p gpurun_out/final; rm -f gpurun_out/final/*; rm -rf gpurun_out/cap_*
timeout 2400 python -m pytest tests -m gpu -q -rP > gpurun_out/final/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/final/pytest_gpu.log | tail -2
FNN_ROUND=$R bash tools/capture_all.sh > gpurun_out/final/capture_all_$R.log 2>&1
for t in bone iso128_r2 iso128_teacher resenc160_r2 resenc160_r2_f8 bone_autocast bone_mirror; do cp gpurun_out/cap_$t/traffic.txt gpurun_out/final/${R}_${t}_traffic_by_kernel.txt 2>/dev/null; done
timeout 600 python bench.py --workload iso128_teacher --folds 5 --steps 3 --warmup 1 2> /dev/null | grep "^{" > gpurun_out/final/${R}_iso128_teacher_folds5_bench.json
timeout 600 python bench.py --gpus 1 --force-sharded --steps 5 --warmup 2 --no-cpu-baseline 2> /dev/null | grep "^{" > gpurun_out/final/${R}_bone_sharded1_bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> /dev/null | grep "^{" > gpurun_out/final/${R}_bone_driver_shaped_bench.json
timeout 1500 python tools/plan_sweep.py --out gpurun_out/final/${R}_plan_sweep > gpurun_out/final/${R}_plan_sweep.log 2>&1
root=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && rm -rf $root/gpurun_out/suite_trace && timeout 1800 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/suite_trace -- python3 -m pytest $root/tests -q -m gpu > $root/gpurun_out/final/suite_trace.log 2>&1)
python tools/kernel_coverage.py gpurun_out/suite_trace > gpurun_out/final/${R}_kernel_coverage.txt 2>&1
rm -rf gpurun_out/suite_trace
tail -3 gpurun_out/final/capture_all_$R.log; tail -3 gpurun_out/final/${R}_kernel_coverage.txt
cut -c1-400 gpurun_out/final/${R}_bone_driver_shaped_bench.json
