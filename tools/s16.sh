#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$*', d['value'], d['ms_per_step'], r.get('achieved'), r.get('frac'))"; }
run --workload iso128_r2
run --workload iso128_teacher
run --workload resenc160_r2
run --workload resenc160_r2 --dtype f8
run --mirror
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --accum fp16_autocast > gpurun_out/bench_autocast.json 2>/dev/null; tail -1 gpurun_out/bench_autocast.json | cut -c1-300
