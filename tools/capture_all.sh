#!/bin/bash
# usage (GPU box, repo root): FNN_ROUND=r04 bash tools/capture_all.sh   - every workload DESIGN.md quotes, each with tools/capture.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
run() { tag=$1; shift; echo "######## $tag: $*"; bash tools/capture.sh $tag "$@" 2>&1 | tail -28; }
run bone
run iso128_r2 --workload iso128_r2
run iso128_teacher --workload iso128_teacher
run resenc160_r2 --workload resenc160_r2
run resenc160_r2_f8 --workload resenc160_r2 --dtype f8
run bone_autocast --accum fp16_autocast
run bone_mirror --mirror
# round 6: three plans of the sweep with a kernel trace of their own (no counter passes): the 2-D plan, the 4-channel teacher, the thick-slice teacher
SKIP_PMC=1 run plan_plane2d_512_r1 --plan tools/plans/plane2d_512_r1.json
SKIP_PMC=1 run plan_brats_128_c4_r1 --plan tools/plans/brats_128_c4_r1.json
SKIP_PMC=1 run plan_prostate_20x320x256_r1 --plan tools/plans/prostate_20x320x256_r1.json
