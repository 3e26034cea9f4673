#!/bin/bash
# usage (this container, repo root): bash tools/build_ab_libs.sh   - builds the round-4 and round-5 libraries (commits 3128df3, 16db403) into
# fast-nnunet_amd/csrc/ab/ (git-ignored; they travel to the GPU box with gpurun) for tools/ab_libs.sh.  ~1 minute per library.
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p fast-nnunet_amd/csrc/ab
for pair in "r04 3128df3" "r05 16db403"; do
  set -- $pair
  rm -rf /tmp/fnn_$1; git worktree add /tmp/fnn_$1 $2 -f > /dev/null 2>&1
  make -C /tmp/fnn_$1/fast-nnunet_amd/csrc -j8 > /tmp/fnn_$1_build.log 2>&1
  cp /tmp/fnn_$1/fast-nnunet_amd/csrc/libfnn_hip.so fast-nnunet_amd/csrc/ab/libfnn_$1.so
  git worktree remove /tmp/fnn_$1 --force
done
ls -la fast-nnunet_amd/csrc/ab/
