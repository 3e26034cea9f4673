#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s6_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/s6_pytest.log
