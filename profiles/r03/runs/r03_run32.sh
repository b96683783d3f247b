#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh.py -m gpu -x -q > gpurun_out/r03/pytest_run32.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03/pytest_run32.log
ROUNDS="1 2 3 4" bash profiles/ab_lib4.sh librtr_hip_head.so librtr_hip_norc.so librtr_hip.so > gpurun_out/r03/ab_r03_4.log 2>&1; cut -c1-250 gpurun_out/r03/ab_r03_4.log
