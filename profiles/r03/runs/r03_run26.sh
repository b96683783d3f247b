#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or mgpu or flight" > gpurun_out/r03/pytest_run26.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03/pytest_run26.log
bash profiles/sweep_frames_per_launch.sh > gpurun_out/r03/sweep_frames_per_launch.log 2>&1
cut -c1-300 gpurun_out/r03/sweep_frames_per_launch.log
