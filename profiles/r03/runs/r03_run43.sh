#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "same_device_rccl_double" > gpurun_out/r03/pytest_run43.log 2>&1; echo "pytest rc $?"; tail -30 gpurun_out/r03/pytest_run43.log | cut -c1-300
