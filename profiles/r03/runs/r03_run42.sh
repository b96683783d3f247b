#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_length" > gpurun_out/r03/pytest_run42.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r03/pytest_run42.log
