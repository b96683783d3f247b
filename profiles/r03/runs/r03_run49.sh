#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_limit" > gpurun_out/r03/pytest_run49.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r03/pytest_run49.log | cut -c1-250
