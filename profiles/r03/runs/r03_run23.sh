#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_run23.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest_run23.log
