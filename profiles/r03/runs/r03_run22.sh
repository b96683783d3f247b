#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/ab_lib4.sh librtr_hip_before.so librtr_hip.so > gpurun_out/r03/ab_resolve_answer_first.log 2>&1; cut -c1-230 gpurun_out/r03/ab_resolve_answer_first.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03/pytest_run22.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03/pytest_run22.log
