#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest7.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest7.log
bash profiles/ab_lib4.sh librtr_hip_q32.so librtr_hip.so > gpurun_out/r03/ab_compact_queue.log 2>&1; cat gpurun_out/r03/ab_compact_queue.log
