#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
ROUNDS="1 2 3" bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_tb512.so librtr_hip_tb1024.so > gpurun_out/r03/ab_trace_block_r03_4.log 2>&1; cut -c1-250 gpurun_out/r03/ab_trace_block_r03_4.log
