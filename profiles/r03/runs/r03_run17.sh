#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/sweep_knobs_r03.sh > gpurun_out/r03/sweep_knobs_r03_3_batch8.log 2>&1; cat gpurun_out/r03/sweep_knobs_r03_3_batch8.log
bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_tb512.so librtr_hip_tb1024.so > gpurun_out/r03/ab_trace_block_batch8.log 2>&1; cat gpurun_out/r03/ab_trace_block_batch8.log | cut -c1-210
