#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
ROUNDS="1 2 3" bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_st14.so librtr_hip_st12.so > gpurun_out/r03/ab_stack_vs_tree_top.log 2>&1; cut -c1-250 gpurun_out/r03/ab_stack_vs_tree_top.log
