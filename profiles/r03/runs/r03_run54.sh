#!/bin/bash
# perf-only A/B (the oracle does not follow these builds): exit widening folded into the ray constants, a 12-entry stack with a 104-record tree top, both
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
ROUNDS="1 2 3" STEPS=64 WARMUP=32 bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_fold.so librtr_hip_st12.so librtr_hip_fold_st12.so > gpurun_out/r03/ab_fold_and_stack12.log 2>&1; cut -c1-240 gpurun_out/r03/ab_fold_and_stack12.log
