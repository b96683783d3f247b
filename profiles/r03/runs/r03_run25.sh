#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
BENCH_EXTRA="--batch 8 --frames-in-flight 8" bash profiles/ab_lib4.sh librtr_hip_b8.so librtr_hip.so > gpurun_out/r03/ab_kernarg_batch16_build.log 2>&1
BENCH_EXTRA="--emulate-rank-of 8 --batch 8 --frames-in-flight 32" bash profiles/ab_lib4.sh librtr_hip_b8.so librtr_hip.so >> gpurun_out/r03/ab_kernarg_batch16_build.log 2>&1
cut -c1-250 gpurun_out/r03/ab_kernarg_batch16_build.log
