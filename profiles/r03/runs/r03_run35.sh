#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
rm -f gpurun_out/r03/ab_trace_block_r03_4_other.log
for extra in "--emulate-rank-of 8" "--emulate-rank-of 4" "--config 3" "--config 1" "--config 2"; do
  echo "== $extra" >> gpurun_out/r03/ab_trace_block_r03_4_other.log
  BENCH_EXTRA="$extra" ROUNDS="1 2" bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_tb1024.so 2>&1 | cut -c1-120 >> gpurun_out/r03/ab_trace_block_r03_4_other.log
done
cat gpurun_out/r03/ab_trace_block_r03_4_other.log
