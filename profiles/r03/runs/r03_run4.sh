#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/ab4.sh "RTR_QUEUE_NT=0" "RTR_QUEUE_NT=1" "RTR_QUEUE_NT=2" "RTR_QUEUE_NT=3" > gpurun_out/r03/ab_nt_split.log 2>&1; cat gpurun_out/r03/ab_nt_split.log
PRESENT=10 bash profiles/ab4.sh "X=0" "RTR_PRIMARY_BLOCK=64" "RTR_PRIMARY_PERM=1" "RTR_PRIMARY_BLOCK=64 RTR_PRIMARY_PERM=1" > gpurun_out/r03/ab_primary_block_perm.log 2>&1; cat gpurun_out/r03/ab_primary_block_perm.log
