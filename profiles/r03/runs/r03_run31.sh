#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
ROUNDS="1 2 3 4 5" bash profiles/ab_lib4.sh librtr_hip_norc.so librtr_hip.so > gpurun_out/r03/ab_refill_consts_in_lds.log 2>&1; cut -c1-250 gpurun_out/r03/ab_refill_consts_in_lds.log
