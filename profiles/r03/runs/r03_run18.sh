#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_resocc.so librtr_hip_resvis.so 2>&1 | head -6 > gpurun_out/r03/resolve_floor_ceiling.log; cat gpurun_out/r03/resolve_floor_ceiling.log | cut -c1-210
