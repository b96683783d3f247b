#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/sweep_refill_r03_4.sh > gpurun_out/r03/sweep_refill_r03_4.log 2>&1; cat gpurun_out/r03/sweep_refill_r03_4.log
