#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/sweep_batch_r03_4.sh > gpurun_out/r03/sweep_batch_r03_4.log 2>&1; sort gpurun_out/r03/sweep_batch_r03_4.log
