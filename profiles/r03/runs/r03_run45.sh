#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/rehearse_ranks_on_one_gpu.sh > gpurun_out/r03/rehearse_ranks_on_one_gpu.log 2>&1; cat gpurun_out/r03/rehearse_ranks_on_one_gpu.log
