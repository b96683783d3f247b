#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/sweep_batches_in_flight.sh > gpurun_out/r03/sweep_batches_in_flight.log 2>&1
cat gpurun_out/r03/sweep_batches_in_flight.log
