#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/ab_queue_affinity.sh > gpurun_out/r03/ab_queue_affinity.log 2>&1; cut -c1-260 gpurun_out/r03/ab_queue_affinity.log
