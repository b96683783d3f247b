#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/ab_primary_persist_auto.sh > gpurun_out/r03/ab_primary_persist_auto.log 2>&1
cut -c1-400 gpurun_out/r03/ab_primary_persist_auto.log
