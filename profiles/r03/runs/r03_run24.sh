#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/sweep_frames_per_launch.sh > gpurun_out/r03/sweep_frames_per_launch.log 2>&1
cut -c1-300 gpurun_out/r03/sweep_frames_per_launch.log
