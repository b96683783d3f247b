#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/sweep_frames_per_launch_32.sh > gpurun_out/r03/sweep_frames_per_launch_32.log 2>&1; sort gpurun_out/r03/sweep_frames_per_launch_32.log
