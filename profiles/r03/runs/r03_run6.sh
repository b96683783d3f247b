#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_cohack.so > gpurun_out/r03/ab_coherent_samples_hack.log 2>&1; cat gpurun_out/r03/ab_coherent_samples_hack.log
cp realtimeraytracer_amd/librtr_hip.so /tmp/b.so; cp realtimeraytracer_amd/librtr_hip_cohack.so realtimeraytracer_amd/librtr_hip.so
python profiles/print_stats.py >> gpurun_out/r03/ab_coherent_samples_hack.log 2>&1; tail -6 gpurun_out/r03/ab_coherent_samples_hack.log
cp /tmp/b.so realtimeraytracer_amd/librtr_hip.so
