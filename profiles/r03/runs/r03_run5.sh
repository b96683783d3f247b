#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest5.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r03/pytest5.log
bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_pairs.so > gpurun_out/r03/ab_tri_pairs.log 2>&1; cat gpurun_out/r03/ab_tri_pairs.log
cp realtimeraytracer_amd/librtr_hip_pairs.so /tmp/p.so; cp realtimeraytracer_amd/librtr_hip.so /tmp/b.so; cp /tmp/p.so realtimeraytracer_amd/librtr_hip.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r03/pytest5_pairs.log 2>&1; echo "pytest(pairs) rc=$?"; tail -2 gpurun_out/r03/pytest5_pairs.log
cp /tmp/b.so realtimeraytracer_amd/librtr_hip.so
