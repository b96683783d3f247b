#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "textured or cornell_256 or sponza_class_parity or batched" > gpurun_out/r03/pytest15.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03/pytest15.log
bash profiles/ab_lib4.sh librtr_hip_head.so librtr_hip.so > gpurun_out/r03/ab_alpha_ptrs.log 2>&1; cat gpurun_out/r03/ab_alpha_ptrs.log | cut -c1-200
