#!/bin/bash
# mutation check of the multi-rank GPU test: the library built with rank 0 receiving every shard from the wrong peer must turn it red
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "same_device_rccl_double" > gpurun_out/r03/pytest_mutation_wrong_peer.log 2>&1; echo "pytest rc $? (expected: 1)"; grep -E "passed|failed|pixels differing" gpurun_out/r03/pytest_mutation_wrong_peer.log | cut -c1-200 | tail -8
