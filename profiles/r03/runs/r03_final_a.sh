#!/bin/bash
# round 3, final measurements (a): GPU tests, counter passes, kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_final.log
bash profiles/pmc_r03.sh r03f > gpurun_out/r03/pmc_r03f.log 2>&1; tail -5 gpurun_out/r03/pmc_r03f.log
cd $GRAFT_REPO_ROOT
bash profiles/stats_r03.sh r03s > gpurun_out/r03/stats_r03s.log 2>&1; tail -3 gpurun_out/r03/stats_r03s.log
