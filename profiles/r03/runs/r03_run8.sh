#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest8.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest8.log
bash profiles/ab4.sh "RTR_TRACE_VIS_FILL=0" "X=auto" > gpurun_out/r03/ab_vis_prefill.log 2>&1; cat gpurun_out/r03/ab_vis_prefill.log
bash profiles/pmc_one.sh w8 WRITE_SIZE | grep -E "trace4<16, true, false|fillBuffer|gen_oct"
