#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/pmc_r03.sh r03_4_b10 > gpurun_out/r03/pmc_r03_4_b10.log 2>&1; echo "pmc rc $?"
grep -A32 "^rtrdev::k_shadow_trace4<16, true, false>" gpurun_out/prof_r03_4_b10/summary.txt
