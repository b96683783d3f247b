#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python3 - <<'PY' > /tmp/show.py
PY
sed -i "s/(j.get('kernels_ms_in_flight_event_brackets') or {}).get('shadow_trace')/(j.get('kernels_ms_in_flight_event_brackets') or {})/" profiles/ab_lib4.sh
ROUNDS="1 2 3" bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_gen256.so librtr_hip_gen1024.so > gpurun_out/r03/ab_gen_oct_block.log 2>&1; cut -c1-330 gpurun_out/r03/ab_gen_oct_block.log
