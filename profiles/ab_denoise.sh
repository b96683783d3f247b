#!/bin/bash
# A/B between builds of the library for the post passes: bash profiles/ab_denoise.sh librtr_hip.so librtr_hip_dn_w4.so ...  (profiles/time_denoise.py's first line)
cd $GRAFT_REPO_ROOT/realtimeraytracer_amd
cp librtr_hip.so /tmp/librtr_hip_base.so
for round in 1 2; do for v in "$@"; do
  if [ "$v" = "librtr_hip.so" ]; then cp /tmp/librtr_hip_base.so librtr_hip.so; else cp $v librtr_hip.so; fi
  echo -n "[$round] $v : "
  (cd .. && timeout -k 5 120 python profiles/time_denoise.py 2>/dev/null | head -1)
done; done
cp /tmp/librtr_hip_base.so librtr_hip.so
