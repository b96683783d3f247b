#!/bin/bash
# A/B of the camera-ray kernel choice: RTR_PRIMARY_PERSIST=0 (one ray per lane) against 2 (by launch size: persistent waves from 6 M camera rays).
cd ${GRAFT_REPO_ROOT:-.}
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); p=j.get('presented_frame') or {}
        print('$1', 'ms/step', j['ms_per_step'], j['value'], j['unit'], '| presented', p.get('ms_per_frame'), p.get('kernels_ms'), '| one at a time', j.get('one_frame_at_a_time'))
"; }
for rep in 1 2; do
for m in 0 2; do
  RTR_PRIMARY_PERSIST=$m python3 bench.py --steps 16 --warmup 4 --present-frames 10 2>/dev/null | show "[config 4 default + presented] persist=$m"
  RTR_PRIMARY_PERSIST=$m python3 bench.py --config 3 --batch 1 --frames-in-flight 4 --steps 40 --warmup 8 --present-frames 0 2>/dev/null | show "[config 3, one frame per launch, 4 in flight] persist=$m"
  RTR_PRIMARY_PERSIST=$m python3 bench.py --config 3 --batch 1 --frames-in-flight 1 --steps 40 --warmup 8 --present-frames 0 2>/dev/null | show "[config 3, one frame per launch, 1 in flight] persist=$m"
  RTR_PRIMARY_PERSIST=$m python3 bench.py --config 5 --steps 3 --warmup 1 --present-frames 0 2>/dev/null | show "[config 5] persist=$m"
done; done
