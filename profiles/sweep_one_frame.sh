#!/bin/bash
# A frame rendered ALONE (one launch per frame, joined before the next): run-time knobs of the any-hit kernel once more, for this case only.
run() { env "$@" python3 bench.py --steps 24 --warmup 6 --batch 1 --frames-in-flight 1 --isolated-frames 24 --no-cpu-baseline --present-frames 0 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '| one frame at a time', j['one_frame_at_a_time']['ms_per_step'], '| ms/step', j['ms_per_step'], '| kernels', j['kernels_ms'])"; }
run RTR_TRACE_BATCH=0
run RTR_TRACE_BATCH=128
run RTR_TRACE_BATCH=192
run RTR_TRACE_BATCH=384
run RTR_TRACE_REFILL=12
run RTR_TRACE_REFILL=28
run RTR_TRACE_INNER_MIN=24
run RTR_TRACE_INNER_MIN=32
run RTR_TRACE_BINNED=0
run RTR_TRACE_TOP_NODES=20
run RTR_TRACE_BATCH=0
