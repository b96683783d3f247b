#!/bin/bash
# Run-time knobs of the any-hit kernel on the bench frame, one frame at a time (bench.py --frames-in-flight 1):
#   bash profiles/sweep_knobs_r02.sh > gpurun_out/r02/sweep_knobs.log
run() { python bench.py --steps 40 --warmup 4 --no-cpu-baseline --frames-in-flight 1 --isolated-frames 0 --present-frames 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels_ms']; print('trace %.4f ms  frame %.4f ms  primary %.3f gen %.3f resolve %.3f' % (k['shadow_trace'], j['ms_per_frame'], k['primary'], k['shadow_gen'], k['resolve']))"; }
for inner in 20 24 28 32 36; do echo -n "inner_min $inner : "; RTR_TRACE_INNER_MIN=$inner run; done
for refill in 12 16 20 24 28; do echo -n "refill $refill : "; RTR_TRACE_REFILL=$refill run; done
for top in 0 20 40; do echo -n "top_nodes $top : "; RTR_TRACE_TOP_NODES=$top run; done
for batch in 128 256 512; do echo -n "batch $batch : "; RTR_TRACE_BATCH=$batch run; done
