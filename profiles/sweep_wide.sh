#!/bin/bash
# knob sweep of the wide any-hit kernel on the bench frame (one frame at a time): profiles/sweep_wide.sh > gpurun_out/r02/sweep_wide.log
for tri in 0 8 16 24 32 40; do for inner in 20 28 36; do
  echo -n "tri_min $tri inner_min $inner : "
  RTR_TRACE_TRI_MIN=$tri RTR_TRACE_INNER_MIN=$inner python profiles/print_stats.py | tail -2 | tr '\n' ' '; echo
done; done
for refill in 8 12 16 24 32; do
  echo -n "refill $refill : "
  RTR_TRACE_REFILL=$refill python profiles/print_stats.py | tail -2 | tr '\n' ' '; echo
done
