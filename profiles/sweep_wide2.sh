#!/bin/bash
for top in 40 0; do for inner in 24 28 32; do
  echo -n "top $top inner_min $inner : "
  RTR_TRACE_TOP_NODES=$top RTR_TRACE_INNER_MIN=$inner python profiles/print_stats.py | tail -3 | tr '\n' ' '; echo
done; done
python bench.py --steps 50 --warmup 5 --no-cpu-baseline | cut -c1-900
