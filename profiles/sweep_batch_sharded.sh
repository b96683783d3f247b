#!/bin/bash
# Batch size of the any-hit kernel for the short queues of 1/N shards:  bash profiles/sweep_batch_sharded.sh
cd $GRAFT_REPO_ROOT
for n in 8 4 2; do for b in 64 128 256; do
  echo -n "one rank of $n batch $b : "
  RTR_TRACE_BATCH=$b timeout -k 5 120 python bench.py --steps 40 --warmup 6 --emulate-rank-of $n 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'], j['kernels_ms'])"
done; done
