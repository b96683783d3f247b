#!/bin/bash
# The few counters that say what binds the any-hit kernel, one frame at a time:  bash profiles/pmc_quick.sh <tag> [extra bench args]
# (separate --pmc passes; --kernel-trace only, as the pool requires)
set -o pipefail
TAG=${1:-quick}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-in-flight 1 --isolated-frames 0 --present-frames 0 $@"
timeout -k 10 170 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/bench_pmc_sq.log" 2>&1 || echo "sq failed"
timeout -k 10 170 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/bench_pmc_sq2.log" 2>&1 || echo "sq2 failed"
timeout -k 10 170 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/pmc_tcp1" -- $BENCH > "$OUT/bench_pmc_tcp1.log" 2>&1 || echo "tcp1 failed"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for tag in ("pmc_sq", "pmc_sq2", "pmc_tcp1"):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:80]][row["Counter_Name"]].append(float(row["Counter_Value"] or 0))
    for k in sorted(agg):
        if "rtrdev" not in k: continue
        print(k)
        for c, v in sorted(agg[k].items()):
            print(f"   {c:40s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
