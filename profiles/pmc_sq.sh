#!/bin/bash
# SQ instruction-mix / lane-utilisation counters for the bench kernels:  bash profiles/pmc_sq.sh <tag> [extra bench args]
set -e -o pipefail
TAG=${1:-sq}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/bench_pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/bench_pmc_sq2.log" 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for tag in ("pmc_sq", "pmc_sq2"):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"] or 0))
    for k in agg:
        if "rtrdev" not in k: continue
        print(k)
        for c, v in sorted(agg[k].items()):
            print(f"   {c:26s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
