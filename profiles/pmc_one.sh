#!/bin/bash
# One counter group for every kernel of the bench frame:  bash profiles/pmc_one.sh <tag> <counter> [<counter> ...]
# frames one at a time, --kernel-trace + --pmc only; per-kernel means on stdout and in gpurun_out/prof_<tag>/means.txt.
# Every pass keeps its own stdout / stderr (bench_<tag>.log, rocprof_<tag>.log): a failing pass leaves its evidence behind.
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 170 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pass" -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 --isolated-frames 0 --present-frames 0 $BENCH_ARGS > "$OUT/bench_$TAG.log" 2> "$OUT/rocprof_$TAG.log" || { echo "pass $TAG failed"; tail -5 "$OUT/rocprof_$TAG.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"] or 0))
with open(os.path.join(out, "means.txt"), "w") as fh:
    for k in sorted(agg):
        if "rtrdev" not in k: continue
        line = k.split("(")[0].replace("void ", "") + "  " + "  ".join(f"{c}={sum(v) / len(v):.1f} (n={len(v)})" for c, v in sorted(agg[k].items()))
        print(line); print(line, file=fh)
PY
