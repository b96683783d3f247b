#!/bin/bash
# Counter passes behind bench.py's roofline block (round 3):  bash profiles/pmc_r03.sh <tag> [extra bench args]
# The driver's bench command (--steps 20 --warmup 5: one launch of 20 frames at a time, so a kernel's counters are its own;
# STEPS=64 WARMUP=32 for launches of 32); separate --pmc passes with --kernel-trace only.
# Writes gpurun_out/prof_<tag>/summary.txt and pmc.json (copy both into profiles/r03/).
set -o pipefail
TAG=${1:-r03}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# equal launches of every kernel, one at a time: kernels never overlap, a kernel's counters are its own
BENCH="python3 $REPO/bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --isolated-frames 0 --present-frames 0 $@"
# every pass keeps its own stdout (the bench line) and stderr (rocprofv3's log): a pass that is refused or aborts leaves its reason behind
pass() { name=$1; shift; timeout -k 10 170 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/bench_$name.log" 2> "$OUT/rocprof_$name.log" || { echo "$name failed"; tail -3 "$OUT/rocprof_$name.log"; }; }
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.log" 2> "$OUT/rocprof_stats.log" || echo "stats failed"
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for tag in ("sq", "sq2", "tcp", "fetch", "write", "l2"):
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"] or 0))
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6)
res = {}
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k in sorted(set(agg) | set(dur)):
        if "rtrdev" not in k: continue
        short = k.split("(")[0].replace("void ", "")
        # a run launches every kernel over ONE frame too (each frame object's first render, the counting pass): the means are over
        # the launches of the timed kind only — those within 20 % of the largest value (every counter grows with the work)
        def big(v):
            m = max(v)
            return [x for x in v if x >= 0.8 * m] if m > 0 else v
        e = {c: sum(big(v)) / len(big(v)) for c, v in agg[k].items()}
        if dur[k]: e["avg_ms"] = sum(big(dur[k])) / len(big(dur[k])); e["calls"] = len(big(dur[k]))
        res[short] = e
        print(short, file=fh)
        for c, v in sorted(e.items()): print(f"   {c:34s} {v:18.4f}", file=fh)
json.dump(res, open(os.path.join(out, "pmc.json"), "w"), indent=1, sort_keys=True)
print(open(os.path.join(out, "summary.txt")).read())
PY
