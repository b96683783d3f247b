#!/bin/bash
# Vector-memory path counters (TA / TCP = L1) for the bench kernels:  bash profiles/pmc_tcp.sh <tag> [extra bench args]
set -e -o pipefail
TAG=${1:-tcp}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-in-flight 1 $@"
timeout -k 10 170 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/pmc_tcp1" -- $BENCH > "$OUT/bench_pmc_tcp1.log" 2> "$OUT/rocprof_pmc_tcp1.log" || echo "tcp1 failed"
# (round 1: a pass with TA_BUSY_avr / TA_TA_BUSY_sum / TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum aborted inside
#  rocprofv3 and the bench process then hung until the run's limit; its output was not kept.  Round 3 met the same abort with its
#  evidence kept (profiles/pmc_one.sh keeps every pass's stdout and stderr; gpurun_out/prof_w1/rocprof_w1.log, quoted in
#  profiles/r03/rocprofv3_error38_counter_group_too_large.log): asking for FETCH_SIZE and WRITE_SIZE in ONE pass makes
#  rocprofiler_create_counter_config fail with "error code 38: Request exceeds the capabilities of the hardware to collect" —
#  a counter group that needs more slots of a block than the block has (TCC: 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2; the guide's
#  PMC-slot table) — and the tool answers with abort().  Four TA_* counters in one pass are the same request against the TA block.
#  The rule for every script here: one block's counters per pass, within its slots; a pass that is refused fails in under two
#  seconds, before the bench has touched the GPU, so it is retried with fewer counters, never in a loop.  The TA pass stays out:
#  nothing in the analysis needs it.)
timeout -k 10 170 rocprofv3 --kernel-trace --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d "$OUT/pmc_tcp3" -- $BENCH > "$OUT/bench_pmc_tcp3.log" 2> "$OUT/rocprof_pmc_tcp3.log" || echo "tcp3 failed"
timeout -k 10 170 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/bench_pmc_sq.log" 2> "$OUT/rocprof_pmc_sq.log" || echo "sq failed"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for tag in ("pmc_tcp1", "pmc_tcp2", "pmc_tcp3", "pmc_sq"):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:72]][row["Counter_Name"]].append(float(row["Counter_Value"] or 0))
    for k in agg:
        if "rtrdev" not in k or "k_resolve<true" in k or ", true>(rtrdev::DeviceScene, rtrdev::RenderA" in k or "trace_count" in k: continue   # the counting (STATS) forms
        print(k)
        for c, v in sorted(agg[k].items()):
            print(f"   {c:40s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
