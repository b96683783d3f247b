#!/bin/bash
# Counter passes behind bench.py's roofline block (round 5):  bash profiles/pmc_r05.sh <tag> [extra bench args]
# The launches of the driver's bench command — moving camera, BATCH frames per launch (default 7: what the 16.7-ms latency bound picks
# on this scene), the frame objects of ONE launch, so launches never overlap and a kernel's counters are its own; given explicitly
# because the probe would time launches under the profiler.  Separate --pmc passes with --kernel-trace only.
# Writes gpurun_out/prof_<tag>/summary.txt, pmc.json and pmc_roofline_entry.json (merge the last into profiles/r05/pmc_roofline.json:
#   PMC_INTO=profiles/r05/pmc_roofline.json python3 profiles/make_pmc_json.py ... — printed at the end).
set -o pipefail
TAG=${1:-r03}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# equal launches of every kernel, one at a time: kernels never overlap, a kernel's counters are its own
B=${BATCH:-7}
BENCH="python3 $REPO/bench.py --steps ${STEPS:-21} --warmup ${WARMUP:-7} --batch $B --frames-in-flight $B --no-cpu-baseline --isolated-frames 0 --present-frames 0 $@"
# every pass keeps its own stdout (the bench line) and stderr (rocprofv3's log): a pass that is refused or aborts leaves its reason behind
pass() { name=$1; shift; timeout -k 10 170 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/bench_$name.log" 2> "$OUT/rocprof_$name.log" || { echo "$name failed"; tail -3 "$OUT/rocprof_$name.log"; }; }
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.log" 2> "$OUT/rocprof_stats.log" || echo "stats failed"
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
# where a wave's cycles go (MI355X_MICROARCH.md: WAIT_ANY = parked at s_waitcnt / a barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; disjoint, they add
# up to WAVE_CYCLES) — the counter-backed answer to "what is k_shadow_gen_oct waiting for" 
pass wait SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - "$OUT" <<'PY' | tee "$OUT/summary_run.log"
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for tag in ("sq", "sq2", "tcp", "wait", "fetch", "write", "l2"):
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
# what make_pmc_json.py needs to know about the launches, from the bench line of the stats pass
j = json.loads([l for l in open(os.path.join(out, "bench_stats.log")) if l.startswith("{")][-1])
r = j["roofline"]
print("MAKE_PMC_ARGS", j["config"]["workload"].split()[0] + "_" + j["config"]["workload"].split()[1] + "_spp1_gpus1", r["kernel"].split("revision ")[-1],
      r["schedule_per_frame"]["shadow_rays"], j["config"]["primary_rays_per_frame"], j["config"]["workload"].split(",")[1].split()[0], j["frames_per_launch"],
      "path" if "scripted walk" in j["config"]["camera"] else "static")
PY
ARGS=$(grep MAKE_PMC_ARGS "$OUT/summary_run.log" 2>/dev/null | tail -1 | cut -d" " -f2-)
if [ -n "$ARGS" ]; then set -- $ARGS; python3 $REPO/profiles/make_pmc_json.py "$OUT/pmc.json" "$1" "$2" "$3" "$4" "$5" "$6" "$7" > "$OUT/pmc_roofline_entry.json" && echo "wrote $OUT/pmc_roofline_entry.json ($ARGS)"; fi
