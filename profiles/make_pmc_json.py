#!/usr/bin/env python3
"""profiles/make_pmc_json.py <gpurun_out/prof_TAG/pmc.json> <workload key> <kernel revision> <rays per launch> > profiles/r02/pmc_roofline.json
Turns the per-kernel counter means of profiles/pmc_r02.sh into the per-launch figures bench.py's roofline block reads, with the
gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section) applied and spelled out."""
import json
import sys

src, key, rev, rays = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
pmc = json.load(open(src))
name = [k for k in pmc if "k_shadow_trace4<16, true, false>" in k or "k_shadow_trace4<16, false, false>" in k]
k = pmc[name[0]]
queue_bytes = rays * 32                          # the ray queue: a wide coalesced stream, which FETCH_SIZE tallies at 1/2 (guide; calibrated in profiles/r01/fetch_calibration.txt)
fetch_kb, write_kb = k["FETCH_SIZE"], k["WRITE_SIZE"]
rest_kb = fetch_kb - queue_bytes / 2 / 1024      # node / triangle lines that missed L2 + queue lines fetched twice: counted at the bytes moved
read_bytes = queue_bytes + rest_kb * 1024
out = {
    "_comment": "per-launch counters of the any-hit kernel from rocprofv3 --pmc passes (profiles/pmc_r02.sh: --kernel-trace only, one "
                "pass per counter group, frames rendered one at a time); read by bench.py for its roofline block",
    key: {
        "kernel": name[0], "kernel_revision": rev, "rays_per_launch": rays,
        "rocprof_avg_launch_ms": k["avg_ms"], "rocprof_calls": k["calls"],
        "SQ_WAVES": k["SQ_WAVES"], "SQ_WAVE_CYCLES_quad": k["SQ_WAVE_CYCLES"],
        "SQ_INSTS_VALU": k["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU_quad": k["SQ_ACTIVE_INST_VALU"],
        "SQ_THREAD_CYCLES_VALU": k["SQ_THREAD_CYCLES_VALU"], "SQ_INSTS_SALU": k["SQ_INSTS_SALU"],
        "SQ_INSTS_VMEM_RD": k["SQ_INSTS_VMEM_RD"], "SQ_INSTS_LDS": k["SQ_INSTS_LDS"],
        "GRBM_GUI_ACTIVE_sum_over_xcds": k["GRBM_GUI_ACTIVE"],
        "TCP_TOTAL_CACHE_ACCESSES": k["TCP_TOTAL_CACHE_ACCESSES_sum"], "TCP_TCC_READ_REQ": k["TCP_TCC_READ_REQ_sum"],
        "TCC_HIT": k["TCC_HIT_sum"], "TCC_MISS": k["TCC_MISS_sum"],
        "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
        "hbm_bytes_per_launch": int(read_bytes + write_kb * 1024),
        "hbm_derivation": f"reads: ray queue {rays} x 32 B = {queue_bytes / 1e6:.1f} MB (a coalesced stream: FETCH_SIZE shows 1/2 of it = {queue_bytes / 2048:.0f} KB) "
                          f"+ remaining {rest_kb:.0f} KB of node / triangle lines that missed L2 (counted at the bytes moved) = {read_bytes / 1e6:.1f} MB; "
                          f"writes: WRITE_SIZE {write_kb:.0f} KB = {write_kb * 1024 / 1e6:.1f} MB (one visibility byte per ray, partial lines)",
        "derived": {
            "shader_cycles_per_wave": k["SQ_WAVE_CYCLES"] * 4 / k["SQ_WAVES"],
            "clock_mhz_from_wave_cycles": k["SQ_WAVE_CYCLES"] * 4 / k["SQ_WAVES"] / (k["avg_ms"] * 1e3),
            "valu_wave_insts_per_ray": k["SQ_INSTS_VALU"] / rays,
            "valu_lane_utilisation": k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64),
            "l2_hit_rate": k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]),
        },
    },
}
json.dump(out, sys.stdout, indent=1)
print()
