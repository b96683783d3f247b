#!/usr/bin/env python3
"""profiles/make_pmc_json.py <gpurun_out/prof_TAG/pmc.json> <workload key> <kernel revision> <rays per frame> <pixel-samples per frame> <triangles> [frames per launch] > profiles/r03/pmc_roofline.json
Turns the per-kernel counter means of profiles/pmc_r03.sh into the per-launch figures bench.py reads: the any-hit kernel's issue /
L1 / HBM counters (its roofline block), and the HBM bytes of every kernel of the frame (roofline_secondary for the queue-build
kernel, frame_hbm), with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section) applied and spelled out:
FETCH_SIZE tallies a wide coalesced streaming read at 1/2 of its bytes, everything else at the bytes moved; WRITE_SIZE is exact."""
import json
import sys

src, key, rev, rays, nps, ntri = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
frames = int(sys.argv[7]) if len(sys.argv) > 7 else 1          # frames per launch of the pipeline (bench.py --batch)
camera = sys.argv[8] if len(sys.argv) > 8 else "static"         # bench.py --camera: "path" = a new view every frame (rays: the mean over the timed frames)
rays, nps = rays * frames, nps * frames
if frames > 1:
    key += f"_batch{frames}"
if camera != "static":
    key += "_" + camera
pmc = json.load(open(src))


def pick(*needles):
    for k in pmc:
        if all(n in k for n in needles) and ", true>" not in k.replace("<16, true, false>", "").replace("<16, false, false>", ""):
            return k, pmc[k]
    raise SystemExit(f"no kernel matching {needles} in {src}")


name, k = pick("k_shadow_trace4<16, true, false>") if any("k_shadow_trace4<16, true, false>" in x for x in pmc) else pick("k_shadow_trace4<16, false, false>")
# the counting form and the set-up renders launch the same kernels over ONE frame: the means must be over the batched launches only;
# pmc_r03.sh therefore keeps per-kernel means of the launches whose duration is within 2x of the longest (see its python part)


def hbm(entry, stream_read_bytes):
    """reads: the streamed part (known size) shows in FETCH_SIZE at half its bytes, the rest at the bytes moved; writes exact"""
    fetch, write = entry.get("FETCH_SIZE", 0.0) * 1024, entry.get("WRITE_SIZE", 0.0) * 1024
    rest = max(fetch - stream_read_bytes / 2, 0.0)
    return {"read_bytes": int(stream_read_bytes + rest), "write_bytes": int(write), "FETCH_SIZE_KB": entry.get("FETCH_SIZE"), "WRITE_SIZE_KB": entry.get("WRITE_SIZE"),
            "streamed_read_bytes": int(stream_read_bytes), "rocprof_avg_ms": entry.get("avg_ms"), "rocprof_calls": entry.get("calls")}


queue_bytes, hit_bytes = rays * 20, nps * 20          # the queue's streamed part: 16-B (direction, tmax) + 4-B slot per ray; the 16-B origins are gathered through the caches
kernels = {}
for short, needles, stream in (("k_primary", ("k_primary<", "false"), 0), ("k_shadow_gen_oct", ("k_shadow_gen_oct",), hit_bytes),
                               ("k_shadow_trace4", (name,), queue_bytes), ("k_resolve", ("k_resolve_compact",), hit_bytes), ("k_resolve", ("k_resolve<false",), hit_bytes)):
    if short in kernels:
        continue
    try:
        kn, e = pick(*needles)
    except SystemExit:
        continue
    kernels[short] = dict(hbm(e, stream), kernel=kn)
frame_bytes = sum(v["read_bytes"] + v["write_bytes"] for v in kernels.values())
t = kernels["k_shadow_trace4"]
out = {
    "_comment": "per-launch counters from rocprofv3 --pmc passes (profiles/pmc_r03.sh: --kernel-trace only, one counter group per pass, frames "
                "rendered one at a time); read by bench.py for its roofline, roofline_secondary and frame_hbm blocks, which check workload, "
                "kernel revision, triangle count and queue length against the run before using them",
    key: {
        "kernel": name, "kernel_revision": rev, "rays_per_launch": rays, "pixel_samples": nps, "triangles": ntri, "frames_per_launch": frames, "camera": camera,
        "rocprof_avg_launch_ms": k["avg_ms"], "rocprof_calls": k["calls"],
        "SQ_WAVES": k["SQ_WAVES"], "SQ_WAVE_CYCLES_quad": k["SQ_WAVE_CYCLES"],
        "SQ_INSTS_VALU": k["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU_quad": k["SQ_ACTIVE_INST_VALU"],
        "SQ_THREAD_CYCLES_VALU": k["SQ_THREAD_CYCLES_VALU"], "SQ_INSTS_SALU": k["SQ_INSTS_SALU"],
        "SQ_INSTS_VMEM_RD": k["SQ_INSTS_VMEM_RD"], "SQ_INSTS_LDS": k["SQ_INSTS_LDS"],
        "GRBM_GUI_ACTIVE_sum_over_xcds": k["GRBM_GUI_ACTIVE"],
        "TCP_TOTAL_CACHE_ACCESSES": k["TCP_TOTAL_CACHE_ACCESSES_sum"], "TCP_TCC_READ_REQ": k["TCP_TCC_READ_REQ_sum"],
        "TCC_HIT": k["TCC_HIT_sum"], "TCC_MISS": k["TCC_MISS_sum"],
        "FETCH_SIZE_KB": k["FETCH_SIZE"], "WRITE_SIZE_KB": k["WRITE_SIZE"],
        "hbm_bytes_per_launch": t["read_bytes"] + t["write_bytes"],
        "hbm_derivation": f"reads: ray queue {rays} x 20 B = {queue_bytes / 1e6:.1f} MB (a coalesced stream: FETCH_SIZE shows 1/2 of it) + the remaining "
                          f"{(t['read_bytes'] - queue_bytes) / 1e6:.1f} MB of node / triangle lines that missed L2 (counted at the bytes moved) = {t['read_bytes'] / 1e6:.1f} MB; "
                          f"writes: WRITE_SIZE = {t['write_bytes'] / 1e6:.1f} MB for {rays / 1e6:.1f} MB of visibility bytes (query-major planes: a wave's 64 results are 64 consecutive bytes; "
                          f"the lanes of a wave retire in several refill passes and every pass's bytes leave the L2 as 32-B sectors)",
        "kernels": kernels,
        "frame_hbm_bytes": frame_bytes,
        "derived": {
            "shader_cycles_per_wave": k["SQ_WAVE_CYCLES"] * 4 / k["SQ_WAVES"],
            "clock_mhz_from_wave_cycles": k["SQ_WAVE_CYCLES"] * 4 / k["SQ_WAVES"] / (k["avg_ms"] * 1e3),
            "valu_wave_insts_per_ray": k["SQ_INSTS_VALU"] / rays,
            "valu_lane_utilisation": k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64),
            "l2_hit_rate": k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]),
        },
    },
}
into = __import__("os").environ.get("PMC_INTO")            # merge the key into an existing file instead of printing a new one
if into:
    have = json.load(open(into))
    have[key] = out[key]
    json.dump(have, open(into, "w"), indent=1)
    print(f"{into}: {key} written")
else:
    json.dump(out, sys.stdout, indent=1)
    print()
