#!/usr/bin/env python3
"""One frame at a time, split into K band-shards on K streams (rtr_render_split): wall clock per frame of BASELINE config 4
(Sponza-class, 1920x1080, 1 spp) for K x band height, next to the unsplit frame (K = 1).  The latency figure of bench.py's
`one_frame_at_a_time` is the K = 1 line.  Usage: python profiles/sweep_split.py [frames] ; GPU_MAX_HW_QUEUES is set per child."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(frames):
    import numpy as np
    from realtimeraytracer_amd import api, scenes
    W, H = 1920, 1080
    s = scenes.sponza_class(W, H)
    ctx = api.Context(0)
    scene = api.Scene(ctx, s.desc)
    whole = api.Frame(ctx, W, H)
    api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H), whole)
    want = whole.download()
    out = []
    combos = [(1, 8)] + [(k, b) for k in (2, 3, 4, 6) for b in (8, 64, 0)]
    fr = api.Frame(ctx, W, H)
    for k, band in combos:
        b = band if band else ((H + k - 1) // k + 7) // 8 * 8       # 0: contiguous slabs (one band per part)
        p = api.make_params(W, H, band_rows=b)
        for i in range(3):
            api.render_split(scene, s.camera, s.scene_info(i), p, fr, k)
        ts, kern = [], []
        for i in range(frames):
            t0 = time.perf_counter()
            api.render_split(scene, s.camera, s.scene_info(i), p, fr, k)
            ts.append((time.perf_counter() - t0) * 1e3)
            kern.append(fr.stats().totalMs)
        api.render_split(scene, s.camera, s.scene_info(0), p, fr, k)
        bad = int((fr.download() != want).sum())
        ts.sort(); kern.sort()
        out.append({"parts": k, "band_rows": b, "wall_ms_median": round(ts[len(ts) // 2], 4), "wall_ms_min": round(ts[0], 4),
                    "gpu_ms_median": round(kern[len(kern) // 2], 4), "pixels_differing": bad})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
        for q, prio in (("4", "1"), ("8", "1"), ("8", "0")):
            print(f"# GPU_MAX_HW_QUEUES={q} RTR_SPLIT_PRIORITIES={prio} (1: part k's stream has the k-th highest priority)", flush=True)
            env = dict(os.environ, GPU_MAX_HW_QUEUES=q, RTR_SPLIT_PRIORITIES=prio)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(frames)], env=env, check=False)
