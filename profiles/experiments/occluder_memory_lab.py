#!/usr/bin/env python3
"""CPU-only pricing of an OCCLUDER MEMORY for the shadow rays (round 5; apply profiles/experiments/occluder_memory_oracle.patch to oracle/ first): per (pixel-sample, query) the leaf in which that query's ray was
stopped the last time the slot was rendered; a ray that does not start at its own leaf starts at the remembered leaf, the root on its
stack.  Shadows move little between frames, an any-hit answer does not depend on where the walk starts.  The oracle walks the bench's
scripted camera path: frame f with the memory of frame f - lag (lag = the frames of a launch: the product's frames of one launch are
walked together, so a frame can only remember the launch before), and reports record visits / triangle tests per ray with and without.

    python profiles/experiments/occluder_memory_lab.py [scene] [W] [H] [lag ...]          # default sponza_class 480 270 lags 1 7 10
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from realtimeraytracer_amd import _abi as A, api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 270
    lags = [int(x) for x in sys.argv[4:]] or [1, 7, 10]
    s = getattr(scenes, name)(W, H)
    p = api.make_params(W, H, spp=1, shadow_rays=3, collect_stats=1)
    bvh = api.host_build_bvh_wide(s.desc)
    nq = 3 * sum(int(s.desc.lights[i].numTriangles) for i in range(s.desc.numLights)) + 1
    path = s.camera_path(64) if s.cam_args and s.walk_scale > 0 else [(s.camera, s.cam_pos)] * 64
    ref_img = {}

    def frame(f, mem_in, want_out=True):
        cam, pos = path[f]
        out = np.zeros((H, W, 1, nq), np.int32) if want_out else None
        r = O.render(s.desc, cam, s.scene_info(f, pos), p, bvh=bvh, threads=8, occluders=(mem_in, out, nq))
        img = r.images[A.IMAGE_SHADOWED]
        if f in ref_img:
            assert (img == ref_img[f]).all(), "the image changed with the walk's start: impossible by construction"
        else:
            ref_img[f] = img.copy()
        return r, out

    print(f"{name} {W}x{H}, {nq} queries per pixel-sample")
    print(f"{'frame':>5s} {'memory of':>10s} | {'visits/ray':>10s} {'tests/ray':>9s} | {'occluded: visits':>16s} {'tests':>6s} | {'visible: visits':>15s} {'tests':>6s} | remembered entries")
    def show(f, src, r, mem_in):
        c, k = r.stats, r.walk
        n = c.numShadowRays
        ent = int((mem_in < 0).sum()) if mem_in is not None else 0
        print(f"{f:5d} {src:>10s} | {c.numShadowNodeVisits / n:10.3f} {c.numShadowTriTests / n:9.3f} | {k.occludedVisits / max(k.occludedRays, 1):16.2f} {k.occludedTests / max(k.occludedRays, 1):6.2f} | "
              f"{k.visibleVisits / max(k.visibleRays, 1):15.2f} {k.visibleTests / max(k.visibleRays, 1):6.2f} | {ent} ({ent / n:.2f} per ray)", flush=True)

    base = 20
    for lag in lags:
        r0, mem = frame(base, None)
        show(base, "nothing", r0, None)
        for step in range(1, 4):                      # three launches on: the memory is what the frame `lag` earlier left, itself walked with a memory
            f = base + step * lag
            rn, _ = frame(f, None, want_out=False)
            show(f, "nothing", rn, None)
            r1, mem2 = frame(f, mem)
            show(f, f"frame {f - lag}", r1, mem)
            mem = mem2
    # a static camera: the same view, new light samples every frame
    cam, pos = path[base]
    mem = None
    for f in range(base, base + 3):
        out = np.zeros((H, W, 1, nq), np.int32)
        r = O.render(s.desc, cam, s.scene_info(f, pos), p, bvh=bvh, threads=8, occluders=(mem, out, nq))
        show(f, "static cam", r, mem)
        mem = out


if __name__ == "__main__":
    main()
