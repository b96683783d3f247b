#!/usr/bin/env python3
"""CPU-only pricing of the order a shadow ray enters the children of a 4-wide record in (round 5), with the oracle's counters on a reduced
frame: nearest entry first (rounds 1-4, the closest-hit order), farthest entry, nearest exit, farthest EXIT first (what the product does
now), and the last with the stacked children sorted as well.  Own-leaf start on in every line.  Visibility cannot change.
    python profiles/experiments/far_first_lab.py [scene ...]            # default sponza_class sponza_mixed bunny_class cornell_box, 480 x 270
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from realtimeraytracer_amd import _abi as A, api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

W, H = 480, 270
for name in sys.argv[1:] or ["sponza_class", "sponza_mixed", "bunny_class", "cornell_box"]:
    s = getattr(scenes, name)(W, H)
    p = api.make_params(W, H, spp=1, shadow_rays=3, collect_stats=1)
    bvh = api.host_build_bvh_wide(s.desc)
    ref = None
    for walk, label in ((4, "nearest entry first (rounds 1-4)"), (8, "farthest entry first"), (12, "nearest exit first"), (0, "FARTHEST EXIT first (the product)"), (16, "farthest exit, the rest sorted too")):
        r = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=bvh, threads=8, shadow_walk=walk)
        if ref is None:
            ref = r.images[A.IMAGE_SHADOWED].copy()
        assert (r.images[A.IMAGE_SHADOWED] == ref).all(), "visibility changed: impossible by construction"
        c, k = r.stats, r.walk
        print(f"{name:13s} {label:36s}: visits/ray {c.numShadowNodeVisits / c.numShadowRays:7.3f} tests/ray {c.numShadowTriTests / c.numShadowRays:6.3f} | occluded ({k.occludedVisits / max(k.occludedRays, 1):6.2f} visits "
              f"{k.occludedTests / max(k.occludedRays, 1):5.2f} tests) visible ({k.visibleVisits / max(k.visibleRays, 1):6.2f}, {k.visibleTests / max(k.visibleRays, 1):5.2f}) | rays redone by the tail kernel {c.shadowTailRays}", flush=True)
