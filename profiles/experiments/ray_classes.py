#!/usr/bin/env python3
"""CPU-only: the shadow rays of the reduced bench frame by class — started at their own leaf and stopped there / walked on, from the root
and occluded / visible, directional — with each class's share of the record visits and triangle tests, and for the occluded area-light
rays where along the ray the occluder was found (what the farthest-exit-first order was read off: profiles/r05/far_first_lab.log).
Uses the oracle's per-ray walk records (oracle_scene::walkRays).

    python profiles/experiments/ray_classes.py [sponza_class | sponza_mixed | bunny_class]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from realtimeraytracer_amd import _abi as A, api, scenes
from oracle import oracle_py as O
W, H = 480, 270
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
s = getattr(scenes, name)(W, H)
p = api.make_params(W, H, spp=1, shadow_rays=3, collect_stats=1)
bvh = api.host_build_bvh_wide(s.desc)
rays = np.zeros((4_000_000, 8), np.float32); cnt = np.zeros(1, np.uint64)
r = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=bvh, threads=8, walk_rays=(rays, cnt))
n = int(min(cnt[0], len(rays))); R = rays[:n]
vis, tests, occ, own, t, tmax = R[:, 0], R[:, 1], R[:, 2] > 0, R[:, 3] > 0, R[:, 4], R[:, 5]
tot_v, tot_t = vis.sum(), tests.sum()
def cls(mask, label):
    m = mask
    print(f"{label:46s} {m.mean()*100:5.1f} % of rays | visits {vis[m].mean() if m.any() else 0:6.2f} each = {vis[m].sum()/tot_v*100:5.1f} % of all visits | tests {tests[m].mean() if m.any() else 0:5.2f} = {tests[m].sum()/tot_t*100:5.1f} % of all tests")
print(name, n, "shadow rays; visits/ray", vis.mean(), "tests/ray", tests.mean())
direc = tmax > 5000
cls(own & occ & (vis == 0), "own leaf: stopped there")
cls(own & ~(occ & (vis == 0)), "own leaf: not stopped (walked)")
cls(~own & occ & ~direc, "area light, from the root, occluded")
cls(~own & ~occ & ~direc, "area light, from the root, visible")
cls(direc & occ, "directional, occluded")
cls(direc & ~occ, "directional, visible")
m = ~own & occ & ~direc
frac = t[m] / tmax[m]
for lo, hi in ((0, .02), (.02, .1), (.1, .3), (.3, .6), (.6, 1.01)):
    mm = (frac >= lo) & (frac < hi)
    print(f"   occluded area-light rays stopped at t/tmax in [{lo},{hi}): {mm.mean()*100:5.1f} % of them, visits {vis[m][mm].mean():6.2f}, t {t[m][mm].mean():7.1f}")
print("   absolute t of the stop for those rays: percentiles 10/50/90:", np.percentile(t[m], [10, 50, 90]))
