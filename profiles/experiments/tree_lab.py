#!/usr/bin/env python3
"""CPU-only pricing of a tree change: builds the bench scene with the host builder under a set of RTR_BVH_* settings, hands the BVH2
and the 4-wide view (rtr_host_build_bvh_wide: the records the device would hold) to the oracle and prints what the oracle's walk —
the any-hit kernel's walk, visit for visit — costs per shadow ray on a reduced frame.
    python profiles/experiments/tree_lab.py [scene] [W] [H]            # default sponza_class 480 270
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from realtimeraytracer_amd import api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

VARIANTS = [
    ("the default: probe, passes if it pays (auto)", {}),
    ("binned SAH, greedy collapse (round 2)", {"RTR_BVH_REINSERT_PASSES": "0", "RTR_BVH_WIDE_GREEDY": "1"}),
    ("binned SAH, cost-driven collapse", {"RTR_BVH_REINSERT_PASSES": "0"}),
    ("+ 1 reinsertion pass", {"RTR_BVH_REINSERT_PASSES": "1"}),
    ("+ 2 reinsertion passes", {"RTR_BVH_REINSERT_PASSES": "2"}),
    ("+ 3 reinsertion passes", {"RTR_BVH_REINSERT_PASSES": "3"}),
    ("+ 5 reinsertion passes", {"RTR_BVH_REINSERT_PASSES": "5"}),
]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 270
    only = os.environ.get("LAB_ONLY")
    s = getattr(scenes, {"sponza_class": "sponza_class", "cornell": "cornell_box", "bunny_class": "bunny_class", "sponza_mixed": "sponza_mixed"}[name])(W, H)
    p = api.make_params(W, H, spp=1, shadow_rays=3, collect_stats=1)
    print(f"{name} {W}x{H}")
    print(f"{'variant':44s} {'build s':>8s} {'SAH2':>8s} {'wide cost':>10s} {'depth':>5s} {'nodes':>8s} {'wide':>8s} {'visits/ray':>10s} {'tests/ray':>9s} {'cam visits':>10s} {'cam tests':>9s} {'tail':>5s}")
    for label, env in VARIANTS:
        if only and only not in label:
            continue
        for k in [k for k in os.environ if k.startswith("RTR_BVH_")]:
            del os.environ[k]
        os.environ.update(env)
        t0 = time.perf_counter()
        bvh = api.host_build_bvh_wide(s.desc)
        dt = time.perf_counter() - t0
        st = bvh.stats
        r = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=bvh, threads=8)
        c = r.stats
        print(f"{label:44s} {dt:8.2f} {st.sahCost:8.2f} {'':>10s} {st.maxDepth:5d} {st.numNodes:8d} {st.numWideNodes:8d} "
              f"{c.numShadowNodeVisits / c.numShadowRays:10.3f} {c.numShadowTriTests / c.numShadowRays:9.3f} "
              f"{(c.numNodeVisits - c.numShadowNodeVisits) / c.numPrimaryRays:10.3f} {(c.numTriTests - c.numShadowTriTests) / c.numPrimaryRays:9.3f} {c.shadowTailRays:5d}", flush=True)


if __name__ == "__main__":
    main()
