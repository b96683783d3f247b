"""The 4-wide view on a CPU-only box (rtr_host_build_bvh_wide: the host restatement of the device kernels that make it) and the two
builder additions of round 3 — the cost-driven collapse and the insertion-based optimisation: structural invariants, the oracle's
wide walk against the O(N) brute-force loop, and that neither addition changes a pixel.  The GPU suite checks the host records
against the device's byte for byte (tests/test_gpu_bvh.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from realtimeraytracer_amd import _abi as A
from realtimeraytracer_amd import api, scenes

from test_oracle_bvh import _check_bvh


def _half(bits):
    return np.asarray(bits, np.uint16).view(np.float16).astype(np.float64)


def _wide_arrays(wide):
    w = np.frombuffer(wide, dtype=np.uint32).reshape(-1, 16)
    planes = w[:, :12].reshape(-1, 4, 3)
    child = w[:, 12:16].view(np.int32)
    return planes, child


def _check_wide(bvh):
    """every triangle in exactly one leaf of the wide tree, every record reached once, child boxes inside their parent's, breadth-first order"""
    nodes, tris, grid = bvh
    planes, child = _wide_arrays(bvh.wide)
    n = len(child)
    ntri = bvh.stats.numTriangles
    seen_tri = np.zeros(max(ntri, 1), int)
    seen_node = np.zeros(n, int)
    lo = np.stack([_half(planes[:, :, 0] & 0xffff), _half(planes[:, :, 0] >> 16), _half(planes[:, :, 2] & 0xffff)], -1)      # [node, slot, axis], grid steps about the wide centre
    hi = np.stack([_half(planes[:, :, 1] & 0xffff), _half(planes[:, :, 1] >> 16), _half(planes[:, :, 2] >> 16)], -1)
    order = [0]
    seen_node[0] = 1
    stack = [(0, None, None)]
    while stack:
        i, plo, phi = stack.pop()
        used = 0
        for k in range(4):
            c = int(child[i, k])
            if c == -2 ** 31:                                   # RTR_WIDE_EMPTY: inside-out infinite box, only in slots 2, 3
                assert k >= 2 and np.all(np.isinf(lo[i, k])) and np.all(lo[i, k] > 0) and np.all(hi[i, k] < 0)
                continue
            used += 1
            assert np.all(lo[i, k] <= hi[i, k])
            if plo is not None:
                assert np.all(lo[i, k] >= plo) and np.all(hi[i, k] <= phi), "a child box sticks out of its parent's"
            if c >= 0:
                assert c > i and seen_node[c] == 0, "breadth-first order: children come after their parent, every record reached once"
                seen_node[c] += 1
                stack.append((c, lo[i, k], hi[i, k]))
            else:
                code = ~c & 0xffffffff
                first, cnt = code >> 3, (code & 7) + 1
                if not (i == 0 and k == 1 and int(child[0, 0]) == c):
                    seen_tri[first:first + cnt] += 1
        assert used >= 2
    assert np.all(seen_node == 1)
    if ntri:
        assert np.all(seen_tri[:ntri] == 1), "every triangle in exactly one leaf of the 4-wide tree"
    # breadth-first numbering: ids assigned in the order records are met level by level
    nxt = 1
    for i in range(n):
        for k in range(4):
            c = int(child[i, k])
            if c >= 0:
                assert c == nxt
                nxt += 1
    assert nxt == n


@pytest.mark.parametrize("scene", ["cornell", "bunny"])
def test_host_wide_view_invariants(scene, scene_cache, monkeypatch):
    s = scenes.cornell_box(64, 64) if scene == "cornell" else scenes.bunny_class(64, 64, subdiv=4)
    for greedy in ("0", "1"):
        monkeypatch.setenv("RTR_BVH_WIDE_GREEDY", greedy)
        bvh = api.host_build_bvh_wide(s.desc)
        assert bvh.stats.wideLayoutVersion > 0 and bvh.stats.numWideNodes == len(bvh.wide) >= 1
        _check_wide(bvh)
    # deterministic, and the plain build reports the same tree
    monkeypatch.delenv("RTR_BVH_WIDE_GREEDY")
    a, b = api.host_build_bvh_wide(s.desc), api.host_build_bvh_wide(s.desc)
    assert bytes(a.wide) == bytes(b.wide) and bytes(a[0]) == bytes(b[0])
    st, nodes, tris = api.host_build_bvh(s.desc)
    assert bytes(nodes) == bytes(a[0]) and bytes(tris) == bytes(a[1])


def test_cost_driven_collapse_needs_fewer_records_than_the_greedy_rule(scene_cache, monkeypatch):
    s = scenes.bunny_class(64, 64, subdiv=4)
    monkeypatch.setenv("RTR_BVH_WIDE_GREEDY", "1")
    g = api.host_build_bvh_wide(s.desc)
    monkeypatch.setenv("RTR_BVH_WIDE_GREEDY", "0")
    c = api.host_build_bvh_wide(s.desc)
    assert bytes(g[0]) == bytes(c[0])                       # the same BVH2 underneath
    assert c.stats.numWideNodes < g.stats.numWideNodes


def test_oracle_wide_walk_equals_brute_force_on_cpu(oracle, scene_cache, monkeypatch):
    """shadow rays over the host-made 4-wide view: same image as the O(N) loop, for both collapses and with the tree optimised"""
    s = scenes.bunny_class(96, 64, subdiv=3)
    p = api.make_params(96, 64, spp=1, shadow_rays=3, collect_stats=1)
    brute = oracle.render(s.desc, s.camera, s.scene_info(2), p, bvh=None, threads=8)
    visits = {}
    for label, env in (("greedy", {"RTR_BVH_WIDE_GREEDY": "1"}), ("cost", {}), ("reinsert", {"RTR_BVH_REINSERT_PASSES": "2"})):
        for k in ("RTR_BVH_WIDE_GREEDY", "RTR_BVH_REINSERT_PASSES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        bvh = api.host_build_bvh_wide(s.desc)
        r = oracle.render(s.desc, s.camera, s.scene_info(2), p, bvh=bvh, threads=8)
        assert np.array_equal(r.images[A.IMAGE_SHADOWED], brute.images[A.IMAGE_SHADOWED]), label
        assert r.stats.numShadowRays == brute.stats.numShadowRays
        visits[label] = r.stats.numShadowNodeVisits
        assert 0 < r.stats.numShadowNodeVisits < brute.stats.numShadowTriTests
    assert len(set(visits.values())) >= 2                   # the variants really are different trees / views


def test_reinsertion_keeps_a_valid_tree(scene_cache, monkeypatch):
    s = scenes.bunny_class(64, 64, subdiv=4)
    st0, n0, t0 = api.host_build_bvh(s.desc)
    monkeypatch.setenv("RTR_BVH_REINSERT_PASSES", "3")
    st, nodes, tris = api.host_build_bvh(s.desc)
    assert bytes(nodes) != bytes(n0)                        # it did something
    _check_bvh(s.desc, st, nodes, tris)
    assert st.numTriangles == st0.numTriangles and st.maxDepth <= 64
    assert st.sahCost <= st0.sahCost * 1.05                 # never much worse; on mixed-size assets better
    _check_wide(api.host_build_bvh_wide(s.desc))


@pytest.mark.parametrize("name", ["sponza_class", "bunny_class", "cornell_box"])
def test_shadow_rays_into_the_surface_start_at_their_own_leaf(name):
    """Round 5: a shadow ray that leaves its surface point INTO the surface (dot(hitNormal, direction) < 0) tests the leaf of the triangle
    it starts on first, then walks from the root (oracle trace_wide's firstLeaf = k_shadow_trace4's refill).  It cannot change a pixel —
    an any-hit answer does not depend on the order triangles are met in — it only makes the walk shorter: on the Sponza-class frame four
    rays in nine start that way and the record visits per ray fall by a quarter."""
    from oracle import oracle_py as O
    W, H = 160, 96
    s = getattr(scenes, name)(W, H)
    p = api.make_params(W, H, spp=1, shadow_rays=3, collect_stats=1)
    bvh = api.host_build_bvh_wide(s.desc)
    on = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=bvh, threads=8)
    off = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=bvh, threads=8, own_leaf=False)
    brute = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=None, threads=8)
    assert np.array_equal(on.images[A.IMAGE_SHADOWED], off.images[A.IMAGE_SHADOWED]) and np.array_equal(on.images[A.IMAGE_SHADOWED], brute.images[A.IMAGE_SHADOWED])
    assert on.stats.numShadowRays == off.stats.numShadowRays == brute.stats.numShadowRays and on.stats.numRays == off.stats.numRays
    assert off.walk.ownLeafRays == 0 and 0 < on.walk.ownLeafRays < on.stats.numShadowRays
    assert on.walk.occludedRays == off.walk.occludedRays and on.walk.visibleRays == off.walk.visibleRays
    assert on.stats.numShadowNodeVisits < off.stats.numShadowNodeVisits
    if name == "sponza_class":
        assert on.walk.ownLeafRays > 0.35 * on.stats.numShadowRays and on.stats.numShadowNodeVisits < 0.8 * off.stats.numShadowNodeVisits
    # a visible ray is never made cheaper by it (every box it hits must still be opened), only dearer by its own leaf's triangles
    assert on.walk.visibleVisits == off.walk.visibleVisits and on.walk.visibleTests >= off.walk.visibleTests


@pytest.mark.parametrize("name", ["sponza_class", "bunny_class", "cornell_box"])
def test_the_child_a_shadow_ray_enters_first_changes_the_work_never_the_answer(name):
    """Round 5: at a 4-wide record a shadow ray enters the hit child whose EXIT distance is the greatest (k_shadow_trace4's inner_nodes4 =
    oracle trace_wide, shadowWalk order 0); rounds 1-4 entered the nearest ENTRY (a library built with -DRTR_SHADOW_FAR_FIRST=0 = the oracle's
    nearest_first).  Any-hit is a pure function of ray and triangles: every order gives the brute-force image, a VISIBLE ray opens the
    same boxes under every order, and on the atrium — occluders towards the lights — the occluded rays' walks are a third shorter."""
    from oracle import oracle_py as O
    W, H = 160, 96
    s = getattr(scenes, name)(W, H)
    p = api.make_params(W, H, spp=1, shadow_rays=3, collect_stats=1)
    bvh = api.host_build_bvh_wide(s.desc)
    brute = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=None, threads=8)
    runs = {}
    for label, kw in (("far_exit", {}), ("near_entry", {"nearest_first": True}), ("far_entry", {"shadow_walk": 8}), ("near_exit", {"shadow_walk": 12})):
        r = O.render(s.desc, s.camera, s.scene_info(0), p, bvh=bvh, threads=8, **kw)
        assert np.array_equal(r.images[A.IMAGE_SHADOWED], brute.images[A.IMAGE_SHADOWED]), label
        assert r.stats.numShadowRays == brute.stats.numShadowRays and r.walk.occludedRays + r.walk.visibleRays == r.stats.numShadowRays, label
        runs[label] = r
    ref = runs["far_exit"]
    for label, r in runs.items():
        assert r.walk.occludedRays == ref.walk.occludedRays and r.walk.visibleRays == ref.walk.visibleRays, label
        assert r.walk.visibleVisits == ref.walk.visibleVisits, label          # nothing in the way: every box the ray meets is opened, in whatever order
        assert r.walk.ownLeafRays == ref.walk.ownLeafRays and r.walk.ownLeafStopped == ref.walk.ownLeafStopped, label
    if name == "sponza_class":
        assert ref.walk.occludedVisits < 0.75 * runs["near_entry"].walk.occludedVisits
        assert ref.stats.numShadowTriTests < 0.9 * runs["near_entry"].stats.numShadowTriTests
        assert ref.stats.numShadowNodeVisits <= min(r.stats.numShadowNodeVisits for r in runs.values())
