#!/usr/bin/env python3
"""CPU-only pricing of an ANY-HIT traversal order for the shadow rays.

The any-hit kernel enters the NEAREST hit child of a 4-wide record first — the closest-hit order — and pays a fifth of a visit's vector
cycles for choosing it.  Three shadow rays in four are occluded: what matters is how soon a walk meets AN occluder.  Here the slots of
every record are put in a static order at build time (several estimates of "where is an occluder met soonest", below) and the walk takes
the first hit slot in slot order, the others popping in slot order (oracle walk rule 1: oracle.h oracle_scene::shadowWalk).  The oracle
walks the reduced bench frame's real shadow rays over the permuted records and reports visits / triangle tests per ray, split by the
ray's answer.  Visibility cannot change (any-hit is a pure function of ray and triangles); the script asserts the image is the same.

    python profiles/experiments/anyhit_order_lab.py [scene] [W] [H]            # default sponza_class 480 270
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from realtimeraytracer_amd import _abi as A, api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

EMPTY = -2147483648


def half_to_float(h):
    return np.asarray(h, dtype=np.uint16).view(np.float16).astype(np.float64)


def slot_metrics(w, tris, scale):
    """w: (N,16) uint32 wide records.  Returns dict of (N,4) arrays."""
    N = len(w)
    planes = w[:, :12].reshape(N, 4, 3)
    child = w[:, 12:16].view(np.int32)
    xmin, ymin = half_to_float(planes[:, :, 0] & 0xffff), half_to_float(planes[:, :, 0] >> 16)
    xmax, ymax = half_to_float(planes[:, :, 1] & 0xffff), half_to_float(planes[:, :, 1] >> 16)
    zmin, zmax = half_to_float(planes[:, :, 2] & 0xffff), half_to_float(planes[:, :, 2] >> 16)
    empty = child == EMPTY
    dx = np.where(empty, 0, (xmax - xmin) * scale[0]); dy = np.where(empty, 0, (ymax - ymin) * scale[1]); dz = np.where(empty, 0, (zmax - zmin) * scale[2])
    area = 2.0 * (dx * dy + dy * dz + dz * dx)                   # whole surface of the slot's box
    leaf = (child < 0) & ~empty
    code = ~child
    first, cnt = code >> 3, (code & 7) + 1
    tri_area = 0.5 * np.linalg.norm(np.cross(tris[:, 4:7].astype(np.float64), tris[:, 8:11].astype(np.float64)), axis=1)
    cum = np.concatenate([[0.0], np.cumsum(tri_area)])
    # per record: own box = union of its slots
    bx0 = np.where(empty, np.inf, xmin).min(1); bx1 = np.where(empty, -np.inf, xmax).max(1)
    by0 = np.where(empty, np.inf, ymin).min(1); by1 = np.where(empty, -np.inf, ymax).max(1)
    bz0 = np.where(empty, np.inf, zmin).min(1); bz1 = np.where(empty, -np.inf, zmax).max(1)
    ox, oy, oz = (bx1 - bx0) * scale[0], (by1 - by0) * scale[1], (bz1 - bz0) * scale[2]
    own = 2.0 * (ox * oy + oy * oz + oz * ox)
    T = np.zeros((N, 4)); ntri = np.zeros((N, 4)); cost = np.zeros((N, 4))
    recT = np.zeros(N); recN = np.zeros(N); recC = np.zeros(N)
    CI = 1.2                                                     # a triangle test against a record visit (77 against 64 vector instructions)
    for i in range(N - 1, -1, -1):                               # breadth-first order: children have larger indices
        for k in range(4):
            c = child[i, k]
            if c == EMPTY:
                continue
            if c < 0:
                f, n = first[i, k], cnt[i, k]
                T[i, k] = cum[f + n] - cum[f]; ntri[i, k] = n; cost[i, k] = n * CI
            else:
                T[i, k] = recT[c]; ntri[i, k] = recN[c]; cost[i, k] = recC[c]
        recT[i] = T[i].sum(); recN[i] = ntri[i].sum()
        # expected cost of a walk that enters this record: the visit + each child weighted by the chance a line through the record's box meets the child's
        recC[i] = 1.0 + float((np.minimum(area[i] / max(own[i], 1e-30), 1.0) * cost[i]).sum())
    return dict(area=area, leaf=leaf, empty=empty, T=T, ntri=ntri, cost=cost, child=child)


def permute(w, key):
    """slots of every record in descending key order, empty slots last (key = -inf)."""
    order = np.argsort(-key, axis=1, kind="stable")
    N = len(w)
    planes = w[:, :12].reshape(N, 4, 3)
    child = w[:, 12:16]
    out = np.empty_like(w)
    out[:, :12] = np.take_along_axis(planes, order[:, :, None], axis=1).reshape(N, 12)
    out[:, 12:16] = np.take_along_axis(child, order, axis=1)
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 270
    s = getattr(scenes, {"sponza_class": "sponza_class", "cornell": "cornell_box", "bunny_class": "bunny_class", "sponza_mixed": "sponza_mixed"}[name])(W, H)
    p = api.make_params(W, H, spp=1, shadow_rays=3, collect_stats=1)
    bvh = api.host_build_bvh_wide(s.desc)
    w0 = np.frombuffer(bvh.wide, dtype=np.uint32).reshape(-1, 16).copy()
    tris = np.frombuffer(bvh[1], dtype=np.float32).reshape(-1, 12)
    m = slot_metrics(w0, tris, list(bvh.stats.grid.scale))
    ninf = np.where(m["empty"], -np.inf, 0.0)
    A_, T, cost = m["area"], m["T"], m["cost"]
    q = 1.0 - np.exp(-2.0 * T / np.maximum(A_, 1e-30))          # chance that a line through the box meets one of its triangles (Poisson, 2 T / A crossings)
    keys = [
        ("as built", None),
        ("largest box first", A_),
        ("smallest box first", -A_),
        ("leaves first, then largest", m["leaf"] * 1e30 + A_),
        ("most triangle area first", T),
        ("densest first (T / A)", T / np.maximum(A_, 1e-30)),
        ("q / cost first", q / np.maximum(cost, 1e-30)),
        ("q first", q),
        ("cheapest subtree first", -cost),
        ("most triangles first", m["ntri"]),
    ]
    print(f"{name} {W}x{H}: {len(w0)} wide records")
    print(f"{'slot order':34s} {'walk':>8s} {'visits/ray':>10s} {'tests/ray':>9s} | {'occluded':>8s} {'visits':>7s} {'tests':>6s} | {'visible':>8s} {'visits':>7s} {'tests':>6s} | tail")
    ref_img = None

    def run(label, w, walk, profile=None, view=None):
        nonlocal ref_img
        arr = (A.RtrWideNode * len(w)).from_buffer_copy(np.ascontiguousarray(w).tobytes())
        b = api.BvhExport((bvh[0], bvh[1], bvh[2]))
        b.wide = arr
        cam, info = (s.camera, s.scene_info(0)) if view is None else view
        r = O.render(s.desc, cam, info, p, bvh=b, threads=8, shadow_walk=walk, walk_profile=profile)
        img = r.images[A.IMAGE_SHADOWED]
        if view is None:
            if ref_img is None:
                ref_img = img.copy()
            assert (img == ref_img).all(), "visibility changed with the order: impossible by construction"
        c, k = r.stats, r.walk
        print(f"{label:34s} {'nearest' if walk == 0 else 'slot':>8s} {c.numShadowNodeVisits / c.numShadowRays:10.3f} {c.numShadowTriTests / c.numShadowRays:9.3f} | "
              f"{k.occludedRays / max(c.numShadowRays, 1):8.3f} {k.occludedVisits / max(k.occludedRays, 1):7.2f} {k.occludedTests / max(k.occludedRays, 1):6.2f} | "
              f"{k.visibleRays / max(c.numShadowRays, 1):8.3f} {k.visibleVisits / max(k.visibleRays, 1):7.2f} {k.visibleTests / max(k.visibleRays, 1):6.2f} | {c.shadowTailRays}", flush=True)

    run("as built", w0, 0)
    for label, key in keys:
        w = w0 if key is None else permute(w0, key + ninf)
        run(label, w, 1)
        if key is not None and label in ("largest box first", "q / cost first"):
            run(label, w, 0)                                       # the nearest-first walk over the same records: ties aside, the order must not matter to it


    # ---- the trained order: profile the walk (entries, work below, occluders found below, per slot), order by found / work ----
    def slot_order(key):
        return np.argsort(-key, axis=1, kind="stable")

    def apply(w, order):
        N = len(w)
        out = np.empty_like(w)
        out[:, :12] = np.take_along_axis(w[:, :12].reshape(N, 4, 3), order[:, :, None], axis=1).reshape(N, 12)
        out[:, 12:16] = np.take_along_axis(w[:, 12:16], order, axis=1)
        return out

    for weight in (2.0, 0.25):
        w, walk, pq, pc = w0, 0, q.copy(), np.maximum(cost, 1e-3)
        for it in range(3):
            prof = np.zeros((len(w0), 4, 3), dtype=np.uint64)
            run(f"  (profiling pass {it}, prior weight {weight})", w, walk, prof)
            found, work = prof[:, :, 2].astype(np.float64), prof[:, :, 1].astype(np.float64)
            key = (found + weight * pq) / (work + weight * pc) + np.where(w[:, 12:16].view(np.int32) == EMPTY, -np.inf, 0.0)
            order = slot_order(key)
            w, pq, pc, walk = apply(w, order), np.take_along_axis(pq, order, axis=1), np.take_along_axis(pc, order, axis=1), 1
            run(f"trained on this frame, round {it + 1}", w, 1)


    # ---- does a trained order carry over to other views?  Train on views the bench never renders, evaluate on the bench view ----
    if name.startswith("sponza"):
        from realtimeraytracer_amd import host
        fov, pos, look, up = s.cam_args
        others = [((900.0, 300.0, -60.0), (-600.0, 420.0, 40.0)), ((0.0, 700.0, 0.0), (-900.0, 100.0, 200.0)), ((-200.0, 250.0, 150.0), (700.0, 500.0, -100.0))]
        views = []
        for (ps, lk) in others:
            c = host.Camera(fov, ps, lk, up, W, H)
            views.append((c.getGPUData(), host.scene_info(7, s.num_lights, ps)))
        w, walk, pq, pc = w0, 0, q.copy(), np.maximum(cost, 1e-3)
        for it in range(3):
            prof = np.zeros((len(w0), 4, 3), dtype=np.uint64)
            for v in views:
                run(f"  (profiling pass {it}, another view)", w, walk, prof, view=v)
            found, work = prof[:, :, 2].astype(np.float64), prof[:, :, 1].astype(np.float64)
            key = (found + 2.0 * pq) / (work + 2.0 * pc) + np.where(w[:, 12:16].view(np.int32) == EMPTY, -np.inf, 0.0)
            order = slot_order(key)
            w, pq, pc, walk = apply(w, order), np.take_along_axis(pq, order, axis=1), np.take_along_axis(pc, order, axis=1), 1
            run(f"trained on 3 OTHER views, round {it + 1}", w, 1)


if __name__ == "__main__":
    main()
