#!/usr/bin/env python3
"""Design experiment (CPU only): dump the host-built BVH2 of a bench scene and a statistically faithful sample of its shadow
rays, for profiles/experiments/wide_sim.cpp (visit / box-test / triangle-test counts of candidate wide-node layouts).
The rays follow the shape of raygen.rgen:165-241,289-313 (3 samples per light triangle + the directional ray, origin lifted
0.01 along the normal, tmax = distance - 0.5) with numpy's RNG instead of the PCG hash: this is for counting, not parity.
    python profiles/experiments/dump_rays.py sponza_class 480 270 /tmp/wsim
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from realtimeraytracer_amd import api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def main():
    name, W, H, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.makedirs(out, exist_ok=True)
    s = getattr(scenes, {"sponza_class": "sponza_class", "cornell": "cornell_box", "bunny_class": "bunny_class"}[name])(W, H)
    st, nodes, tris = api.host_build_bvh(s.desc)
    nodes_np = np.frombuffer(nodes, dtype=np.uint8).reshape(-1, 32).copy()
    tris_np = np.frombuffer(tris, dtype=np.float32).reshape(-1, 12).copy()
    tri_u32 = tris_np.view(np.uint32)
    grid = np.array(list(st.grid.origin) + [0] + list(st.grid.scale) + [0], dtype=np.float32)
    p = api.make_params(W, H, spp=1, shadow_rays=3)
    t, u, v, cu, pr = O.primary_hits(s.desc, s.camera, p, bvh=(nodes, tris, st.grid), threads=8)
    nl = s.num_lights
    custom, prim = tri_u32[:, 3], tri_u32[:, 7]
    key = custom.astype(np.uint64) << np.uint64(32) | prim.astype(np.uint64)
    order = np.argsort(key)
    hit = (cu != 0xFFFFFFFF) & (cu >= nl)
    hk = cu[hit].astype(np.uint64) << np.uint64(32) | pr[hit].astype(np.uint64)
    ti = order[np.searchsorted(key[order], hk)]
    v0, e1, e2 = tris_np[ti, 0:3], tris_np[ti, 4:7], tris_np[ti, 8:11]
    P = v0 + e1 * u[hit, None] + e2 * v[hit, None]
    cam = np.array(s.cam_pos, dtype=np.float32)
    N = np.cross(e1, e2)
    N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-30)
    N[np.einsum("ij,ij->i", N, P - cam) > 0] *= -1
    org = P + N * 0.01
    rng = np.random.default_rng(1)
    rays, groups, bundles = [], [], []
    pix = np.arange(len(P), dtype=np.uint32)
    lt = np.nonzero(custom < nl)[0]
    for k in lt:
        A, B, Cc = tris_np[k, 0:3], tris_np[k, 0:3] + tris_np[k, 4:7], tris_np[k, 0:3] + tris_np[k, 8:11]
        ln = np.cross(Cc - B, A - B)
        ln /= np.linalg.norm(ln)
        front = np.einsum("j,ij->i", ln, P - A) >= 0
        for _ in range(3):
            r1, r2 = rng.random(len(P), dtype=np.float32), rng.random(len(P), dtype=np.float32)
            f = r1 + r2 > 1
            r1[f], r2[f] = 1 - r1[f], 1 - r2[f]
            L = A + (B - A) * r1[:, None] + (Cc - A) * r2[:, None]
            d = L - P
            dist = np.linalg.norm(d, axis=1)
            d /= dist[:, None]
            rays.append(np.concatenate([org[front], d[front], (dist[front] - 0.5)[:, None]], axis=1))
            groups.append(np.stack([pix[front], np.full(front.sum(), custom[k], np.uint32)], axis=1))
            bundles.append(np.stack([pix[front], np.full(front.sum(), k, np.uint32)], axis=1))       # the 3 rays one surface point sends at ONE light triangle: bundle_sim.cpp
    dl = np.array([-1.0, 1.0, -0.5], dtype=np.float32)
    dl /= np.linalg.norm(dl)
    m = N @ dl > 0
    rays.append(np.concatenate([org[m], np.broadcast_to(dl, (m.sum(), 3)), np.full((m.sum(), 1), 10000.0, np.float32)], axis=1))
    groups.append(np.stack([pix[m], np.full(m.sum(), 0xFFFF, np.uint32)], axis=1))
    bundles.append(np.stack([pix[m], np.full(m.sum(), 0xFFFFFFFF, np.uint32)], axis=1))
    rays = np.concatenate(rays).astype(np.float32)
    groups = np.concatenate(groups).astype(np.uint32)
    keep = rays[:, 6] > 0.001
    rays, groups = rays[keep], groups[keep]
    np.concatenate(bundles).astype(np.uint32)[keep].tofile(os.path.join(out, "bundles.bin"))    # per ray: surface point, light TRIANGLE (0xFFFFFFFF = the directional light)
    groups.tofile(os.path.join(out, "groups.bin"))        # per ray: surface point, light (0xFFFF = the directional light) — shadow_cache_sim.cpp
    nodes_np.tofile(os.path.join(out, "nodes.bin"))
    tris_np.tofile(os.path.join(out, "tris.bin"))
    grid.tofile(os.path.join(out, "grid.bin"))
    rays.tofile(os.path.join(out, "rays.bin"))
    print(f"{name} {W}x{H}: {len(nodes_np)} nodes, {len(tris_np)} tris, {hit.sum()} surface hits, {len(rays)} shadow rays "
          f"({len(rays) / max(hit.sum(), 1):.2f} per hit)")


if __name__ == "__main__":
    main()
