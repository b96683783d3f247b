"""BVH build and refit on the device (kernels/rtr_bvh.hip) — invariants of the LBVH, image equality
with the host-SAH tree (results do not depend on the tree), oracle parity with the device-built tree, and refit after
instance transforms change (TLAS::updateTransform/refit counterpart)."""
import ctypes as C
import os

import numpy as np
import pytest

from realtimeraytracer_amd import _abi as A
from realtimeraytracer_amd import api, scenes
from test_oracle_bvh import _check_bvh

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("queue_mode")]


def _with_flags(desc, flags):
    d = A.rtr_scene_desc.from_buffer_copy(bytes(desc))
    d.buildFlags = flags
    return d


def _render(ctx, scene, setup, p, frame_no=0):
    rows = api.shard_rows(p.height, p.bandRows or 8, p.shardCount or 1)
    frame = api.Frame(ctx, p.width, rows, p.images or A.IMAGES_FRAMEBUFFER)
    api.render(scene, setup.camera, setup.scene_info(frame_no), p, frame)
    return frame


@pytest.mark.parametrize("which", ["cornell", "bunny", "sponza", "room"])
def test_device_lbvh_build(gpu_ctx, oracle, scene_cache, which):
    W, H = 320, 184
    s = {"cornell": lambda: scenes.cornell_box(W, H), "bunny": lambda: scenes.bunny_class(W, H, subdiv=5),
         "sponza": lambda: scenes.sponza_class(W, H), "room": lambda: scenes.textured_room(W, H)}[which]()
    sah = api.Scene(gpu_ctx, s.desc)
    lbvh = api.Scene(gpu_ctx, _with_flags(s.desc, A.BUILD_DEVICE_LBVH))
    st = lbvh.stats()
    assert st.numTriangles == sah.stats().numTriangles
    if st.numTriangles >= 16:                        # tiny scenes (the 14-triangle room) fall back to the host builder
        assert st.numNodes == st.numTriangles - 1
    assert st.maxLeafSize <= 4 and st.maxDepth <= 64 and st.stackEntries >= st.maxDepth
    exported = lbvh.export_bvh()             # (nodes, tris, grid) + the 4-wide view as .wide
    nodes, tris, grid = exported
    _check_bvh(s.desc, st, nodes, tris, grid)
    p = api.make_params(W, H, spp=2, collect_stats=1)
    f_sah, f_lbvh = _render(gpu_ctx, sah, s, p), _render(gpu_ctx, lbvh, s, p)
    assert np.array_equal(f_sah.download(), f_lbvh.download()), "image must not depend on which builder made the tree"
    ref = oracle.render(s.desc, s.camera, s.scene_info(0), p, bvh=exported, threads=16)
    assert np.array_equal(f_lbvh.download(), ref.images[A.IMAGE_SHADOWED])
    g = f_lbvh.stats()
    assert (g.numRays, g.numNodeVisits, g.numTriTests, g.numHits) == (ref.stats.numRays, ref.stats.numNodeVisits, ref.stats.numTriTests, ref.stats.numHits)
    # the timed kernels (no counters) on the device-built tree, against the oracle directly
    p0 = api.make_params(W, H, spp=2, collect_stats=0)
    assert np.array_equal(_render(gpu_ctx, lbvh, s, p0).download(), ref.images[A.IMAGE_SHADOWED]), "timed kernels on the LBVH tree vs oracle"
    # deterministic: a second device build gives the same bytes
    n2, t2, g2 = api.Scene(gpu_ctx, _with_flags(s.desc, A.BUILD_DEVICE_LBVH)).export_bvh()
    assert bytes(n2) == bytes(nodes) and bytes(t2) == bytes(tris) and bytes(g2) == bytes(grid)


def _moved(setup, which, delta, scale=1.0):
    """copies of the instance list / light list with instance `which` translated by delta (and uniformly scaled)"""
    inst = [A.RtrInstance.from_buffer_copy(bytes(i)) for i in setup.host.instances()]
    m = np.array(inst[which].transform[:], np.float32).reshape(3, 4)
    m[:, :3] *= np.float32(scale)
    m[:, 3] += np.array(delta, np.float32)
    for k, v in enumerate(m.reshape(-1)):
        inst[which].transform[k] = float(v)
    lights = [A.RtrAreaLightInfo.from_buffer_copy(bytes(l)) for l in setup.host.lightInfos()]
    if which < len(lights):                      # a light instance: its LightInfo.transform is the same matrix, column-major
        cm = np.zeros((4, 4), np.float32); cm[:3, :] = m; cm[3, 3] = 1
        for k, v in enumerate(cm.T.reshape(-1)):
            lights[which].transform[k] = float(v)
    return inst, lights


@pytest.mark.parametrize("flags", [A.BUILD_HOST_SAH, A.BUILD_DEVICE_LBVH])
def test_refit_matches_fresh_build(gpu_ctx, oracle, scene_cache, flags):
    W, H = 256, 160
    s = scenes.cornell_box(W, H)
    scene = api.Scene(gpu_ctx, _with_flags(s.desc, flags))
    p = api.make_params(W, H, spp=1, collect_stats=1)
    before = _render(gpu_ctx, scene, s, p).download()
    # move the tall block (last instance) and the area light (instance 0), shrink the short block
    inst, lights = _moved(s, len(s.host.instances()) - 1, (-120.0, 0.0, -60.0))
    tmp = type("T", (), {})()
    tmp.host = type("H", (), {"instances": lambda self=None: inst, "lightInfos": lambda self=None: lights})()
    inst, lights = _moved(tmp, 0, (40.0, -30.0, 20.0))
    tmp.host = type("H", (), {"instances": lambda self=None: inst, "lightInfos": lambda self=None: lights})()
    inst, lights = _moved(tmp, len(inst) - 2, (15.0, 0.0, -25.0), scale=0.8)
    scene.update_instances(inst, lights)
    after = _render(gpu_ctx, scene, s, p)
    img = after.download()
    assert not np.array_equal(img, before)
    # the same scene built from scratch with the new transforms
    d2 = A.rtr_scene_desc.from_buffer_copy(bytes(s.desc))
    iarr = (A.RtrInstance * len(inst))(*inst)
    larr = (A.RtrAreaLightInfo * len(lights))(*lights)
    d2.instances = C.cast(iarr, C.POINTER(A.RtrInstance))
    d2.lights = C.cast(larr, C.POINTER(A.RtrAreaLightInfo))
    fresh = api.Scene(gpu_ctx, d2)
    assert np.array_equal(_render(gpu_ctx, fresh, s, p).download(), img), "refit image != image of a fresh build"
    # oracle on the refitted tree (exported from the device) and by brute force
    exported = scene.export_bvh()
    nodes, tris, grid = exported
    _check_bvh(d2, scene.stats(), nodes, tris, grid)
    ref = oracle.render(d2, s.camera, s.scene_info(0), p, bvh=exported, threads=8)
    assert np.array_equal(img, ref.images[A.IMAGE_SHADOWED])
    assert after.stats().numNodeVisits == ref.stats.numNodeVisits
    brute = oracle.render(d2, s.camera, s.scene_info(0), p, bvh=None, threads=8)
    assert np.array_equal(img, brute.images[A.IMAGE_SHADOWED])
    # refit back to the original transforms restores the original image bit for bit
    scene.update_instances(s.host.instances(), s.host.lightInfos())
    assert np.array_equal(_render(gpu_ctx, scene, s, p).download(), before)


def _refit_seeds():
    spec = os.environ.get("RTR_FUZZ_SEEDS", "")
    if not spec:
        return [11, 12, 13, 14]
    out = []
    for part in spec.split(","):
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


@pytest.mark.parametrize("seed", _refit_seeds())
def test_refit_random_affine_transforms(gpu_ctx, oracle, scene_cache, seed):
    """Every instance (lights included) gets a random affine transform — rotation, non-uniform scale, shear, mirroring, moves far
    outside the old bounds (the quantisation grid must follow) — by refit; the image must equal a fresh build's and the
    oracle's brute-force loop, for both kinds of tree."""
    rng = np.random.default_rng(seed)
    W, H = 192, 120
    s = scenes.cornell_box(W, H)
    base = [A.RtrInstance.from_buffer_copy(bytes(i)) for i in s.host.instances()]
    base_l = [A.RtrAreaLightInfo.from_buffer_copy(bytes(l)) for l in s.host.lightInfos()]
    centre = np.array([278.0, 273.0, 280.0])
    inst, lights = [], [A.RtrAreaLightInfo.from_buffer_copy(bytes(l)) for l in base_l]
    for k, b in enumerate(base):
        m = np.array(b.transform[:], np.float64).reshape(3, 4)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        lin = q @ np.diag(rng.uniform(0.3, 1.8, 3) * rng.choice([1.0, 1.0, 1.0, -1.0], 3))
        if rng.random() < 0.3:
            lin = lin + np.triu(rng.normal(0, 0.3, (3, 3)), 1)                       # shear
        if k < len(base_l):
            lin = np.eye(3) * rng.uniform(0.6, 1.5)                                  # lights: keep them facing the room
        move = rng.normal(0, 60, 3) + (rng.normal(0, 900, 3) if rng.random() < 0.15 else 0)
        a = np.eye(4); a[:3, :3] = lin; a[:3, 3] = centre - lin @ centre + move      # about the room centre, then moved
        full = np.eye(4); full[:3, :] = m
        out = (a @ full)[:3, :].astype(np.float32)
        i2 = A.RtrInstance.from_buffer_copy(bytes(b))
        for j, v in enumerate(out.reshape(-1)):
            i2.transform[j] = float(v)
        inst.append(i2)
        if k < len(base_l):
            cm = np.zeros((4, 4), np.float32); cm[:3, :] = out; cm[3, 3] = 1
            for j, v in enumerate(cm.T.reshape(-1)):
                lights[k].transform[j] = float(v)
    d2 = A.rtr_scene_desc.from_buffer_copy(bytes(s.desc))
    iarr = (A.RtrInstance * len(inst))(*inst)
    larr = (A.RtrAreaLightInfo * len(lights))(*lights)
    d2.instances = C.cast(iarr, C.POINTER(A.RtrInstance))
    d2.lights = C.cast(larr, C.POINTER(A.RtrAreaLightInfo))
    p = api.make_params(W, H, spp=1, collect_stats=1)
    brute = oracle.render(d2, s.camera, s.scene_info(seed), p, bvh=None, threads=8).images[A.IMAGE_SHADOWED]
    for flags in (A.BUILD_HOST_SAH, A.BUILD_DEVICE_LBVH):
        scene = api.Scene(gpu_ctx, _with_flags(s.desc, flags))
        scene.update_instances(inst, lights)
        img = _render(gpu_ctx, scene, s, p, frame_no=seed).download()
        nodes, tris, grid = scene.export_bvh()
        _check_bvh(d2, scene.stats(), nodes, tris, grid)
        assert np.array_equal(img, brute), (seed, flags, int((img != brute).sum()))
        fresh = api.Scene(gpu_ctx, _with_flags(d2, flags))
        assert np.array_equal(_render(gpu_ctx, fresh, s, p, frame_no=seed).download(), img), (seed, flags)
        p0 = api.make_params(W, H, spp=1)
        assert np.array_equal(_render(gpu_ctx, scene, s, p0, frame_no=seed).download(), img), (seed, flags, "production kernels")
        scene.close(); fresh.close()


def test_update_instances_rejects_topology_changes(gpu_ctx, scene_cache):
    s = scenes.cornell_box(64, 64)
    scene = api.Scene(gpu_ctx, s.desc)
    inst = [A.RtrInstance.from_buffer_copy(bytes(i)) for i in s.host.instances()]
    with pytest.raises(api.RtrError):
        scene.update_instances(inst[:-1])
    inst[3].meshIndex = 1
    with pytest.raises(api.RtrError):
        scene.update_instances(inst)
    inst = [A.RtrInstance.from_buffer_copy(bytes(i)) for i in s.host.instances()]
    inst[2].transform[3] = float("nan")
    with pytest.raises(api.RtrError):
        scene.update_instances(inst)


def test_lbvh_sponza_1080p_counts(gpu_ctx, scene_cache):
    """Quality of the device-built tree on the headline workload: same image, more node visits than SAH (reported)."""
    W, H = 1920, 1080
    s = scenes.sponza_class(W, H)
    p = api.make_params(W, H, collect_stats=1)
    sah = api.Scene(gpu_ctx, s.desc)
    lbvh = api.Scene(gpu_ctx, _with_flags(s.desc, A.BUILD_DEVICE_LBVH))
    a, b = _render(gpu_ctx, sah, s, p), _render(gpu_ctx, lbvh, s, p)
    assert np.array_equal(a.download(), b.download())
    ra, rb = a.stats(), b.stats()
    assert ra.numRays == rb.numRays
    print(f"\nSAH: build {sah.stats().buildMs:.1f} ms, {ra.numNodeVisits / ra.numRays:.1f} nodes/ray, {ra.totalMs:.2f} ms/frame; "
          f"LBVH: build {lbvh.stats().buildMs:.1f} ms (incl. upload + read-back), {rb.numNodeVisits / rb.numRays:.1f} nodes/ray, {rb.totalMs:.2f} ms/frame")
    assert rb.numNodeVisits < 4 * ra.numNodeVisits


def test_wide_centre_follows_the_flat_ground_of_a_small_object_scene(gpu_ctx, oracle, scene_cache):
    """RtrBvhGrid::wideCentreXY / Z (what the 4-wide records' half-float planes are offsets from) is chosen on the device from the
    leaf boxes: the bunny-class scene — a small object on a large ground plane — must get the window of its flat ground on y (where
    the planes then are exact: 6.3 instead of 9.1 visits per shadow ray, profiles/r02/wide_centre.log) and the mean leaf midpoint
    on x and z; the same bytes on every build; and the any-hit counters of the walk about that centre are the oracle's."""
    W, H = 320, 184
    s = scenes.bunny_class(W, H, subdiv=5)
    scene = api.Scene(gpu_ctx, s.desc)
    g = scene.stats().grid
    cx, cy, cz = g.wideCentreXY & 0xffff, g.wideCentreXY >> 16, g.wideCentreZ
    assert cy == 2048, (cx, cy, cz)                                   # the lowest 4096-step window: the ground
    assert abs(cx - 32768) < 2048 and abs(cz - 32768) < 2048, (cx, cy, cz)
    g2 = api.Scene(gpu_ctx, s.desc).stats().grid
    assert (g2.wideCentreXY, g2.wideCentreZ) == (g.wideCentreXY, g.wideCentreZ)
    wide = scene.export_bvh().wide
    inf_lo, inf_hi = 0x7c00, 0xfc00
    empties = [(n.plane[k][0], n.plane[k][1], n.plane[k][2]) for n in wide for k in (2, 3) if n.child[k] == -2**31]
    assert empties and all(e == (inf_lo | inf_lo << 16, inf_hi | inf_hi << 16, inf_lo | inf_hi << 16) for e in empties)
    p = api.make_params(W, H, spp=2, collect_stats=1)
    f = _render(gpu_ctx, scene, s, p)
    ref = oracle.render(s.desc, s.camera, s.scene_info(0), p, bvh=scene.export_bvh(), threads=16)
    assert np.array_equal(f.download(), ref.images[A.IMAGE_SHADOWED])
    st = f.stats()
    assert (st.numShadowNodeVisits, st.numShadowTriTests) == (ref.stats.numShadowNodeVisits, ref.stats.numShadowTriTests)


@pytest.mark.parametrize("which", ["cornell", "bunny", "sponza"])
@pytest.mark.parametrize("collapse", ["cost", "greedy"])
def test_device_wide_view_equals_its_host_restatement(gpu_ctx, scene_cache, which, collapse, monkeypatch):
    """The records the any-hit kernel walks (k_wide_centre_* + k_wide_nodes following the host builder's shapes, or the greedy rule +
    the breadth-first order) byte for byte against bvh_build.cpp's make_wide_host (rtr_host_build_bvh_wide), wide centre included —
    what lets CPU-only runs price a tree change with the oracle's counters (profiles/experiments/tree_lab.py)."""
    monkeypatch.setenv("RTR_BVH_WIDE_GREEDY", "1" if collapse == "greedy" else "0")
    s = {"cornell": lambda: scenes.cornell_box(64, 64), "bunny": lambda: scenes.bunny_class(64, 64), "sponza": lambda: scenes.sponza_class(64, 64)}[which]()
    scene = api.Scene(gpu_ctx, s.desc)
    dev = scene.export_bvh()
    host = api.host_build_bvh_wide(s.desc)
    assert bytes(dev[0]) == bytes(host[0]) and bytes(dev[1]) == bytes(host[1])
    g, h = scene.stats().grid, host.stats.grid
    assert (g.wideCentreXY, g.wideCentreZ) == (h.wideCentreXY, h.wideCentreZ)
    assert scene.stats().numWideNodes == host.stats.numWideNodes
    assert bytes(dev.wide) == bytes(host.wide)
