"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE.  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under realtimeraytracer_amd/ may
import this module (tests/test_abi.py::test_product_never_touches_the_oracle enforces it)."""
import ctypes as C
import os
import subprocess

import numpy as np

from realtimeraytracer_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


class oracle_scene(C.Structure):
    _fields_ = [("desc", A.rtr_scene_desc), ("nodes", C.POINTER(A.RtrBvhNode)), ("numNodes", A.u32),
                ("tris", C.POINTER(A.RtrBvhTri)), ("numTris", A.u32), ("grid", A.RtrBvhGrid),
                ("wide", C.POINTER(A.RtrWideNode)), ("numWide", A.u32), ("primaryPackets", A.u32), ("shadowWalk", A.u32), ("walkProfile", C.POINTER(C.c_uint64)), ("walkRays", C.POINTER(C.c_float)), ("walkRaysCap", C.c_uint64), ("walkRaysCount", C.POINTER(C.c_uint64))]


class oracle_walk_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("occludedRays", "occludedVisits", "occludedTests", "visibleRays", "visibleVisits", "visibleTests", "ownLeafRays", "ownLeafStopped")]


class oracle_out(C.Structure):
    _fields_ = [("analytic", C.POINTER(A.u32)), ("shadowed", C.POINTER(A.u32)), ("unshadowed", C.POINTER(A.u32)),
                ("normal", C.POINTER(A.u32)), ("position", C.POINTER(A.u32)), ("hdr", C.POINTER(A.f32)),
                ("stats", A.rtr_frame_stats), ("walk", oracle_walk_stats)]


_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.POINTER(oracle_scene), C.POINTER(A.RtrCameraData), C.POINTER(A.RtrSceneInfo),
                                    C.POINTER(A.rtr_render_params), C.POINTER(oracle_out), C.c_int]
        L.oracle_primary_hits.restype = C.c_int
        L.oracle_primary_hits.argtypes = [C.POINTER(oracle_scene), C.POINTER(A.RtrCameraData), C.POINTER(A.rtr_render_params),
                                          A.VP, A.VP, A.VP, A.VP, A.VP, C.c_int]
        L.oracle_denoise_combine.restype = C.c_int
        L.oracle_denoise_combine.argtypes = [A.u32, A.u32] + [A.VP] * 8 + [C.c_int]
        L.oracle_pcg_hash.restype = A.u32
        L.oracle_pcg_hash.argtypes = [A.u32]
        for n in ("oracle_random",):
            getattr(L, n).restype = A.f32
            getattr(L, n).argtypes = [A.u32]
        L.oracle_pow.restype = A.f32
        L.oracle_pow.argtypes = [A.f32, A.f32]
        for n in ("oracle_log2", "oracle_exp2"):
            getattr(L, n).restype = A.f32
            getattr(L, n).argtypes = [A.f32]
        L.oracle_pack_bgra8.restype = A.u32
        L.oracle_pack_bgra8.argtypes = [A.f32, A.f32, A.f32]
        L.oracle_unorm8_to_float_fast.restype = A.f32
        L.oracle_unorm8_to_float_fast.argtypes = [A.u32]
        L.oracle_div_by_mismatches.restype = C.c_uint64
        L.oracle_div_by_mismatches.argtypes = [A.f32, A.u32, A.u32]
        L.oracle_mt.restype = C.c_int
        L.oracle_mt.argtypes = [C.POINTER(A.f32)] * 5 + [A.f32, C.POINTER(A.f32)]
        _lib = L
    return _lib


def make_scene(desc, bvh=None, primary_packets=False, primary_wide=False, shadow_walk=0, walk_profile=None, walk_rays=None):
    """bvh = (nodes, tris, grid) from api.Scene.export_bvh() (ctypes arrays + the RtrBvhGrid of rtr_scene_stats), or None
    for brute force.  primary_packets: the staged pipeline's camera rays are walked one ray per lane (the product's default, k_primary) or
    tile by tile (tunable primary_packet = 1, k_primary_packet) — it decides work counters only."""
    s = oracle_scene()
    s.desc = desc
    s.shadowWalk = int(shadow_walk)
    if walk_rays is not None:         # (numpy float32 (cap, 8), numpy uint64 (1,)): experiments only
        s.walkRays = walk_rays[0].ctypes.data_as(C.POINTER(C.c_float)); s.walkRaysCap = len(walk_rays[0]); s.walkRaysCount = walk_rays[1].ctypes.data_as(C.POINTER(C.c_uint64))
    if walk_profile is not None:      # numpy uint64 (numWide, 4, 3): experiments only
        s.walkProfile = walk_profile.ctypes.data_as(C.POINTER(C.c_uint64))
    s.primaryPackets = 2 if primary_wide else (1 if primary_packets else 0)      # how the staged pipeline walks its camera rays: 0 one ray per lane over the BVH2, 1 8x8 packets, 2 one ray per lane over the 4-wide view
    if bvh is not None:
        nodes, tris, grid = bvh
        wide = getattr(bvh, "wide", None)
        if wide is not None:
            s.wide = C.cast(wide, C.POINTER(A.RtrWideNode))
            s.numWide = len(wide)
        s.grid = grid
        s.nodes = C.cast(nodes, C.POINTER(A.RtrBvhNode))
        s.numNodes = len(nodes)
        s.tris = C.cast(tris, C.POINTER(A.RtrBvhTri))
        s.numTris = len(tris)
    s._keep = bvh
    return s


class Result:
    pass


def render(desc, camera, scene_info, params, bvh=None, images=A.IMAGES_FRAMEBUFFER, hdr=None, threads=1, primary_packets=False, primary_wide=False, shadow_walk=0, walk_profile=None, own_leaf=True, walk_rays=None, nearest_first=False):
    """nearest_first=True: the shadow walk of rounds 1-4 (a library built with -DRTR_SHADOW_FAR_FIRST=0).  own_leaf=False: the product's tunable trace_own_leaf = 0 (shadow rays never start at their own triangle's leaf).
    Returns a Result with numpy uint32 images (rows x width) keyed like rtr_image, .hdr and .stats."""
    L = lib()
    rows = _shard_rows(params.height, params.bandRows or 8, params.shardCount or 1)
    W = params.width
    sc = make_scene(desc, bvh, primary_packets, primary_wide, int(shadow_walk) | (0 if own_leaf else 2) | (4 if nearest_first else 0), walk_profile, walk_rays)
    out = oracle_out()
    r = Result()
    r.images = {}
    names = {A.IMAGE_ANALYTIC: "analytic", A.IMAGE_SHADOWED: "shadowed", A.IMAGE_UNSHADOWED: "unshadowed",
             A.IMAGE_NORMAL: "normal", A.IMAGE_POSITION: "position"}
    for which, field in names.items():
        if images & A.IMG_BIT(which):
            arr = np.zeros((rows, W), dtype=np.uint32)
            r.images[which] = arr
            setattr(out, field, arr.ctypes.data_as(C.POINTER(A.u32)))
    if hdr is not None or (images & A.IMG_BIT(A.IMAGE_HDR)):
        r.hdr = hdr if hdr is not None else np.zeros((rows, W, 4), dtype=np.float32)
        out.hdr = r.hdr.ctypes.data_as(C.POINTER(A.f32))
    else:
        r.hdr = None
    rc = L.oracle_render(C.byref(sc), C.byref(camera), C.byref(scene_info), C.byref(params), C.byref(out), int(threads))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed: {rc}")
    r.stats = out.stats
    r.walk = out.walk
    return r


def primary_hits(desc, camera, params, bvh=None, threads=1):
    L = lib()
    n = params.width * params.height * params.spp
    t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    cu = np.zeros(n, np.uint32); pr = np.zeros(n, np.uint32)
    sc = make_scene(desc, bvh)
    rc = L.oracle_primary_hits(C.byref(sc), C.byref(camera), C.byref(params), t.ctypes.data, u.ctypes.data, v.ctypes.data,
                               cu.ctypes.data, pr.ctypes.data, int(threads))
    if rc != 0:
        raise RuntimeError(f"oracle_primary_hits failed: {rc}")
    return t, u, v, cu, pr


def denoise_combine(analytic, shadowed, unshadowed, normal, position, iterations=4):
    """Returns dict with the 8-image state after the reference's denoise+combine protocol."""
    L = lib()
    H, W = analytic.shape
    sh, un = shadowed.copy(), unshadowed.copy()
    dsh, dun, fin = np.zeros_like(sh), np.zeros_like(sh), np.zeros_like(sh)
    an, no, po = [np.ascontiguousarray(x) for x in (analytic, normal, position)]
    rc = L.oracle_denoise_combine(W, H, an.ctypes.data, sh.ctypes.data, un.ctypes.data, no.ctypes.data, po.ctypes.data,
                                  dsh.ctypes.data, dun.ctypes.data, fin.ctypes.data, int(iterations))
    if rc != 0:
        raise RuntimeError(f"oracle_denoise_combine failed: {rc}")
    return {A.IMAGE_SHADOWED: sh, A.IMAGE_UNSHADOWED: un, A.IMAGE_DENOISED_SHADOWED: dsh,
            A.IMAGE_DENOISED_UNSHADOWED: dun, A.IMAGE_FINAL: fin}


def _shard_rows(height, band_rows, shard_count):
    if shard_count <= 1:
        return height
    bands = (height + band_rows - 1) // band_rows
    per = (bands + shard_count - 1) // shard_count
    return per * band_rows
