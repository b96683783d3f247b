"""Thin Python handles over the C ABI (include/rtr.h): Context / Scene / Frame / render.
Every non-zero status becomes RtrError carrying rtr_last_error() — the same convention as the C++
shim csrc/host/renderer.hpp (reference: exceptions caught once in main, src/main.cpp:12-15)."""
import ctypes as C

import numpy as np

from . import _abi as A


class RtrError(RuntimeError):
    def __init__(self, status, what):
        lib = A.hip_lib()
        self.status = status
        self.status_name = lib.rtr_status_string(status).decode()
        super().__init__(f"{what}: {self.status_name}: {lib.rtr_last_error().decode()}")


def _check(status, what):
    if status != 0:
        raise RtrError(status, what)


class Context:
    def __init__(self, device=0, test_hooks=False):
        """test_hooks: everything made from this context goes through librtr_hip_test.so (the product's sources + the test switches)"""
        self.lib = A.hip_lib_with_hooks() if test_hooks else A.hip_lib()
        self.h = A.VP()
        _check(self.lib.rtr_ctx_create(device, C.byref(self.h)), "rtr_ctx_create")

    def set_tunable(self, name, value):
        _check(self.lib.rtr_ctx_set_tunable(self.h, name.encode(), int(value)), "rtr_ctx_set_tunable")

    def get_tunable(self, name):
        v = C.c_uint32(0)
        _check(self.lib.rtr_ctx_get_tunable(self.h, name.encode(), C.byref(v)), "rtr_ctx_get_tunable")
        return int(v.value)

    def set_stream(self, stream_ptr):
        _check(self.lib.rtr_ctx_set_stream(self.h, A.VP(stream_ptr) if stream_ptr else None), "rtr_ctx_set_stream")

    def device_name(self):
        buf = C.create_string_buffer(256)
        _check(self.lib.rtr_ctx_device_name(self.h, buf, 256), "rtr_ctx_device_name")
        return buf.value.decode()

    def close(self):
        if self.h:
            self.lib.rtr_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BvhExport(tuple):
    """(nodes, tris, grid) of rtr_scene_export_bvh, with the 4-wide view of rtr_scene_export_wide as .wide"""
    wide = None
    stats = None


class Scene:
    def __init__(self, ctx, desc, like=None):
        """like: a Scene made from the same description whose tree is uploaded instead of built again (rtr_scene_create_like)"""
        self.ctx, self.lib = ctx, ctx.lib
        self.h = A.VP()
        if like is None:
            _check(self.lib.rtr_scene_create(ctx.h, C.byref(desc), C.byref(self.h)), "rtr_scene_create")
        else:
            _check(self.lib.rtr_scene_create_like(ctx.h, C.byref(desc), like.h, C.byref(self.h)), "rtr_scene_create_like")

    def stats(self):
        s = A.rtr_scene_stats()
        _check(self.lib.rtr_scene_get_stats(self.h, C.byref(s)), "rtr_scene_get_stats")
        return s

    def export_bvh(self):
        s = self.stats()
        nodes = (A.RtrBvhNode * s.numNodes)()
        # numTriangles==0 scenes carry one dummy record
        ntri = max(s.numTriangles, 1)
        tris = (A.RtrBvhTri * ntri)()
        _check(self.lib.rtr_scene_export_bvh(self.h, nodes, C.sizeof(nodes), tris, C.sizeof(tris)), "rtr_scene_export_bvh")
        out = BvhExport((nodes, tris, s.grid))      # the nodes' 16-bit planes live on s.grid
        if s.numWideNodes:
            wn = (A.RtrWideNode * s.numWideNodes)()
            _check(self.lib.rtr_scene_export_wide(self.h, wn, C.sizeof(wn)), "rtr_scene_export_wide")
            out.wide = wn
        return out

    def update_instances(self, instances, lights=None):
        """rtr_scene_update_instances: new transforms (+ optional light infos) -> device-side re-flatten + BVH refit."""
        arr = (A.RtrInstance * len(instances))(*instances)
        if lights is None:
            _check(self.lib.rtr_scene_update_instances(self.h, arr, len(instances), None, 0), "rtr_scene_update_instances")
        else:
            larr = (A.RtrAreaLightInfo * len(lights))(*lights)
            _check(self.lib.rtr_scene_update_instances(self.h, arr, len(instances), larr, len(lights)), "rtr_scene_update_instances")

    def update_lights(self, lights):
        arr = (A.RtrAreaLightInfo * len(lights))(*lights)
        _check(self.lib.rtr_scene_update_lights(self.h, arr, len(lights)), "rtr_scene_update_lights")

    def close(self):
        if self.h:
            self.lib.rtr_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Frame:
    def __init__(self, ctx, width, rows, images=A.IMAGES_FRAMEBUFFER):
        self.ctx, self.lib = ctx, ctx.lib
        self.width, self.rows = width, rows
        self.h = A.VP()
        _check(self.lib.rtr_frame_create(ctx.h, width, rows, images, C.byref(self.h)), "rtr_frame_create")

    def download(self, which=A.IMAGE_SHADOWED):
        if which == A.IMAGE_HDR:
            out = np.empty((self.rows, self.width, 4), dtype=np.float32)
        else:
            out = np.empty((self.rows, self.width), dtype=np.uint32)
        _check(self.lib.rtr_frame_download(self.h, which, out.ctypes.data_as(A.VP), out.nbytes), "rtr_frame_download")
        return out

    def bind_external(self, which, device_ptr, nbytes):
        _check(self.lib.rtr_frame_bind_external(self.h, which, A.VP(device_ptr), nbytes), "rtr_frame_bind_external")

    def device_ptr(self, which=A.IMAGE_SHADOWED):
        p, n = A.VP(), C.c_size_t()
        _check(self.lib.rtr_frame_device_ptr(self.h, which, C.byref(p), C.byref(n)), "rtr_frame_device_ptr")
        return p.value, n.value

    def clear(self):
        _check(self.lib.rtr_frame_clear(self.h), "rtr_frame_clear")

    def denoise_combine(self, iterations=4):
        """reference frame loop tail (application.cppm:391-445): 4 x a-trous on both sampled images, then combine."""
        _check(self.lib.rtr_denoise_combine(self.h, iterations), "rtr_denoise_combine")

    def wait(self):
        _check(self.lib.rtr_frame_wait(self.h), "rtr_frame_wait")

    def stats(self):
        s = A.rtr_frame_stats()
        _check(self.lib.rtr_frame_get_stats(self.h, C.byref(s)), "rtr_frame_get_stats")
        return s

    def close(self):
        if self.h:
            self.lib.rtr_frame_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_params(width, height, spp=1, shadow_rays=3, images=A.IMAGES_FRAMEBUFFER, band_rows=8, shard_index=0,
                shard_count=1, accumulate=0, accumulated_frames=0, collect_stats=0, pipeline=0):
    return A.rtr_render_params(width, height, spp, shadow_rays, images, band_rows, shard_index, shard_count,
                               accumulate, accumulated_frames, collect_stats, pipeline)


def shard_rows(height, band_rows=8, shard_count=1):
    return A.hip_lib().rtr_shard_rows(height, band_rows, shard_count)


def render(scene, camera, scene_info, params, frame, asynchronous=False):
    fn = scene.lib.rtr_render_async if asynchronous else scene.lib.rtr_render
    _check(fn(scene.h, C.byref(camera), C.byref(scene_info), C.byref(params), frame.h), "rtr_render")


def render_split(scene, camera, scene_info, params, frame, parts, asynchronous=False):
    """rtr_render_split[_async]: ONE frame as `parts` band-shards on streams of their own, written in place (the latency form)"""
    fn = scene.lib.rtr_render_split_async if asynchronous else scene.lib.rtr_render_split
    _check(fn(scene.h, C.byref(camera), C.byref(scene_info), C.byref(params), frame.h, int(parts)), "rtr_render_split")


def marshal_batch(cameras, scene_infos, frames):
    """the argument arrays of rtr_render_batch_async, built once (a caller that knows its next launches prepares them ahead)"""
    n = len(frames)
    return ((A.RtrCameraData * n)(*cameras), (A.RtrSceneInfo * n)(*scene_infos), (A.VP * n)(*[f.h.value for f in frames]), n)


def render_batch(scene, cameras, scene_infos, params, frames, marshalled=None):
    """rtr_render_batch_async: len(frames) <= A.MAX_BATCH frames in one launch of every kernel; asynchronous, join with frames[k].wait()."""
    cams, infos, hs, n = marshalled if marshalled is not None else marshal_batch(cameras, scene_infos, frames)
    _check(scene.lib.rtr_render_batch_async(scene.h, cams, infos, C.byref(params), hs, n), "rtr_render_batch_async")


def render_batch_limit(scene, params, num_area_lights):
    """rtr_render_batch_limit: how many frames of these params one launch of the pipeline takes (<= A.MAX_BATCH)"""
    n = C.c_uint32(0)
    _check(scene.lib.rtr_render_batch_limit(scene.h, C.byref(params), num_area_lights, C.byref(n)), "rtr_render_batch_limit")
    return int(n.value)


def deinterleave_bands(ctx, gathered_ptr, dst_ptr, width, height, band_rows, shard_count):
    _check(ctx.lib.rtr_deinterleave_bands(ctx.h, A.VP(gathered_ptr), A.VP(dst_ptr), width, height, band_rows, shard_count),
           "rtr_deinterleave_bands")


def host_build_bvh(desc):
    """rtr_host_build_bvh: the product's BVH builder without a device -> (stats, nodes, tris)."""
    lib = A.hip_lib()
    st = A.rtr_scene_stats()
    _check(lib.rtr_host_build_bvh(C.byref(desc), C.byref(st), None, 0, None, 0), "rtr_host_build_bvh")
    nodes = (A.RtrBvhNode * st.numNodes)()
    tris = (A.RtrBvhTri * max(st.numTriangles, 1))()
    _check(lib.rtr_host_build_bvh(C.byref(desc), C.byref(st), nodes, C.sizeof(nodes), tris, C.sizeof(tris)), "rtr_host_build_bvh")
    return st, nodes, tris


def host_build_bvh_wide(desc):
    """rtr_host_build_bvh_wide: the build plus the 4-wide view the device would hold for it -> BvhExport (nodes, tris, grid) with
    .wide and .stats; what the oracle needs to walk a shadow ray the way k_shadow_trace4 does, on a CPU-only box."""
    lib = A.hip_lib()
    st = A.rtr_scene_stats()
    _check(lib.rtr_host_build_bvh_wide(C.byref(desc), C.byref(st), None, 0, None, 0, None, 0), "rtr_host_build_bvh_wide")
    nodes = (A.RtrBvhNode * st.numNodes)()
    tris = (A.RtrBvhTri * max(st.numTriangles, 1))()
    wide = (A.RtrWideNode * st.numWideNodes)()
    _check(lib.rtr_host_build_bvh_wide(C.byref(desc), C.byref(st), nodes, C.sizeof(nodes), tris, C.sizeof(tris), wide, C.sizeof(wide)), "rtr_host_build_bvh_wide")
    out = BvhExport((nodes, tris, st.grid))
    out.wide = wide
    out.stats = st
    return out
