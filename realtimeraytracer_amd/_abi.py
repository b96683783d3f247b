"""ctypes mirror of include/rtr_types.h and include/rtr.h, and the loader of the two in-tree
shared libraries.  This is harness plumbing (tests, bench, smoke drive the C ABI through it); the
product is librtr_hip.so.  There is NO fallback: if librtr_hip.so is missing or a symbol of
include/rtr.h is not exported, import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_HIP_PATH = os.path.join(_HERE, "librtr_hip.so")
LIB_HOST_PATH = os.path.join(_HERE, "librtr_host.so")

f32, u32, u64, i32 = C.c_float, C.c_uint32, C.c_uint64, C.c_int32


class RtrVertex(C.Structure):
    _fields_ = [("position", f32 * 3), ("pad0", f32), ("normal", f32 * 3), ("pad1", f32), ("uv", f32 * 2), ("pad2", f32 * 2)]


class RtrCameraData(C.Structure):
    _fields_ = [("position", f32 * 3), ("_pad0", f32), ("topLeftViewportCorner", f32 * 3), ("_pad1", f32),
                ("horizontalViewportDelta", f32 * 3), ("_pad2", f32), ("verticalViewportDelta", f32 * 3), ("_pad3", f32)]


class RtrSceneInfo(C.Structure):
    _fields_ = [("frame", u32), ("numAreaLights", u32), ("_pad0", u32), ("_pad1", u32), ("camPosition", f32 * 3), ("pad2_", f32)]


class RtrObjectInfo(C.Structure):
    _fields_ = [("vertexOffset", u32), ("indexOffset", u32), ("pad0_", f32 * 2),
                ("usesColorMap", u32), ("usesSpecularMap", u32), ("usesMetallicMap", u32), ("usesOpacityMap", u32),
                ("colorIndex", u32), ("specularIndex", u32), ("metallicIndex", u32), ("opacityIndex", u32),
                ("color", f32 * 3), ("pad1_", f32), ("specular", f32), ("metallic", f32), ("pad3_", f32 * 2)]


class RtrAreaLightInfo(C.Structure):
    _fields_ = [("color", f32 * 3), ("intensity", f32), ("vertexOffset", u32), ("indexOffset", u32),
                ("numTriangles", u32), ("isTwoSided", u32), ("transform", f32 * 16)]


class RtrMesh(C.Structure):
    _fields_ = [("vertexOffset", u32), ("indexOffset", u32), ("vertexCount", u32), ("indexCount", u32),
                ("isOpaque", u32), ("_pad", u32 * 3)]


class RtrInstance(C.Structure):
    _fields_ = [("meshIndex", u32), ("customIndex", u32), ("_pad", u32 * 2), ("transform", f32 * 12)]


class RtrBvhNode(C.Structure):
    _fields_ = [("q", C.c_uint16 * 12), ("child", i32 * 2)]


class RtrBvhGrid(C.Structure):
    _fields_ = [("origin", f32 * 3), ("wideCentreXY", u32), ("scale", f32 * 3), ("wideCentreZ", u32)]


class RtrBvhTri(C.Structure):
    _fields_ = [("v0", f32 * 3), ("customIndex", u32), ("e1", f32 * 3), ("primitiveId", u32), ("e2", f32 * 3), ("flags", u32)]


class RtrWideNode(C.Structure):
    _fields_ = [("plane", (u32 * 3) * 4), ("child", i32 * 4)]


class rtr_texture(C.Structure):
    _fields_ = [("pixels", C.POINTER(C.c_uint8)), ("width", u32), ("height", u32), ("channels", u32), ("_pad", u32)]


class rtr_scene_desc(C.Structure):
    _fields_ = [("vertices", C.POINTER(RtrVertex)), ("numVertices", u32),
                ("indices", C.POINTER(u32)), ("numIndices", u32),
                ("meshes", C.POINTER(RtrMesh)), ("numMeshes", u32),
                ("instances", C.POINTER(RtrInstance)), ("numInstances", u32),
                ("objects", C.POINTER(RtrObjectInfo)), ("numObjects", u32),
                ("lights", C.POINTER(RtrAreaLightInfo)), ("numLights", u32),
                ("ltc1", C.POINTER(f32)), ("ltc2", C.POINTER(f32)),
                ("skyColor", f32 * 3), ("_pad", f32),
                ("textures", C.POINTER(rtr_texture)), ("numTextures", u32),
                ("hdri", C.POINTER(rtr_texture)), ("buildFlags", u32), ("_pad2", u32)]


class rtr_scene_stats(C.Structure):
    _fields_ = [("numTriangles", u32), ("numNodes", u32), ("maxDepth", u32), ("maxLeafSize", u32),
                ("bvhLayoutVersion", u32), ("stackEntries", u32), ("buildMs", f32), ("sahCost", f32),
                ("boundsMin", f32 * 3), ("boundsMax", f32 * 3), ("boxPad", f32), ("_pad", f32), ("grid", RtrBvhGrid),
                ("numWideNodes", u32), ("wideLayoutVersion", u32), ("_pad2", u32 * 2)]


class rtr_render_params(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("spp", u32), ("numShadowRays", u32), ("images", u32),
                ("bandRows", u32), ("shardIndex", u32), ("shardCount", u32), ("accumulate", u32),
                ("accumulatedFrames", u32), ("collectStats", u32), ("pipeline", u32)]


class rtr_frame_stats(C.Structure):
    _fields_ = [("numRays", u64), ("numPrimaryRays", u64), ("numShadowRays", u64), ("numNodeVisits", u64),
                ("numTriTests", u64), ("numHits", u64), ("numLightFetches", u64), ("numLightTriFetches", u64),
                ("numTexFetches", u64), ("numAlphaTests", u64), ("algorithmicBytes", u64), ("numShadowNodeVisits", u64), ("numShadowTriTests", u64), ("shadowTraceBytes", u64),
                ("totalMs", f32), ("primaryMs", f32), ("shadowGenMs", f32), ("shadowTraceMs", f32), ("resolveMs", f32),
                ("localRows", u32), ("localPixels", u32), ("pipelineUsed", u32), ("shadowTraceClockMHz", f32),
                ("shadowInnerIterations", u64), ("shadowInnerActiveLanes", u64), ("shadowTriIterations", u64), ("shadowTriActiveLanes", u64),
                ("shadowRefills", u64), ("shadowTailMs", f32), ("_padTail", u32), ("primaryTailRays", u64), ("shadowTailRays", u64),
                ("shadowTraceClockMinMHz", f32), ("shadowTraceClockMaxMHz", f32)]


assert C.sizeof(RtrVertex) == 48 and C.sizeof(RtrCameraData) == 64 and C.sizeof(RtrSceneInfo) == 32
assert C.sizeof(RtrObjectInfo) == 80 and C.sizeof(RtrAreaLightInfo) == 96
assert C.sizeof(RtrBvhNode) == 32 and C.sizeof(RtrBvhGrid) == 32 and C.sizeof(RtrBvhTri) == 48 and C.sizeof(RtrWideNode) == 64

# enum rtr_image
IMAGE_ANALYTIC, IMAGE_SHADOWED, IMAGE_UNSHADOWED = 0, 1, 2
IMAGE_DENOISED_SHADOWED, IMAGE_DENOISED_UNSHADOWED, IMAGE_FINAL, IMAGE_NORMAL, IMAGE_POSITION = 3, 4, 5, 6, 7
IMAGE_HDR = 16


def IMG_BIT(which):
    return 1 << which


IMAGES_FRAMEBUFFER = IMG_BIT(IMAGE_SHADOWED)
IMAGES_RAYGEN5 = IMG_BIT(0) | IMG_BIT(1) | IMG_BIT(2) | IMG_BIT(6) | IMG_BIT(7)
IMAGES_DENOISE = IMG_BIT(3) | IMG_BIT(4) | IMG_BIT(5)
BUILD_HOST_SAH, BUILD_DEVICE_LBVH = 0, 1

P = C.POINTER
VP = C.c_void_p

# every entry point include/rtr.h declares: name -> (restype, argtypes)
RTR_SYMBOLS = {
    "rtr_ctx_create": (C.c_int, [C.c_int, P(VP)]),
    "rtr_ctx_destroy": (None, [VP]),
    "rtr_ctx_set_stream": (C.c_int, [VP, VP]),
    "rtr_ctx_get_stream": (C.c_int, [VP, P(VP)]),
    "rtr_ctx_device_name": (C.c_int, [VP, C.c_char_p, C.c_size_t]),
    "rtr_ctx_set_tunable": (C.c_int, [VP, C.c_char_p, u32]),
    "rtr_ctx_get_tunable": (C.c_int, [VP, C.c_char_p, P(u32)]),
    "rtr_scene_create": (C.c_int, [VP, P(rtr_scene_desc), P(VP)]),
    "rtr_scene_create_like": (C.c_int, [VP, P(rtr_scene_desc), VP, P(VP)]),
    "rtr_scene_destroy": (None, [VP]),
    "rtr_scene_get_stats": (C.c_int, [VP, P(rtr_scene_stats)]),
    "rtr_scene_export_bvh": (C.c_int, [VP, VP, C.c_size_t, VP, C.c_size_t]),
    "rtr_scene_export_wide": (C.c_int, [VP, VP, C.c_size_t]),
    "rtr_host_build_bvh": (C.c_int, [P(rtr_scene_desc), P(rtr_scene_stats), VP, C.c_size_t, VP, C.c_size_t]),
    "rtr_host_build_bvh_wide": (C.c_int, [P(rtr_scene_desc), P(rtr_scene_stats), VP, C.c_size_t, VP, C.c_size_t, VP, C.c_size_t]),
    "rtr_scene_update_lights": (C.c_int, [VP, P(RtrAreaLightInfo), u32]),
    "rtr_scene_update_instances": (C.c_int, [VP, P(RtrInstance), u32, P(RtrAreaLightInfo), u32]),
    "rtr_frame_create": (C.c_int, [VP, u32, u32, u32, P(VP)]),
    "rtr_frame_destroy": (None, [VP]),
    "rtr_frame_bind_external": (C.c_int, [VP, C.c_int, VP, C.c_size_t]),
    "rtr_frame_device_ptr": (C.c_int, [VP, C.c_int, P(VP), P(C.c_size_t)]),
    "rtr_frame_download": (C.c_int, [VP, C.c_int, VP, C.c_size_t]),
    "rtr_frame_clear": (C.c_int, [VP]),
    "rtr_frame_get_stats": (C.c_int, [VP, P(rtr_frame_stats)]),
    "rtr_shard_rows": (u32, [u32, u32, u32]),
    "rtr_render": (C.c_int, [VP, P(RtrCameraData), P(RtrSceneInfo), P(rtr_render_params), VP]),
    "rtr_render_async": (C.c_int, [VP, P(RtrCameraData), P(RtrSceneInfo), P(rtr_render_params), VP]),
    "rtr_render_batch_async": (C.c_int, [VP, P(RtrCameraData), P(RtrSceneInfo), P(rtr_render_params), P(VP), u32]),
    "rtr_render_batch_limit": (C.c_int, [VP, P(rtr_render_params), u32, P(u32)]),
    "rtr_render_split_async": (C.c_int, [VP, P(RtrCameraData), P(RtrSceneInfo), P(rtr_render_params), VP, u32]),
    "rtr_render_split": (C.c_int, [VP, P(RtrCameraData), P(RtrSceneInfo), P(rtr_render_params), VP, u32]),
    "rtr_frame_wait": (C.c_int, [VP]),
    "rtr_deinterleave_bands": (C.c_int, [VP, VP, VP, u32, u32, u32, u32]),
    "rtr_denoise_combine": (C.c_int, [VP, C.c_int]),
    "rtr_last_error": (C.c_char_p, []),
    "rtr_status_string": (C.c_char_p, [C.c_int]),
    "rtr_abi_version": (C.c_int, []),
    "rtr_kernel_revision": (C.c_char_p, []),
    "rtr_check_scene_limits": (C.c_int, [C.c_uint64, C.c_uint64]),
}

RTRH_SYMBOLS = {
    "rtrh_last_error": (C.c_char_p, []),
    "rtrh_scene_new": (VP, []),
    "rtrh_scene_free": (None, [VP]),
    "rtrh_add_light": (C.c_int, [VP, f32, P(f32), C.c_int, C.c_int, C.c_char_p]),
    "rtrh_light_move": (C.c_int, [VP, C.c_int, P(f32)]),
    "rtrh_light_scale": (C.c_int, [VP, C.c_int, P(f32)]),
    "rtrh_light_rotate": (C.c_int, [VP, C.c_int, P(f32)]),
    "rtrh_light_transform": (C.c_int, [VP, C.c_int, P(f32)]),
    "rtrh_add_object": (C.c_int, [VP, C.c_char_p]),
    "rtrh_object_move": (C.c_int, [VP, C.c_int, P(f32)]),
    "rtrh_object_scale": (C.c_int, [VP, C.c_int, f32]),
    "rtrh_object_rotate": (C.c_int, [VP, C.c_int, P(f32)]),
    "rtrh_object_set_color": (C.c_int, [VP, C.c_int, P(f32)]),
    "rtrh_object_set_color_map": (C.c_int, [VP, C.c_int, C.c_char_p]),
    "rtrh_object_set_specular": (C.c_int, [VP, C.c_int, f32]),
    "rtrh_object_set_metallic": (C.c_int, [VP, C.c_int, f32]),
    "rtrh_object_transform": (C.c_int, [VP, C.c_int, P(f32)]),
    "rtrh_num_objects": (C.c_int, [VP]),
    "rtrh_num_lights": (C.c_int, [VP]),
    "rtrh_add_obj_mtl_pair": (C.c_int, [VP, C.c_char_p, C.c_char_p]),
    "rtrh_set_ltc": (C.c_int, [VP, P(f32), P(f32)]),
    "rtrh_set_sky": (C.c_int, [VP, P(f32)]),
    "rtrh_set_hdri": (C.c_int, [VP, C.c_char_p]),
    "rtrh_load_image": (C.c_int, [C.c_char_p, C.c_int, P(C.c_int), P(C.c_int), VP, C.c_size_t]),
    "rtrh_build": (C.c_int, [VP]),
    "rtrh_get_desc": (C.c_int, [VP, P(rtr_scene_desc)]),
    "rtrh_object_info": (C.c_int, [VP, C.c_int, P(u32), P(u32), P(u32)]),
    "rtrh_load_model": (C.c_int, [VP, C.c_char_p]),
    "rtrh_obj_dump": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p]),
    "rtrh_camera_new": (VP, [f32, P(f32), P(f32), P(f32), C.c_int, C.c_int]),
    "rtrh_camera_free": (None, [VP]),
    "rtrh_camera_get": (C.c_int, [VP, P(RtrCameraData)]),
    "rtrh_camera_set_position": (C.c_int, [VP, P(f32)]),
    "rtrh_camera_rotate_y": (C.c_int, [VP, f32]),
    "rtrh_camera_mouse": (C.c_int, [VP, f32, f32]),
    "rtrh_camera_apply_input": (C.c_int, [VP, C.c_char_p, f32, f32, f32, f32, P(C.c_int)]),
    "rtrh_camera_state": (C.c_int, [VP, P(f32)]),
}


class rtr_mgpu_info(C.Structure):
    _fields_ = [("nranks", C.c_int), ("nlocal", C.c_int), ("firstRank", C.c_int), ("framesInFlight", C.c_int),
                ("selfExchange", C.c_int), ("aborted", C.c_int), ("rcclVersion", C.c_int), ("timeoutMs", C.c_int),
                ("enqueueHostMs", C.c_double), ("enqueueRcclMs", C.c_double), ("enqueuedFrames", C.c_ulonglong)]


class rtr_mgpu_op(C.Structure):
    """one operation of rtr_mgpu_plan (include/rtr_mgpu.h)"""
    _fields_ = [("kind", C.c_int32), ("stream", C.c_int32), ("peer", C.c_int32), ("buffer", C.c_int32), ("event", C.c_int32),
                ("slot", C.c_int32), ("offset", C.c_uint64), ("bytes", C.c_uint64)]


MGPU_OP_WAIT, MGPU_OP_RENDER, MGPU_OP_RECORD, MGPU_OP_GROUP_START, MGPU_OP_RECV, MGPU_OP_SEND, MGPU_OP_GROUP_END, MGPU_OP_DEINTERLEAVE = range(1, 9)
MGPU_STREAM_RENDER, MGPU_STREAM_COMM = 0, 1
MGPU_BUF_NONE, MGPU_BUF_LOCAL, MGPU_BUF_GATHER, MGPU_BUF_SELF_SRC, MGPU_BUF_FULL = range(5)
MGPU_EV_NONE, MGPU_EV_RENDER_DONE, MGPU_EV_COMM_DONE = range(3)
MGPU_PLAN_MAX_OPS = 32
MGPU_MAX_RANKS = 16
MGPU_BATCH_PLAN_MAX_OPS = 6 + 32 * (5 + MGPU_MAX_RANKS)


MAX_BATCH = 32
MAX_SPLIT = 16
MGPU_ID_BYTES = 128
MGPU_MAX_SLOTS = 64
MGPU_NO_EXCHANGE = 1
MGPU_GROUP_PER_SLOT = 2

# every entry point include/rtr_mgpu.h declares
MGPU_SYMBOLS = {
    "rtr_mgpu_unique_id": (C.c_int, [VP]),
    "rtr_mgpu_create": (C.c_int, [P(C.c_int), C.c_int, C.c_int, P(VP)]),
    "rtr_mgpu_create_rank": (C.c_int, [C.c_int, C.c_int, C.c_int, VP, C.c_int, P(VP)]),
    "rtr_mgpu_destroy": (None, [VP]),
    "rtr_mgpu_scene_create": (C.c_int, [VP, P(rtr_scene_desc)]),
    "rtr_mgpu_render_async": (C.c_int, [VP, C.c_int, P(RtrCameraData), P(RtrSceneInfo), P(rtr_render_params), C.c_int]),
    "rtr_mgpu_render_batch_async": (C.c_int, [VP, P(C.c_int), C.c_int, P(RtrCameraData), P(RtrSceneInfo), P(rtr_render_params), C.c_int]),
    "rtr_mgpu_wait": (C.c_int, [VP, C.c_int]),
    "rtr_mgpu_render": (C.c_int, [VP, P(RtrCameraData), P(RtrSceneInfo), P(rtr_render_params)]),
    "rtr_mgpu_frame_download": (C.c_int, [VP, C.c_int, VP, C.c_size_t]),
    "rtr_mgpu_frame_device_ptr": (C.c_int, [VP, C.c_int, P(VP), P(C.c_size_t)]),
    "rtr_mgpu_shard_download": (C.c_int, [VP, C.c_int, C.c_int, VP, C.c_size_t]),
    "rtr_mgpu_frame_stats": (C.c_int, [VP, C.c_int, C.c_int, P(rtr_frame_stats)]),
    "rtr_mgpu_get_info": (C.c_int, [VP, P(rtr_mgpu_info)]),
    "rtr_mgpu_plan": (C.c_int, [C.c_int, C.c_int, u32, u32, u32, C.c_int, C.c_int, P(rtr_mgpu_op), C.c_int, P(C.c_int)]),
    "rtr_mgpu_plan_batch": (C.c_int, [C.c_int, C.c_int, u32, u32, u32, C.c_int, C.c_int, C.c_int, P(rtr_mgpu_op), C.c_int, P(C.c_int)]),
    "rtr_mgpu_set_timeout_ms": (C.c_int, [VP, u32]),
    "rtr_mgpu_last_error": (C.c_char_p, []),
}


def _bind(path, table, what):
    if not os.path.exists(path):
        raise ImportError(
            f"{what} not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C realtimeraytracer_amd/csrc`). There is no pure-Python / CPU fallback for the ray-tracing path.")
    lib = C.CDLL(path)
    for name, (res, args) in table.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{path} does not export {name} (declared in include/rtr.h)") from e
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None
_hip_hooks = None
_host = None
_mgpu = None
LIB_MGPU_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librtr_mgpu.so")
# The same sources built with -DRTR_TEST_HOOKS (csrc/Makefile): the environment switches the tests need — a short overflow list, a
# forced pre-fill of the visibility array, ranks sharing one device, a one-rank communicator that still exchanges — exist ONLY in these
# two; the product libraries above do not contain them.
LIB_HIP_HOOKS_PATH = os.path.join(_HERE, "librtr_hip_test.so")
LIB_MGPU_HOOKS_PATH = os.path.join(_HERE, "librtr_mgpu_test.so")


def test_hooks_requested():
    """This process wants the multi-GPU library's test build: RTR_TEST_HOOKS=1, or one of the two switches only that build has is set
    (child processes of the GPU tests, bench.py's one-GPU rehearsal of N ranks)"""
    e = os.environ
    return e.get("RTR_TEST_HOOKS") == "1" or e.get("RTR_MGPU_SELF_EXCHANGE") == "1" or e.get("RTR_MGPU_TEST_SHARED_DEVICE") == "1" or e.get("RTR_MGPU_TEST_WRONG_PLACE") in ("1", "2")


def mgpu_lib():
    """librtr_mgpu.so (include/rtr_mgpu.h): the tile-sharded frame over several GPUs, RCCL inside.  Loaded after librtr_hip.so.
    With RTR_TEST_HOOKS=1: librtr_mgpu_test.so, the same sources with the test hooks compiled in."""
    global _mgpu
    if _mgpu is None:
        hip_lib()
        if test_hooks_requested():
            _mgpu = _bind(LIB_MGPU_HOOKS_PATH, MGPU_SYMBOLS, "librtr_mgpu_test.so")
        else:
            _mgpu = _bind(LIB_MGPU_PATH, MGPU_SYMBOLS, "librtr_mgpu.so")
    return _mgpu


def hip_lib_with_hooks():
    """librtr_hip_test.so: librtr_hip.so's sources with -DRTR_TEST_HOOKS, for the few tests that force a rare path (api.Context(...,
    test_hooks=True)); may live in one process beside the product library."""
    global _hip_hooks
    if _hip_hooks is None:
        hip_lib()
        _hip_hooks = _bind(LIB_HIP_HOOKS_PATH, RTR_SYMBOLS, "librtr_hip_test.so")
    return _hip_hooks


def hip_lib():
    """librtr_hip.so (the product).  Loaded on first use; raises ImportError if absent/incomplete.

    PyTorch-ROCm wheels bundle their own copy of the HIP runtime (torch/lib/libamdhip64.so, loaded into the
    global symbol scope).  A process must end up with ONE initialised HIP runtime: if this library were loaded
    first it would bind to /opt/rocm's runtime and torch's copy would later find "No HIP GPUs".  The harness
    (tests, bench) uses torch for device tensors and RCCL, so torch is imported first and librtr_hip.so then
    resolves its hip* symbols against the runtime already in the process — which also makes a torch stream
    handle valid for rtr_ctx_set_stream.  A C/C++ host without torch simply uses the ROCm runtime it links."""
    global _hip
    if _hip is None:
        try:
            import torch  # noqa: F401  (see docstring: load order matters, nothing else of torch is used here)
        except ImportError:
            pass
        _hip = _bind(LIB_HIP_PATH, RTR_SYMBOLS, "librtr_hip.so")
    return _hip


def host_lib():
    """librtr_host.so (C shim over the C++ scene layer; no HIP dependency)."""
    global _host
    if _host is None:
        _host = _bind(LIB_HOST_PATH, RTRH_SYMBOLS, "librtr_host.so")
    return _host
