"""realtimeraytracer_amd — MI355X-native replacement for the ray-tracing hot path of
DallinClark/RealTimeRaytracer (Vulkan dispatch + GLSL RT shaders -> hand-written HIP for gfx950
behind the C ABI of include/rtr.h).

The product is `librtr_hip.so` (C ABI) plus the C++ host scene layer under csrc/host/.  This Python
package is the harness-side binding used by tests/, bench.py and __graft_entry__.py:

    from realtimeraytracer_amd import api, host, scenes
    sc = scenes.cornell_box(width=256, height=256)        # host scene (Camera/Object/AreaLight/OBJ ingest)
    ctx = api.Context(0); scene = api.Scene(ctx, sc.desc) # -> rtr_scene_create (BVH build + upload)
    frame = api.Frame(ctx, 256, 256); api.render(scene, sc.camera, sc.scene_info(0), params, frame)

No CPU fallback exists: importing `api` without librtr_hip.so raises ImportError, and creating a
Context without a HIP device raises RtrError(RTR_ERR_NO_DEVICE).
"""
__all__ = ["api", "host", "scenes", "_abi"]
