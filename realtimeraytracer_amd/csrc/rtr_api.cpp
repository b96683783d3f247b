/* rtr_api.cpp — implementation of the C ABI in include/rtr.h (host side of librtr_hip.so).
 * HIP runtime for device memory / streams / events; kernels in kernels/rtr_kernels.hip.
 * There is NO CPU rendering fallback in this library: without a HIP device every entry point
 * that needs one fails with RTR_ERR_NO_DEVICE / RTR_ERR_HIP. */
#include "../../include/rtr.h"
#include "../../include/rtr_math.h"
#include "bvh_build.h"
#include "kernels/rtr_kernels.h"
#include "kernels/rtr_post.h"
#include "kernels/rtr_bvh.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using rtrdev::Counters;
using rtrdev::DeviceScene;
using rtrdev::FrameOut;
using rtrdev::RenderArgs;
using rtrdev::Workspace;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? RTR_ERR_OUT_OF_MEMORY : RTR_ERR_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                    \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    hipError_t alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const T* src, size_t count, hipStream_t s) {
        hipError_t e = alloc(count);
        if (e != hipSuccess) return e;
        if (count == 0 || !src) return hipSuccess;
        e = hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(s);
    }
};

}  // namespace

struct rtr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    hipDeviceProp_t prop;
    int numXccs = 0;                     /* hipDeviceAttributeNumberOfXccs (reported by rtr_ctx_device_name): the kernels' eight batch cursors stay private to one XCD each for 8, 4, 2 or 1 XCDs — kernels/rtr_kernels.h, kQueueRegions */
    /* scenes and frames keep a pointer to their context: a context destroyed while it still has children lives on,
     * unusable, until the last child is gone (garbage-collected bindings destroy objects in any order) */
    int children = 0;
    bool destroyed = false;
    rtrdev::Tunables tun;                /* run-time tunables of the staged pipeline: environment at creation, rtr_ctx_set_tunable afterwards */
};

static void ctx_free(rtr_ctx* c) {
    (void)hipSetDevice(c->device);
    if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
static void ctx_release_child(rtr_ctx* c) {
    if (--c->children == 0 && c->destroyed) ctx_free(c);
}

struct rtr_scene {
    rtr_ctx* ctx = nullptr;
    DevBuf<uint4> nodes;                 /* RtrBvhNode, 2 x uint4 each */
    DevBuf<float4> nodesF;               /* rtr::BvhNodeF, 4 x float4 each: device build / refit only */
    DevBuf<RtrBvhGrid> grid;
    DevBuf<unsigned long long> wideSums;     /* scratch of bvh_make_wide */
    DevBuf<uint4> nodes4tmp;             /* the 4-wide entries in BVH2-id order, before the breadth-first permutation */
    DevBuf<uint32_t> wideRemap;
    DevBuf<uint8_t> wideShape;           /* per BVH2 node: which entries its 4-wide record opens (host builder's cost-driven collapse); empty = greedy */
    std::vector<uint8_t> hostWideShape;
    uint32_t wideReached = 0;            /* entries the 4-wide tree reaches (they come first in nodes4) */
    DevBuf<uint4> nodes4;                /* RtrWideNode: 4-wide view of the tree for the any-hit kernel, breadth-first order (kernels/rtr_bvh.hip) */
    DevBuf<float4> tris;
    DevBuf<RtrVertex> vertices;
    DevBuf<uint32_t> indices;
    DevBuf<RtrObjectInfo> objects;
    DevBuf<RtrAreaLightInfo> lights;
    DevBuf<float4> lightTris;            /* 4 x float4 per light triangle (rtrdev::launch_light_tris) */
    DevBuf<uint32_t> lightTriFirst;      /* first record of light l */
    DevBuf<float> xforms, nmats, ltc1, ltc2;
    std::vector<DevBuf<uint8_t>> texPixels;
    DevBuf<uint8_t> hdriPixels;
    DevBuf<rtrdev::DeviceTexture> texTable;
    std::vector<RtrBvhNode> hostNodes;
    std::vector<RtrBvhTri> hostTris;
    std::vector<RtrAreaLightInfo> hostLights;
    /* device build / refit state (kernels/rtr_bvh.hip) */
    std::vector<RtrInstance> hostInstances;
    std::vector<RtrMesh> hostMeshes;
    std::vector<RtrObjectInfo> hostObjects;
    DevBuf<rtrdev::PrimRef> prims;
    DevBuf<rtrdev::InstanceRef> instRefs;
    DevBuf<float4> boxMin, boxMax;
    DevBuf<int32_t> parent;
    DevBuf<uint32_t> counters, depth, slotOfPrim, red;
    uint32_t numPrims = 0, numNodeSlots = 0;
    bool refitReady = false;
    rtr_scene_stats stats{};
    DeviceScene dev{};
    uint32_t numLights = 0, numObjects = 0, numVertices = 0, numIndices = 0;
    bool hasLtc = false;
};

struct rtr_frame {
    rtr_ctx* ctx = nullptr;
    uint32_t width = 0, rows = 0, images = 0;
    DevBuf<uint32_t> img[8];
    uint32_t* ext[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    DevBuf<float4> hdr;
    /* wavefront scratch */
    DevBuf<float4> hitTuvp, rayDT, rayOrigin;      /* hit records; per ray (direction, tmax); per pixel-sample the shadow rays' origin */
    DevBuf<uint32_t> raySlot;                      /* per ray: index of its visibility byte */
    uint32_t slotStride = 0;                       /* distance of the visibility planes: a power of two >= the pixel-sample slots */
    uint32_t visFill = 1;                          /* pre-fill of the visibility array for the next launch: the commoner outcome of the last one (1 = occluded) */
    DevBuf<uint32_t> hitCustom, queueCount;
    DevBuf<uint8_t> vis;
    DevBuf<int32_t> spill;
    DevBuf<uint32_t> overflow;
    uint32_t overflowCap = 0;
    DevBuf<uint2> batchLists;
    DevBuf<unsigned long long> clk;
    uint32_t listStride = 0;
    DevBuf<Counters> counters;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   /* [5]: between the any-hit kernel and k_shadow_tail */
    hipEvent_t evMega[2] = {nullptr, nullptr};
    rtr_frame_stats stats{};
    bool pendingStats = false, pendingWave = false, pendingCounters = false;
    uint32_t pendingImagesK = 0; bool pendingHdr = false, pendingAccum = false;
    hipEvent_t evDone = nullptr;         /* a frame rendered as a later frame of a batch: recorded on the leading frame's stream behind the launch */
    bool viaBatch = false;
    /* ordering between a batch launch (on the leading frame's stream) and the other frames' own streams, paid only when it is needed:
     * batchStream = the stream of the launch that last wrote this frame as a later frame of a batch and has not been joined (evDone
     * marks its end); ownPending = the frame's own stream may still hold work for it (a launch it led); evOwn orders a later batch
     * behind that work */
    hipStream_t batchStream = nullptr;
    bool ownPending = false;
    hipEvent_t evOwn = nullptr;
    /* rtr_render_split_async: the frame as band-shards ("parts") on streams of their own.  A part is an internal frame object — its
     * own scratch, events, counters and context (= stream) — that owns no image: it is bound to THIS frame's images and writes its
     * rows where they belong (RenderArgs::directRows) */
    std::vector<rtr_ctx*> partCtx;
    std::vector<rtr_frame*> parts;
    std::vector<hipEvent_t> evPart;      /* part k's launches are done (recorded on its stream, waited for by this frame's) */
    hipEvent_t evSplit[2] = {nullptr, nullptr};     /* on this frame's stream: the fork, and behind the join — the split render's duration */
    uint32_t pendingSplit = 0;           /* parts of the split render in flight (0: the last render was not split) */
    float4* extHdr = nullptr;            /* a part: the HDR image of the frame it belongs to */
    uint32_t* image_ptr(int which) const { return ext[which] ? ext[which] : img[which].p; }
    float4* hdr_ptr() const { return extHdr ? extHdr : hdr.p; }
};

extern "C" {

const char* rtr_last_error(void) { return g_err.c_str(); }

const char* rtr_status_string(int s) {
    switch (s) {
        case RTR_OK: return "RTR_OK";
        case RTR_ERR_INVALID_ARGUMENT: return "RTR_ERR_INVALID_ARGUMENT";
        case RTR_ERR_HIP: return "RTR_ERR_HIP";
        case RTR_ERR_NO_DEVICE: return "RTR_ERR_NO_DEVICE";
        case RTR_ERR_UNSUPPORTED: return "RTR_ERR_UNSUPPORTED";
        case RTR_ERR_OUT_OF_MEMORY: return "RTR_ERR_OUT_OF_MEMORY";
        case RTR_ERR_BVH_TOO_DEEP: return "RTR_ERR_BVH_TOO_DEEP";
        case RTR_ERR_IO: return "RTR_ERR_IO";
        default: return "RTR_ERR_UNKNOWN";
    }
}

int rtr_abi_version(void) { return RTR_ABI_VERSION; }

const char* rtr_kernel_revision(void) { return RTR_ANYHIT_KERNEL_REVISION; }

uint32_t rtr_shard_rows(uint32_t height, uint32_t bandRows, uint32_t shardCount) {
    if (bandRows == 0) bandRows = 8;
    if (shardCount <= 1) return height;               /* unsharded: no padding rows */
    uint32_t bands = (height + bandRows - 1) / bandRows;
    uint32_t per = (bands + shardCount - 1) / shardCount;
    return per * bandRows;
}

/* ---- context ------------------------------------------------------------------------------ */
static int ctx_create_prio(int ordinal, int priorityRank, rtr_ctx** out);

int rtr_ctx_create(int ordinal, rtr_ctx** out) { return ctx_create_prio(ordinal, -1, out); }

/* priorityRank < 0: a stream of default priority.  >= 0: rank 0 gets the device's highest stream priority, rank 1 the next ... (the
 * parts of a split render: the dispatcher then prefers the workgroups of an earlier part wherever two parts compete for a slot) */
static int ctx_create_prio(int ordinal, int priorityRank, rtr_ctx** out) {
    if (!out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(RTR_ERR_NO_DEVICE, "rtr_ctx_create: no HIP device (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (ordinal < 0 || ordinal >= count) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_create: ordinal %d not in [0,%d)", ordinal, count);
    HIP_TRY(hipSetDevice(ordinal));
    rtr_ctx* c = new rtr_ctx();
    c->device = ordinal;
    e = hipGetDeviceProperties(&c->prop, ordinal);
    if (e != hipSuccess) { delete c; return fail(RTR_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e)); }
    if (hipDeviceGetAttribute(&c->numXccs, hipDeviceAttributeNumberOfXccs, ordinal) != hipSuccess) c->numXccs = 0;
    if (priorityRank >= 0) {
        int least = 0, greatest = 0;                     /* numerically: greatest priority = the smaller number */
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
        int prio = greatest + priorityRank;
        if (prio > least) prio = least;
        e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio);
    } else e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(RTR_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->ownStream = true;
    c->tun = rtrdev::tunables_from_env();
    *out = c;
    return RTR_OK;
}

int rtr_ctx_set_tunable(rtr_ctx* c, const char* name, uint32_t value) {
    if (!c || !name) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_set_tunable: null argument");
    if (!rtrdev::tunable_set(c->tun, name, value)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_set_tunable: no tunable '%s', or %u is outside its range", name, value);
    return RTR_OK;
}

int rtr_ctx_get_tunable(const rtr_ctx* c, const char* name, uint32_t* value) {
    if (!c || !name || !value) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_get_tunable: null argument");
    if (!rtrdev::tunable_get(c->tun, name, value)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_get_tunable: no tunable '%s'", name);
    return RTR_OK;
}

void rtr_ctx_destroy(rtr_ctx* c) {
    if (!c || c->destroyed) return;
    if (c->children > 0) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); c->destroyed = true; return; }
    ctx_free(c);
}

int rtr_ctx_set_stream(rtr_ctx* c, void* s) {
    if (!c) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_set_stream: ctx is null");
    HIP_TRY(hipSetDevice(c->device));
    if (c->ownStream && c->stream) { HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipStreamDestroy(c->stream); }
    if (s) { c->stream = (hipStream_t)s; c->ownStream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->ownStream = true; }
    return RTR_OK;
}

int rtr_ctx_get_stream(rtr_ctx* c, void** out) {
    if (!c || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_get_stream: null argument");
    *out = (void*)c->stream;
    return RTR_OK;
}

int rtr_ctx_device_name(rtr_ctx* c, char* buf, size_t bytes) {
    if (!c || !buf || bytes == 0) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_ctx_device_name: bad argument");
    if (c->numXccs > 0) snprintf(buf, bytes, "%s (%s, %d CUs in %d XCDs)", c->prop.name, c->prop.gcnArchName, c->prop.multiProcessorCount, c->numXccs);
    else snprintf(buf, bytes, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    return RTR_OK;
}

/* ---- scene -------------------------------------------------------------------------------- */
static int validate_desc(const rtr_scene_desc* d) {
    if (!d) return fail(RTR_ERR_INVALID_ARGUMENT, "scene desc is null");
    if ((d->numVertices && !d->vertices) || (d->numIndices && !d->indices) || (d->numMeshes && !d->meshes) ||
        (d->numInstances && !d->instances) || (d->numObjects && !d->objects) || (d->numLights && !d->lights))
        return fail(RTR_ERR_INVALID_ARGUMENT, "scene desc: null array with non-zero count");
    if ((d->ltc1 == nullptr) != (d->ltc2 == nullptr)) return fail(RTR_ERR_INVALID_ARGUMENT, "scene desc: ltc1 and ltc2 must both be given or both null");
    for (uint32_t m = 0; m < d->numMeshes; ++m) {
        const RtrMesh& me = d->meshes[m];
        if (me.indexCount % 3u) return fail(RTR_ERR_INVALID_ARGUMENT, "mesh %u: indexCount %u not a multiple of 3", m, me.indexCount);
        if ((uint64_t)me.indexOffset + me.indexCount > d->numIndices) return fail(RTR_ERR_INVALID_ARGUMENT, "mesh %u: index range exceeds the index array", m);
        if ((uint64_t)me.vertexOffset + me.vertexCount > d->numVertices) return fail(RTR_ERR_INVALID_ARGUMENT, "mesh %u: vertex range exceeds the vertex array", m);
        for (uint32_t i = 0; i < me.indexCount; ++i)
            if (d->indices[me.indexOffset + i] >= me.vertexCount)
                return fail(RTR_ERR_INVALID_ARGUMENT, "mesh %u: index %u (= %u) outside its %u vertices", m, i, d->indices[me.indexOffset + i], me.vertexCount);
    }
    std::vector<uint8_t> seen(d->numInstances, 0);
    for (uint32_t i = 0; i < d->numInstances; ++i) {
        const RtrInstance& in = d->instances[i];
        if (in.meshIndex >= d->numMeshes) return fail(RTR_ERR_INVALID_ARGUMENT, "instance %u: meshIndex %u >= %u", i, in.meshIndex, d->numMeshes);
        if (in.customIndex >= d->numInstances || seen[in.customIndex]) return fail(RTR_ERR_INVALID_ARGUMENT, "instance %u: customIndex %u out of range or duplicated", i, in.customIndex);
        seen[in.customIndex] = 1;
        if (in.customIndex >= d->numLights && in.customIndex - d->numLights >= d->numObjects)
            return fail(RTR_ERR_INVALID_ARGUMENT, "instance %u: customIndex %u has no ObjectInfo (numLights %u, numObjects %u)", i, in.customIndex, d->numLights, d->numObjects);
        if (in.customIndex >= d->numLights) {
            /* the hit shader reads indices / vertices through the ObjectInfo's offsets (closesthit.rchit:59-65), which the
             * reference sets to the mesh's own when it builds the TLAS (tlas.cppm:58-71): anything else would fetch another
             * mesh's data, or none */
            const RtrObjectInfo& oi = d->objects[in.customIndex - d->numLights];
            const RtrMesh& me = d->meshes[in.meshIndex];
            if (oi.vertexOffset != me.vertexOffset || oi.indexOffset != me.indexOffset)
                return fail(RTR_ERR_INVALID_ARGUMENT, "instance %u: ObjectInfo offsets (%u, %u) differ from its mesh's (%u, %u)", i, oi.vertexOffset, oi.indexOffset,
                            me.vertexOffset, me.indexOffset);
        }
        for (int k = 0; k < 12; ++k)
            if (!(in.transform[k] == in.transform[k]) || in.transform[k] > 3.0e38f || in.transform[k] < -3.0e38f)
                return fail(RTR_ERR_INVALID_ARGUMENT, "instance %u: non-finite transform", i);
    }
    if (d->numLights > d->numInstances) return fail(RTR_ERR_INVALID_ARGUMENT, "numLights %u > numInstances %u (lights are the first instances)", d->numLights, d->numInstances);
    if (d->numTextures && !d->textures) return fail(RTR_ERR_INVALID_ARGUMENT, "scene desc: null texture array with non-zero count");
    auto tex_ok = [&](const rtr_texture& t) { return t.pixels && t.width > 0 && t.height > 0 && t.width <= 65536 && t.height <= 65536 && (t.channels == 1 || t.channels == 4); };
    for (uint32_t t = 0; t < d->numTextures; ++t)
        if (d->textures[t].pixels && !tex_ok(d->textures[t])) return fail(RTR_ERR_INVALID_ARGUMENT, "texture %u: bad extent %ux%u or channels %u (1 or 4)", t, d->textures[t].width, d->textures[t].height, d->textures[t].channels);
    if (d->hdri && !tex_ok(*d->hdri)) return fail(RTR_ERR_INVALID_ARGUMENT, "hdri: bad extent or channels");
    for (uint32_t o = 0; o < d->numObjects; ++o) {
        const RtrObjectInfo& oi = d->objects[o];
        const struct { uint32_t uses, index; const char* what; } maps[4] = {{oi.usesColorMap, oi.colorIndex, "color"}, {oi.usesSpecularMap, oi.specularIndex, "specular"},
                                                                            {oi.usesMetallicMap, oi.metallicIndex, "metallic"}, {oi.usesOpacityMap, oi.opacityIndex, "opacity"}};
        for (const auto& m : maps)
            if (m.uses && (m.index >= d->numTextures || !d->textures[m.index].pixels))
                return fail(RTR_ERR_INVALID_ARGUMENT, "object %u uses a %s map but texture index %u is not in the texture array (%u entries); "
                            "the library never substitutes a constant for a missing texture", o, m.what, m.index, d->numTextures);
    }
    for (uint32_t l = 0; l < d->numLights; ++l) {
        const RtrAreaLightInfo& li = d->lights[l];
        if ((uint64_t)li.indexOffset + 3ull * li.numTriangles > d->numIndices) return fail(RTR_ERR_INVALID_ARGUMENT, "light %u: triangle range exceeds the index array", l);
        for (uint32_t i = 0; i < 3u * li.numTriangles; ++i)
            if ((uint64_t)li.vertexOffset + d->indices[li.indexOffset + i] >= d->numVertices)
                return fail(RTR_ERR_INVALID_ARGUMENT, "light %u: vertex reference outside the vertex array", l);
    }
    return RTR_OK;
}

/* flatten TLAS instances to one world-space triangle soup (instance order, then primitive order),
 * fill the per-customIndex transform tables, build the BVH */
static int flatten_and_build(const rtr_scene_desc* d, rtr::BvhResult& bvh, std::vector<float>& xforms, std::vector<float>& nmats,
                             uint32_t* stackEntries, size_t* numTris) {
    std::vector<rtr::WorldTriangle> soup;
    size_t total = 0;
    for (uint32_t i = 0; i < d->numInstances; ++i) total += d->meshes[d->instances[i].meshIndex].indexCount / 3u;
    soup.reserve(total);
    xforms.assign(12 * (size_t)d->numInstances, 0.f);
    nmats.assign(12 * (size_t)d->numInstances, 0.f);
    for (uint32_t i = 0; i < d->numInstances; ++i) {
        const RtrInstance& in = d->instances[i];
        const RtrMesh& me = d->meshes[in.meshIndex];
        memcpy(&xforms[12 * (size_t)in.customIndex], in.transform, 12 * sizeof(float));
        rtr_normal_matrix(in.transform, &nmats[12 * (size_t)in.customIndex]);
        for (uint32_t t = 0; t < me.indexCount / 3u; ++t) {
            rtr::WorldTriangle w;
            for (int k = 0; k < 3; ++k) {
                const uint32_t idx = d->indices[me.indexOffset + 3u * t + k] + me.vertexOffset;
                const rtr_v3 p = rtr_xform_point34(in.transform, rtr_ld3(d->vertices[idx].position));
                w.v[k][0] = p.x; w.v[k][1] = p.y; w.v[k][2] = p.z;
            }
            w.customIndex = in.customIndex; w.primitiveId = t;
            /* any-hit (opacity.rahit) runs only on non-opaque geometry (blas.cppm:98-100) of objects with an opacity map */
            w.flags = (in.customIndex >= d->numLights && d->objects[in.customIndex - d->numLights].usesOpacityMap != 0 && me.isOpaque == 0) ? 1u : 0u;
            soup.push_back(w);
        }
    }
    std::string err;
    if (!rtr::build_bvh(soup, bvh, &err)) return fail(RTR_ERR_INVALID_ARGUMENT, "BVH build: %s", err.c_str());
    if (bvh.maxDepth > 64)
        return fail(RTR_ERR_BVH_TOO_DEEP, "BVH depth %u exceeds the 64-entry LDS traversal stack", bvh.maxDepth);
    *stackEntries = bvh.maxDepth <= 16 ? 16 : (bvh.maxDepth <= 32 ? 32 : 64);
    *numTris = soup.size();
    return RTR_OK;
}

static void fill_stats(rtr_scene_stats& st, const rtr::BvhResult& bvh, uint32_t stackEntries, size_t numTris) {
    memset(&st, 0, sizeof st);
    st.numTriangles = (uint32_t)numTris;
    st.numNodes = (uint32_t)bvh.nodes.size();
    st.maxDepth = bvh.maxDepth;
    st.maxLeafSize = bvh.maxLeafSize;
    st.bvhLayoutVersion = RTR_BVH_LAYOUT_VERSION;
    st.stackEntries = stackEntries;
    st.buildMs = bvh.buildMs;
    st.sahCost = bvh.sahCost;
    st.grid = bvh.grid;
    for (int k = 0; k < 3; ++k) { st.boundsMin[k] = bvh.boundsMin[k]; st.boundsMax[k] = bvh.boundsMax[k]; }
    st.boxPad = bvh.boxPad;
}

int rtr_host_build_bvh(const rtr_scene_desc* d, rtr_scene_stats* stats, RtrBvhNode* nodes, size_t nodeBytes, RtrBvhTri* tris, size_t triBytes) {
    return rtr_host_build_bvh_wide(d, stats, nodes, nodeBytes, tris, triBytes, nullptr, 0);
}

int rtr_host_build_bvh_wide(const rtr_scene_desc* d, rtr_scene_stats* stats, RtrBvhNode* nodes, size_t nodeBytes, RtrBvhTri* tris, size_t triBytes,
                            RtrWideNode* wide, size_t wideBytes) {
    int rc = validate_desc(d);
    if (rc != RTR_OK) return rc;
    rtr::BvhResult bvh; std::vector<float> xf, nm; uint32_t stackEntries = 0; size_t numTris = 0;
    rc = flatten_and_build(d, bvh, xf, nm, &stackEntries, &numTris);
    if (rc != RTR_OK) return rc;
    if (stats) fill_stats(*stats, bvh, stackEntries, numTris);
    if (wide || (stats && !nodes && !tris)) {
        std::vector<RtrWideNode> w;
        rtr::make_wide_host(bvh.nodes.data(), bvh.nodes.size(), bvh.wideShape.size() == bvh.nodes.size() ? bvh.wideShape.data() : nullptr, bvh.grid, w);
        if (stats) { stats->grid = bvh.grid; stats->numWideNodes = (uint32_t)w.size(); stats->wideLayoutVersion = RTR_WIDE_LAYOUT_VERSION; }
        if (wide) {
            if (wideBytes != w.size() * sizeof(RtrWideNode)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_host_build_bvh_wide: wideBytes %zu != %zu", wideBytes, w.size() * sizeof(RtrWideNode));
            memcpy(wide, w.data(), wideBytes);
        }
    }
    if (nodes) {
        if (nodeBytes != bvh.nodes.size() * sizeof(RtrBvhNode)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_host_build_bvh: nodeBytes %zu != %zu", nodeBytes, bvh.nodes.size() * sizeof(RtrBvhNode));
        memcpy(nodes, bvh.nodes.data(), nodeBytes);
    }
    if (tris) {
        if (triBytes != bvh.tris.size() * sizeof(RtrBvhTri)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_host_build_bvh: triBytes %zu != %zu", triBytes, bvh.tris.size() * sizeof(RtrBvhTri));
        memcpy(tris, bvh.tris.data(), triBytes);
    }
    return RTR_OK;
}

/* canonical (instance-major) primitive table + per-customIndex instance table for the device flatten kernel */
static void make_prim_tables(const rtr_scene_desc* d, const RtrInstance* instances, std::vector<rtrdev::PrimRef>& prims,
                             std::vector<rtrdev::InstanceRef>& refs) {
    prims.clear();
    refs.assign(d->numInstances, rtrdev::InstanceRef{});
    for (uint32_t i = 0; i < d->numInstances; ++i) {
        const RtrInstance& in = instances[i];
        const RtrMesh& me = d->meshes[in.meshIndex];
        rtrdev::InstanceRef& r = refs[in.customIndex];
        memcpy(r.transform, in.transform, sizeof r.transform);
        r.vertexOffset = me.vertexOffset; r.indexOffset = me.indexOffset;
        const uint32_t flags = (in.customIndex >= d->numLights && d->objects[in.customIndex - d->numLights].usesOpacityMap != 0 && me.isOpaque == 0) ? 1u : 0u;
        for (uint32_t t = 0; t < me.indexCount / 3u; ++t) prims.push_back(rtrdev::PrimRef{in.customIndex, t, flags, 0u});
    }
}

/* (re)builds the 4-wide view of the tree the any-hit kernel walks, on the device, from the quantised BVH2 nodes */
/* (re)computes the per-light-triangle records light_loops reads; the light transforms live in s->lights on the device */
static int make_light_tris(rtr_scene* s) {
    if (s->numLights == 0) return RTR_OK;
    hipError_t e = rtrdev::launch_light_tris(s->lights.p, s->vertices.p, s->indices.p, s->lightTriFirst.p, s->numLights, s->lightTris.p, s->ctx->stream);
    if (e != hipSuccess) return fail(RTR_ERR_HIP, "light-triangle records: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));     /* frames of other contexts may render this scene next */
    return RTR_OK;
}

static int make_wide_nodes(rtr_scene* s) {
    const uint32_t n = (uint32_t)s->hostNodes.size();
    if (!s->nodes4.p) { HIP_TRY(s->nodes4.alloc((size_t)n * 4)); HIP_TRY(s->nodes4tmp.alloc((size_t)n * 4)); HIP_TRY(s->wideRemap.alloc(n)); HIP_TRY(s->wideSums.alloc(rtrdev::bvh_wide_scratch_words())); }
    hipStream_t st = s->ctx->stream;
    if (!s->hostWideShape.empty() && !s->wideShape.p) HIP_TRY(s->wideShape.upload(s->hostWideShape.data(), s->hostWideShape.size(), st));
    hipError_t e = rtrdev::bvh_make_wide(s->nodes.p, n, s->refitReady ? s->parent.p : nullptr, s->grid.p, s->hostWideShape.size() == n ? s->wideShape.p : nullptr, s->nodes4tmp.p, s->wideSums.p, st);
    if (e != hipSuccess) return fail(RTR_ERR_HIP, "4-wide node build: %s", hipGetErrorString(e));
    /* breadth-first order of the 4-wide tree (child codes = the 4th 16 bytes of every entry), so its top levels are the first
     * entries: k_shadow_trace4 keeps those in LDS.  Entries the 4-wide tree does not reach keep the ids after them. */
    std::vector<uint32_t> codes((size_t)n * 4), remap(n, 0xffffffffu), order;
    HIP_TRY(hipMemcpy2DAsync(codes.data(), 16, reinterpret_cast<const char*>(s->nodes4tmp.p) + 48, 64, 16, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    order.reserve(n);
    order.push_back(0); remap[0] = 0;
    for (size_t head = 0; head < order.size(); ++head)
        for (int k = 0; k < 4; ++k) {
            const int32_t c = (int32_t)codes[(size_t)order[head] * 4 + k];
            if (c >= 0 && (uint32_t)c < n && remap[c] == 0xffffffffu) { remap[c] = (uint32_t)order.size(); order.push_back((uint32_t)c); }
        }
    uint32_t next = (uint32_t)order.size();
    for (uint32_t i = 0; i < n; ++i) if (remap[i] == 0xffffffffu) remap[i] = next++;
    HIP_TRY(hipMemcpyAsync(s->wideRemap.p, remap.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    e = rtrdev::bvh_permute_wide(s->nodes4tmp.p, n, s->wideRemap.p, s->nodes4.p, st);
    if (e != hipSuccess) return fail(RTR_ERR_HIP, "4-wide node order: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(st));
    s->wideReached = (uint32_t)order.size();
    HIP_TRY(hipMemcpy(&s->stats.grid, s->grid.p, sizeof(RtrBvhGrid), hipMemcpyDeviceToHost));     /* the wide centre was set on the device */
    s->stats.numWideNodes = s->wideReached; s->stats.wideLayoutVersion = RTR_WIDE_LAYOUT_VERSION;
    return RTR_OK;
}

static rtrdev::BvhDeviceArrays device_arrays(rtr_scene* s) {
    rtrdev::BvhDeviceArrays a{};
    a.nodes = s->nodes.p; a.nodesF = s->nodesF.p; a.grid = s->grid.p; a.tris = s->tris.p; a.boxMin = s->boxMin.p; a.boxMax = s->boxMax.p; a.parent = s->parent.p;
    a.counters = s->counters.p; a.depth = s->depth.p; a.slotOfPrim = s->slotOfPrim.p; a.red = s->red.p;
    return a;
}

/* Device LBVH build into s->nodes / s->tris (+ refit arrays); fills hostNodes/hostTris and the stats. */
static int build_on_device(rtr_scene* s, const rtr_scene_desc* d, size_t numPrims) {
    hipStream_t st = s->ctx->stream;
    std::vector<rtrdev::PrimRef> prims; std::vector<rtrdev::InstanceRef> refs;
    make_prim_tables(d, d->instances, prims, refs);
    const uint32_t n = (uint32_t)numPrims, numNodes = n - 1;
    for (uint32_t v = 0; v < d->numVertices; ++v)
        for (int k = 0; k < 3; ++k)
            if (!(d->vertices[v].position[k] > -3.0e38f && d->vertices[v].position[k] < 3.0e38f))
                return fail(RTR_ERR_INVALID_ARGUMENT, "BVH build: non-finite vertex position in vertex %u", v);
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(s->prims.upload(prims.data(), prims.size(), st));
    HIP_TRY(s->instRefs.upload(refs.data(), refs.size(), st));
    HIP_TRY(s->nodes.alloc((size_t)numNodes * 2)); HIP_TRY(s->nodesF.alloc((size_t)numNodes * 4)); HIP_TRY(s->grid.alloc(1));
    HIP_TRY(s->tris.alloc((size_t)n * 3));
    HIP_TRY(s->boxMin.alloc(n)); HIP_TRY(s->boxMax.alloc(n)); HIP_TRY(s->parent.alloc(numNodes));
    HIP_TRY(s->counters.alloc(numNodes)); HIP_TRY(s->depth.alloc(numNodes)); HIP_TRY(s->slotOfPrim.alloc(n)); HIP_TRY(s->red.alloc(8));
    DevBuf<float4> trisCanon, minCanon, maxCanon; DevBuf<unsigned long long> keysIn, keysOut; DevBuf<int2> range, rawChild; DevBuf<uint8_t> sortTemp;
    HIP_TRY(trisCanon.alloc((size_t)n * 3)); HIP_TRY(minCanon.alloc(n)); HIP_TRY(maxCanon.alloc(n));
    HIP_TRY(keysIn.alloc(n)); HIP_TRY(keysOut.alloc(n)); HIP_TRY(range.alloc(numNodes)); HIP_TRY(rawChild.alloc(numNodes));
    rtrdev::BvhScratch sc{};
    sc.sortTempBytes = rtrdev::bvh_sort_temp_bytes(n);
    HIP_TRY(sortTemp.alloc(sc.sortTempBytes));
    sc.trisCanon = trisCanon.p; sc.minCanon = minCanon.p; sc.maxCanon = maxCanon.p; sc.keysIn = keysIn.p; sc.keysOut = keysOut.p;
    sc.range = range.p; sc.rawChild = rawChild.p; sc.sortTemp = sortTemp.p;
    HIP_TRY(hipMemsetAsync(s->nodesF.p, 0, (size_t)numNodes * 64, st));
    rtrdev::BvhInputs in{s->prims.p, s->instRefs.p, s->vertices.p, s->indices.p};
    hipError_t e = rtrdev::bvh_build_lbvh(in, n, device_arrays(s), sc, st);
    if (e != hipSuccess) return fail(RTR_ERR_HIP, "device BVH build: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(st));
    s->hostNodes.resize(numNodes); s->hostTris.resize(n);
    uint32_t red[8];
    HIP_TRY(hipMemcpy(s->hostNodes.data(), s->nodes.p, (size_t)numNodes * sizeof(RtrBvhNode), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(s->hostTris.data(), s->tris.p, (size_t)n * 48, hipMemcpyDeviceToHost));
    RtrBvhGrid grid;
    HIP_TRY(hipMemcpy(&grid, s->grid.p, sizeof grid, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(red, s->red.p, sizeof red, hipMemcpyDeviceToHost));
    const float buildMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (red[7] > 64) return fail(RTR_ERR_BVH_TOO_DEEP, "device-built BVH depth %u exceeds the 64-entry LDS traversal stack", red[7]);
    memset(&s->stats, 0, sizeof s->stats);
    s->stats.numTriangles = n; s->stats.numNodes = numNodes; s->stats.maxDepth = red[7]; s->stats.maxLeafSize = 4;
    s->stats.bvhLayoutVersion = RTR_BVH_LAYOUT_VERSION;
    s->stats.stackEntries = red[7] <= 16 ? 16 : (red[7] <= 32 ? 32 : 64);
    s->stats.buildMs = buildMs;
    s->stats.grid = grid;
    float mabs; memcpy(&mabs, &red[6], 4);
    s->stats.boxPad = (mabs > 1e-6f ? mabs : 1e-6f) * 3.814697265625e-06f;
    s->numPrims = n; s->numNodeSlots = numNodes; s->refitReady = true;
    return RTR_OK;
}

static void instance_tables(uint32_t numInstances, const RtrInstance* instances, std::vector<float>& xforms, std::vector<float>& nmats) {
    xforms.assign(12 * (size_t)numInstances, 0.f);
    nmats.assign(12 * (size_t)numInstances, 0.f);
    for (uint32_t i = 0; i < numInstances; ++i) {
        memcpy(&xforms[12 * (size_t)instances[i].customIndex], instances[i].transform, 12 * sizeof(float));
        rtr_normal_matrix(instances[i].transform, &nmats[12 * (size_t)instances[i].customIndex]);
    }
}

static int scene_create_impl(rtr_ctx* ctx, const rtr_scene_desc* d, const rtr_scene* like, rtr_scene** out);

int rtr_scene_create(rtr_ctx* ctx, const rtr_scene_desc* d, rtr_scene** out) { return scene_create_impl(ctx, d, nullptr, out); }

int rtr_scene_create_like(rtr_ctx* ctx, const rtr_scene_desc* d, const rtr_scene* built, rtr_scene** out) {
    if (!built) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_create_like: null scene to copy the tree from");
    return scene_create_impl(ctx, d, built, out);
}

static int scene_create_impl(rtr_ctx* ctx, const rtr_scene_desc* d, const rtr_scene* like, rtr_scene** out) {
    if (!ctx || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_create: null ctx/out");
    if (ctx->destroyed) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_create: the context has been destroyed");
    *out = nullptr;
    int rc = validate_desc(d);
    if (rc != RTR_OK) return rc;
    if (d->buildFlags > RTR_BUILD_DEVICE_LBVH) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_create: unknown buildFlags %u", d->buildFlags);
    HIP_TRY(hipSetDevice(ctx->device));
    size_t totalPrims = 0;
    for (uint32_t i = 0; i < d->numInstances; ++i) totalPrims += d->meshes[d->instances[i].meshIndex].indexCount / 3u;
    /* every triangle may end up in a leaf of its own: numTriangles - 1 inner nodes, hence as many 4-wide records at most */
    rc = rtr_check_scene_limits(totalPrims, totalPrims ? totalPrims - 1 : 0);
    if (rc != RTR_OK) return rc;
    /* tiny scenes always take the host builder (the radix tree needs a root with more than one leaf's worth of primitives) */
    const bool deviceBuild = !like && d->buildFlags == RTR_BUILD_DEVICE_LBVH && totalPrims >= 16;
    rtr::BvhResult bvh; std::vector<float> xforms, nmats; uint32_t stackEntries = 0; size_t numTris = 0;
    if (like) {
        /* the tree of `like`, as its host copy holds it (nodes and records are kept in step with the device by every update) */
        if (like->stats.numTriangles != totalPrims || like->hostNodes.empty() || like->hostTris.empty())
            return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_create_like: the built scene has %u triangles, this description %zu", like->stats.numTriangles, totalPrims);
        bvh.nodes = like->hostNodes; bvh.tris = like->hostTris; bvh.grid = like->stats.grid; bvh.wideShape = like->hostWideShape;
        bvh.maxDepth = like->stats.maxDepth; bvh.maxLeafSize = like->stats.maxLeafSize; bvh.sahCost = like->stats.sahCost; bvh.boxPad = like->stats.boxPad;
        for (int k = 0; k < 3; ++k) { bvh.boundsMin[k] = like->stats.boundsMin[k]; bvh.boundsMax[k] = like->stats.boundsMax[k]; }
        bvh.buildMs = 0.f;                                   /* nothing was built here */
        stackEntries = like->stats.stackEntries; numTris = like->stats.numTriangles;
        instance_tables(d->numInstances, d->instances, xforms, nmats);
    } else if (!deviceBuild) {
        rc = flatten_and_build(d, bvh, xforms, nmats, &stackEntries, &numTris);
        if (rc != RTR_OK) return rc;
    } else {
        instance_tables(d->numInstances, d->instances, xforms, nmats);
    }

    rtr_scene* s = new rtr_scene();
    s->ctx = ctx; ++ctx->children;
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    static_assert(sizeof(RtrBvhNode) == 2 * sizeof(uint4) && sizeof(RtrBvhTri) == 3 * sizeof(float4), "layout");
    if (!deviceBuild) {
        chk(s->nodes.upload(reinterpret_cast<const uint4*>(bvh.nodes.data()), bvh.nodes.size() * 2, st));
        chk(s->grid.upload(&bvh.grid, 1, st));
        chk(s->tris.upload(reinterpret_cast<const float4*>(bvh.tris.data()), bvh.tris.size() * 3, st));
    }
    chk(s->vertices.upload(d->vertices, d->numVertices, st));
    chk(s->indices.upload(d->indices, d->numIndices, st));
    chk(s->objects.upload(d->objects, d->numObjects, st));
    chk(s->lights.upload(d->lights, d->numLights, st));
    {
        std::vector<uint32_t> first(d->numLights);
        uint32_t total = 0;
        for (uint32_t l = 0; l < d->numLights; ++l) { first[l] = total; total += d->lights[l].numTriangles; }
        chk(s->lightTriFirst.upload(first.data(), first.size(), st));
        chk(s->lightTris.alloc((size_t)total * rtrdev::kLightTriRecord));
    }
    chk(s->xforms.upload(xforms.data(), xforms.size(), st));
    chk(s->nmats.upload(nmats.data(), nmats.size(), st));
    if (d->ltc1) {
        chk(s->ltc1.upload(d->ltc1, 64 * 64 * 4, st));
        chk(s->ltc2.upload(d->ltc2, 64 * 64 * 4, st));
        s->hasLtc = true;
    }
    std::vector<rtrdev::DeviceTexture> table(d->numTextures);
    s->texPixels.resize(d->numTextures);
    for (uint32_t t = 0; t < d->numTextures; ++t) {
        const rtr_texture& tx = d->textures[t];
        table[t] = rtrdev::DeviceTexture{nullptr, 0, 0, 0, 0};
        if (!tx.pixels) continue;
        chk(s->texPixels[t].upload(tx.pixels, (size_t)tx.width * tx.height * tx.channels, st));
        table[t] = rtrdev::DeviceTexture{s->texPixels[t].p, tx.width, tx.height, tx.channels, 0};
    }
    chk(s->texTable.upload(table.data(), table.size(), st));
    rtrdev::DeviceTexture hdri{nullptr, 0, 0, 0, 0};
    if (d->hdri) {
        chk(s->hdriPixels.upload(d->hdri->pixels, (size_t)d->hdri->width * d->hdri->height * d->hdri->channels, st));
        hdri = rtrdev::DeviceTexture{s->hdriPixels.p, d->hdri->width, d->hdri->height, d->hdri->channels, 0};
    }
    if (e != hipSuccess) {
        delete s; ctx_release_child(ctx);
        return fail(e == hipErrorOutOfMemory ? RTR_ERR_OUT_OF_MEMORY : RTR_ERR_HIP, "scene upload: %s", hipGetErrorString(e));
    }
    s->numLights = d->numLights; s->numObjects = d->numObjects; s->numVertices = d->numVertices; s->numIndices = d->numIndices;
    if (d->numLights) s->hostLights.assign(d->lights, d->lights + d->numLights);
    if (d->numInstances) s->hostInstances.assign(d->instances, d->instances + d->numInstances);
    if (d->numMeshes) s->hostMeshes.assign(d->meshes, d->meshes + d->numMeshes);
    if (d->numObjects) s->hostObjects.assign(d->objects, d->objects + d->numObjects);
    if (deviceBuild) {
        rc = build_on_device(s, d, totalPrims);
        if (rc != RTR_OK) { delete s; ctx_release_child(ctx); return rc; }
    } else {
        fill_stats(s->stats, bvh, stackEntries, numTris);
        s->hostNodes.swap(bvh.nodes);
        s->hostTris.swap(bvh.tris);
        s->hostWideShape.swap(bvh.wideShape);
        s->numPrims = (uint32_t)s->hostTris.size(); s->numNodeSlots = (uint32_t)s->hostNodes.size();
    }

    rc = make_wide_nodes(s);
    if (rc == RTR_OK) rc = make_light_tris(s);
    if (rc != RTR_OK) { delete s; ctx_release_child(ctx); return rc; }
    DeviceScene& dv = s->dev;
    dv.nodes = s->nodes.p; dv.nodes4 = s->nodes4.p; dv.numNodes4 = (uint32_t)s->hostNodes.size(); dv.grid = s->grid.p; dv.tris = s->tris.p;
    dv.vertices = s->vertices.p; dv.indices = s->indices.p;
    dv.objects = s->objects.p; dv.lights = s->lights.p;
    dv.lightTris = s->lightTris.p; dv.lightTriFirst = s->lightTriFirst.p;
    dv.xforms = s->xforms.p; dv.nmats = s->nmats.p;
    dv.ltc1 = s->hasLtc ? s->ltc1.p : nullptr; dv.ltc2 = s->hasLtc ? s->ltc2.p : nullptr;
    for (int k = 0; k < 3; ++k) dv.skyLinear[k] = rtr_to_linear(d->skyColor[k]);
    dv.numLights = d->numLights;
    dv.numLightTris = 0;
    for (uint32_t l = 0; l < d->numLights; ++l) dv.numLightTris += d->lights[l].numTriangles;
    dv.textures = s->texTable.p;
    dv.hdri = hdri;
    *out = s;
    return RTR_OK;
}

/* A host-built tree gets its refit arrays on the first update: parent links from the node array, the
 * canonical-primitive -> leaf-slot map from the ids stored in the triangle records. */
static int ensure_refit_ready(rtr_scene* s) {
    if (s->refitReady) return RTR_OK;
    if (s->hostInstances.empty()) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_instances: the scene has no instances");
    hipStream_t st = s->ctx->stream;
    const uint32_t numNodes = (uint32_t)s->hostNodes.size(), n = (uint32_t)s->hostTris.size();
    std::vector<int32_t> parent(numNodes, -2);
    parent[0] = -1;
    std::vector<uint32_t> stack{0};
    while (!stack.empty()) {
        const uint32_t i = stack.back(); stack.pop_back();
        for (int sl = 0; sl < 2; ++sl) {
            const int32_t c = s->hostNodes[i].child[sl];
            if (c >= 0 && parent[(size_t)c] == -2) { parent[(size_t)c] = (int32_t)((i << 1) | (uint32_t)sl); stack.push_back((uint32_t)c); }
        }
    }
    /* canonical order = instances in creation order, primitives in mesh order */
    std::vector<uint32_t> base(s->hostInstances.size(), 0);       /* by customIndex */
    uint32_t acc = 0;
    for (const RtrInstance& in : s->hostInstances) { base[in.customIndex] = acc; acc += s->hostMeshes[in.meshIndex].indexCount / 3u; }
    if (acc != n) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_instances: scene has no geometry to refit");
    std::vector<uint32_t> slotOfPrim(n, 0);
    for (uint32_t slot = 0; slot < n; ++slot) slotOfPrim[base[s->hostTris[slot].customIndex] + s->hostTris[slot].primitiveId] = slot;
    /* the fit works on fp32 planes: child codes from the host tree, boxes recomputed by the refit */
    std::vector<rtr::BvhNodeF> nf(numNodes);
    for (uint32_t i = 0; i < numNodes; ++i) {
        memset(&nf[i], 0, sizeof nf[i]);
        nf[i].child[0] = s->hostNodes[i].child[0]; nf[i].child[1] = s->hostNodes[i].child[1];
    }
    HIP_TRY(s->nodesF.upload(reinterpret_cast<const float4*>(nf.data()), (size_t)numNodes * 4, st));
    HIP_TRY(s->parent.upload(parent.data(), parent.size(), st));
    HIP_TRY(s->slotOfPrim.upload(slotOfPrim.data(), slotOfPrim.size(), st));
    HIP_TRY(s->boxMin.alloc(n)); HIP_TRY(s->boxMax.alloc(n));
    HIP_TRY(s->counters.alloc(numNodes)); HIP_TRY(s->depth.alloc(numNodes)); HIP_TRY(s->red.alloc(8));
    s->numPrims = n; s->numNodeSlots = numNodes; s->refitReady = true;
    return RTR_OK;
}

int rtr_scene_update_instances(rtr_scene* s, const RtrInstance* instances, uint32_t numInstances, const RtrAreaLightInfo* lights, uint32_t numLights) {
    if (!s || (!instances && numInstances)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_instances: null argument");
    if (numInstances != s->hostInstances.size()) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_instances: %u instances given, scene has %zu", numInstances, s->hostInstances.size());
    if (lights && numLights != s->numLights) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_instances: %u lights given, scene has %u", numLights, s->numLights);
    for (uint32_t i = 0; i < numInstances; ++i) {
        if (instances[i].meshIndex != s->hostInstances[i].meshIndex || instances[i].customIndex != s->hostInstances[i].customIndex)
            return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_instances: instance %u changed mesh or customIndex; only transforms may change (a refit keeps the topology)", i);
        for (int k = 0; k < 12; ++k)
            if (!(instances[i].transform[k] > -3.0e38f && instances[i].transform[k] < 3.0e38f))
                return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_instances: instance %u has a non-finite transform", i);
    }
    if (lights)
        for (uint32_t l = 0; l < numLights; ++l)
            if (lights[l].vertexOffset != s->hostLights[l].vertexOffset || lights[l].indexOffset != s->hostLights[l].indexOffset ||
                lights[l].numTriangles != s->hostLights[l].numTriangles)
                return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_instances: light %u changed its mesh", l);
    if (s->hostTris.empty() || s->hostTris[0].customIndex == 0xffffffffu) return RTR_OK;     /* empty scene: nothing to refit */
    HIP_TRY(hipSetDevice(s->ctx->device));
    int rc = ensure_refit_ready(s);
    if (rc != RTR_OK) return rc;
    /* frames of OTHER contexts (other streams) may be rendering this scene: everything enqueued on the device so far is joined
     * before the nodes, records and light tables are rewritten (contract in rtr.h) */
    HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = s->ctx->stream;
    rtr_scene_desc view{};
    view.meshes = s->hostMeshes.data(); view.numMeshes = (uint32_t)s->hostMeshes.size();
    view.numInstances = numInstances; view.objects = s->hostObjects.data(); view.numObjects = (uint32_t)s->hostObjects.size();
    view.numLights = s->numLights;
    std::vector<rtrdev::PrimRef> prims; std::vector<rtrdev::InstanceRef> refs;
    make_prim_tables(&view, instances, prims, refs);
    HIP_TRY(s->prims.upload(prims.data(), prims.size(), st));
    HIP_TRY(s->instRefs.upload(refs.data(), refs.size(), st));
    std::vector<float> xforms, nmats;
    instance_tables(numInstances, instances, xforms, nmats);
    HIP_TRY(hipMemcpyAsync(s->xforms.p, xforms.data(), xforms.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->nmats.p, nmats.data(), nmats.size() * sizeof(float), hipMemcpyHostToDevice, st));
    if (lights && numLights) HIP_TRY(hipMemcpyAsync(s->lights.p, lights, numLights * sizeof(RtrAreaLightInfo), hipMemcpyHostToDevice, st));
    rtrdev::BvhInputs in{s->prims.p, s->instRefs.p, s->vertices.p, s->indices.p};
    hipError_t e = rtrdev::bvh_refit(in, s->numPrims, s->numNodeSlots, device_arrays(s), st);
    if (e != hipSuccess) return fail(RTR_ERR_HIP, "device BVH refit: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(st));
    /* keep the host mirror (rtr_scene_export_bvh) and the stats in step */
    uint32_t red[8];
    HIP_TRY(hipMemcpy(s->hostNodes.data(), s->nodes.p, s->hostNodes.size() * sizeof(RtrBvhNode), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(s->hostTris.data(), s->tris.p, s->hostTris.size() * 48, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&s->stats.grid, s->grid.p, sizeof(RtrBvhGrid), hipMemcpyDeviceToHost));
    { const int rc4 = make_wide_nodes(s); if (rc4 != RTR_OK) return rc4; }
    if (lights && numLights) { const int rcl = make_light_tris(s); if (rcl != RTR_OK) return rcl; }
    HIP_TRY(hipMemcpy(red, s->red.p, sizeof red, hipMemcpyDeviceToHost));
    float mabs; memcpy(&mabs, &red[6], 4);
    s->stats.boxPad = (mabs > 1e-6f ? mabs : 1e-6f) * 3.814697265625e-06f;
    s->hostInstances.assign(instances, instances + numInstances);
    if (lights && numLights) s->hostLights.assign(lights, lights + numLights);
    return RTR_OK;
}

void rtr_scene_destroy(rtr_scene* s) {
    if (!s) return;
    rtr_ctx* c = s->ctx;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    delete s;
    ctx_release_child(c);
}

int rtr_scene_get_stats(const rtr_scene* s, rtr_scene_stats* out) {
    if (!s || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_get_stats: null argument");
    *out = s->stats;
    return RTR_OK;
}

int rtr_scene_export_bvh(const rtr_scene* s, RtrBvhNode* nodes, size_t nodeBytes, RtrBvhTri* tris, size_t triBytes) {
    if (!s) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_export_bvh: null scene");
    if (nodes) {
        if (nodeBytes != s->hostNodes.size() * sizeof(RtrBvhNode)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_export_bvh: nodeBytes %zu != %zu", nodeBytes, s->hostNodes.size() * sizeof(RtrBvhNode));
        memcpy(nodes, s->hostNodes.data(), nodeBytes);
    }
    if (tris) {
        if (triBytes != s->hostTris.size() * sizeof(RtrBvhTri)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_export_bvh: triBytes %zu != %zu", triBytes, s->hostTris.size() * sizeof(RtrBvhTri));
        memcpy(tris, s->hostTris.data(), triBytes);
    }
    return RTR_OK;
}

int rtr_scene_export_wide(const rtr_scene* s, RtrWideNode* nodes, size_t nodeBytes) {
    if (!s || !nodes) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_export_wide: null argument");
    if (nodeBytes != (size_t)s->wideReached * sizeof(RtrWideNode)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_export_wide: nodeBytes %zu != %zu", nodeBytes, (size_t)s->wideReached * sizeof(RtrWideNode));
    HIP_TRY(hipSetDevice(s->ctx->device));
    HIP_TRY(hipMemcpy(nodes, s->nodes4.p, nodeBytes, hipMemcpyDeviceToHost));      /* the records the tree reaches come first */
    return RTR_OK;
}

int rtr_scene_update_lights(rtr_scene* s, const RtrAreaLightInfo* lights, uint32_t n) {
    if (!s || (!lights && n)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_lights: null argument");
    if (n != s->numLights) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_lights: %u lights given, scene has %u (light geometry is part of the BVH)", n, s->numLights);
    for (uint32_t l = 0; l < n; ++l) {
        if (lights[l].vertexOffset != s->hostLights[l].vertexOffset || lights[l].indexOffset != s->hostLights[l].indexOffset ||
            lights[l].numTriangles != s->hostLights[l].numTriangles || memcmp(lights[l].transform, s->hostLights[l].transform, sizeof lights[l].transform))
            return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_scene_update_lights: light %u geometry/transform changed; only colour, intensity and sidedness may be updated without a rebuild", l);
    }
    HIP_TRY(hipSetDevice(s->ctx->device));
    if (n) {
        HIP_TRY(hipDeviceSynchronize());       /* frames in flight on other streams read s->lights: joined before the rewrite (contract in rtr.h) */
        HIP_TRY(hipMemcpyAsync(s->lights.p, lights, n * sizeof(RtrAreaLightInfo), hipMemcpyHostToDevice, s->ctx->stream));
        HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        s->hostLights.assign(lights, lights + n);
    }
    return RTR_OK;
}

/* The traversal kernels address triangle records, BVH2 nodes and 4-wide records through 32-bit byte offsets from one buffer base
 * each ((first + i) * 48, node << 5, record << 6 in signed 32-bit lane arithmetic: kernels/rtr_device.h trace(), kernels/rtr_kernels.hip
 * inner_nodes4 / tri_any): a scene whose arrays reach 2 GiB would silently intersect the wrong records, so it is refused here. */
int rtr_check_scene_limits(uint64_t numTriangles, uint64_t numNodes) {
    const uint64_t kMaxBytes = 1ull << 31;
    if (numTriangles >= (1ull << 28)) return fail(RTR_ERR_INVALID_ARGUMENT, "%llu triangles: too many for the leaf encoding (2^28)", (unsigned long long)numTriangles);
    if (numTriangles * sizeof(RtrBvhTri) >= kMaxBytes)
        return fail(RTR_ERR_INVALID_ARGUMENT, "%llu triangles: the %zu-byte triangle records would reach 2 GiB, past the kernels' 32-bit record offsets (at most %llu triangles)",
                    (unsigned long long)numTriangles, sizeof(RtrBvhTri), (unsigned long long)((kMaxBytes - 1) / sizeof(RtrBvhTri)));
    if (numNodes * RTR_WIDE_NODE_BYTES >= kMaxBytes)
        return fail(RTR_ERR_INVALID_ARGUMENT, "%llu BVH nodes: the %d-byte 4-wide records would reach 2 GiB, past the kernels' 32-bit record offsets (at most %llu nodes)",
                    (unsigned long long)numNodes, RTR_WIDE_NODE_BYTES, (unsigned long long)((kMaxBytes - 1) / RTR_WIDE_NODE_BYTES));
    return RTR_OK;
}

/* ---- frame -------------------------------------------------------------------------------- */
static int frame_new(rtr_ctx* ctx, uint32_t width, uint32_t rows, uint32_t images, bool ownImages, rtr_frame** out);

int rtr_frame_create(rtr_ctx* ctx, uint32_t width, uint32_t rows, uint32_t images, rtr_frame** out) {
    return frame_new(ctx, width, rows, images, true, out);
}

/* ownImages == false: a part of a split render (no images of its own, see rtr_frame::parts) */
static int frame_new(rtr_ctx* ctx, uint32_t width, uint32_t rows, uint32_t images, bool ownImages, rtr_frame** out) {
    if (!ctx || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_create: null ctx/out");
    if (ctx->destroyed) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_create: the context has been destroyed");
    *out = nullptr;
    if (width == 0 || rows == 0 || width > 65536 || rows > 65536) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_create: bad extent %ux%u", width, rows);
    if (images == 0) images = RTR_IMAGES_FRAMEBUFFER;
    const uint32_t known = 0xffu | RTR_IMG_BIT(RTR_IMAGE_HDR);
    if (images & ~known) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_create: unknown image bits 0x%x", images & ~known);
    HIP_TRY(hipSetDevice(ctx->device));
    rtr_frame* f = new rtr_frame();
    f->ctx = ctx; ++ctx->children; f->width = width; f->rows = rows; f->images = images;
    const size_t px = (size_t)width * rows;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 8 && e == hipSuccess && ownImages; ++i)
        if (images & RTR_IMG_BIT(i)) { e = f->img[i].alloc(px); if (e == hipSuccess) e = hipMemsetAsync(f->img[i].p, 0, px * 4, ctx->stream); }
    if (e == hipSuccess && ownImages && (images & RTR_IMG_BIT(RTR_IMAGE_HDR))) { e = f->hdr.alloc(px); if (e == hipSuccess) e = hipMemsetAsync(f->hdr.p, 0, px * 16, ctx->stream); }
    if (e == hipSuccess) e = f->counters.alloc(1);
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipEventCreate(&f->ev[i]);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreate(&f->evMega[i]);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { rtr_frame_destroy(f); return fail(e == hipErrorOutOfMemory ? RTR_ERR_OUT_OF_MEMORY : RTR_ERR_HIP, "rtr_frame_create: %s", hipGetErrorString(e)); }
    *out = f;
    return RTR_OK;
}

void rtr_frame_destroy(rtr_frame* f) {
    if (!f) return;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    for (auto& e : f->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : f->evMega) if (e) (void)hipEventDestroy(e);
    if (f->evDone) { (void)hipEventSynchronize(f->evDone); (void)hipEventDestroy(f->evDone); }
    if (f->evOwn) (void)hipEventDestroy(f->evOwn);
    for (rtr_frame* part : f->parts) rtr_frame_destroy(part);
    for (rtr_ctx* pc : f->partCtx) rtr_ctx_destroy(pc);
    for (auto& e : f->evPart) if (e) (void)hipEventDestroy(e);
    for (auto& e : f->evSplit) if (e) (void)hipEventDestroy(e);
    rtr_ctx* c = f->ctx;
    delete f;
    ctx_release_child(c);
}

int rtr_frame_bind_external(rtr_frame* f, int which, void* dptr, size_t bytes) {
    if (!f || which < 0 || which > 7) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_bind_external: bad argument");
    if (dptr && bytes != (size_t)f->width * f->rows * 4) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_bind_external: %zu bytes given, image is %zu", bytes, (size_t)f->width * f->rows * 4);
    if (dptr && ((uintptr_t)dptr & 3u)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_bind_external: pointer not 4-byte aligned");
    f->ext[which] = (uint32_t*)dptr;
    if (dptr) f->images |= RTR_IMG_BIT(which);
    return RTR_OK;
}

int rtr_frame_device_ptr(const rtr_frame* f, int which, void** dptr, size_t* bytes) {
    if (!f || !dptr) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_device_ptr: null argument");
    if (which == RTR_IMAGE_HDR) {
        if (!f->hdr.p) return fail(RTR_ERR_INVALID_ARGUMENT, "frame has no HDR image");
        *dptr = f->hdr.p; if (bytes) *bytes = (size_t)f->width * f->rows * 16; return RTR_OK;
    }
    if (which < 0 || which > 7 || !f->image_ptr(which)) return fail(RTR_ERR_INVALID_ARGUMENT, "frame has no image %d", which);
    *dptr = f->image_ptr(which); if (bytes) *bytes = (size_t)f->width * f->rows * 4;
    return RTR_OK;
}

int rtr_frame_download(const rtr_frame* f, int which, void* dst, size_t bytes) {
    if (!f || !dst) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_download: null argument");
    void* src = nullptr; size_t need = 0;
    int rc = rtr_frame_device_ptr(f, which, &src, &need);
    if (rc != RTR_OK) return rc;
    if (bytes != need) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_download: %zu bytes given, image %d is %zu", bytes, which, need);
    HIP_TRY(hipSetDevice(f->ctx->device));
    if (f->viaBatch) HIP_TRY(hipEventSynchronize(f->evDone));          /* its pixels came from another frame's stream */
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, f->ctx->stream));
    HIP_TRY(hipStreamSynchronize(f->ctx->stream));
    return RTR_OK;
}

/* Work about to be enqueued for `f` on its own stream comes behind a batch launch on another stream that wrote it last (the launch
 * runs on its leading frame's stream), and a later batch must come behind this work (ownPending): the ordering contract of
 * rtr_render_batch_async for everything that is not itself a render */
static int order_on_own_stream(rtr_frame* f) {
    hipStream_t st = f->ctx->stream;
    if (f->batchStream && f->batchStream != st) HIP_TRY(hipStreamWaitEvent(st, f->evDone, 0));
    f->batchStream = nullptr;
    f->ownPending = true;
    return RTR_OK;
}

int rtr_frame_clear(rtr_frame* f) {
    if (!f) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_clear: null frame");
    HIP_TRY(hipSetDevice(f->ctx->device));
    { const int rc = order_on_own_stream(f); if (rc != RTR_OK) return rc; }
    const size_t px = (size_t)f->width * f->rows;
    for (int i = 0; i < 8; ++i) if (f->image_ptr(i)) HIP_TRY(hipMemsetAsync(f->image_ptr(i), 0, px * 4, f->ctx->stream));
    if (f->hdr.p) HIP_TRY(hipMemsetAsync(f->hdr.p, 0, px * 16, f->ctx->stream));
    HIP_TRY(hipStreamSynchronize(f->ctx->stream));
    return RTR_OK;
}

/* ---- dispatch ------------------------------------------------------------------------------ */
/* Enqueues one launch of the pipeline over n frames (n = 1: rtr_render / rtr_render_async).  frames[0] leads: its context's stream
 * carries the work, its scratch holds the batch, its statistics describe the launch. */
static_assert(RTR_MAX_BATCH == rtrdev::kMaxBatch, "include/rtr.h and kernels/rtr_device.h disagree on the frames per launch");
static int enqueue_render(rtr_scene* s, const RtrCameraData* cams, const RtrSceneInfo* infos, const rtr_render_params* pin, rtr_frame* const* frames, uint32_t n, bool direct = false) {
    if (!s || !cams || !infos || !pin || !frames || n < 1 || !frames[0]) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: null argument");
    if (n > rtrdev::kMaxBatch) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_batch_async: %u frames, at most %u per launch", n, rtrdev::kMaxBatch);
    rtr_frame* f = frames[0];
    /* work is enqueued on the (leading) FRAME's context stream; the (read-only) scene may belong to another context of the same
     * device, so two frames on two streams can be in flight against one scene */
    if (s->ctx->device != f->ctx->device) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: scene and frame live on different devices");
    rtr_render_params p = *pin;
    if (p.bandRows == 0) p.bandRows = 8;
    if (p.shardCount == 0) p.shardCount = 1;
    if (p.images == 0) p.images = RTR_IMAGES_FRAMEBUFFER;
    if (p.width == 0 || p.height == 0 || p.spp == 0 || p.spp > 1024 || p.numShadowRays > 1024)
        return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: bad width/height/spp/numShadowRays (%u,%u,%u,%u)", p.width, p.height, p.spp, p.numShadowRays);
    if (p.bandRows % 8u) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: bandRows %u must be a multiple of 8 (one wave = one 8x8 tile)", p.bandRows);
    if (p.shardIndex >= p.shardCount) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: shardIndex %u >= shardCount %u", p.shardIndex, p.shardCount);
    const uint32_t rows = rtr_shard_rows(p.height, p.bandRows, p.shardCount);
    const uint32_t frameRows = direct ? p.height : rows;     /* direct: the images are the whole frame, this shard writes its own rows of them */
    if (p.images & RTR_IMAGES_DENOISE) return fail(RTR_ERR_UNSUPPORTED, "rtr_render: denoise/combine images are produced by rtr_denoise_combine, not by the ray-gen dispatch");
    if ((p.images & RTR_IMG_BIT(RTR_IMAGE_ANALYTIC)) && !s->hasLtc) return fail(RTR_ERR_UNSUPPORTED, "rtr_render: RTR_IMAGE_ANALYTIC needs the LTC tables (rtr_scene_desc.ltc1/ltc2)");
    const bool wantHdr = (p.images & RTR_IMG_BIT(RTR_IMAGE_HDR)) != 0 || p.accumulate;
    if (n > 1 && p.pipeline == 1) return fail(RTR_ERR_UNSUPPORTED, "rtr_render_batch_async: the megakernel renders one frame per launch");

    rtrdev::FrameBatch fb{};
    fb.n = n;
    uint32_t k = 0;
    uint64_t maxRays = 1;
    for (uint32_t b = 0; b < n; ++b) {
        rtr_frame* fr = frames[b];
        if (!fr) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_batch_async: frame %u is null", b);
        for (uint32_t c = 0; c < b; ++c) if (frames[c] == fr) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_batch_async: frame %u given twice", b);
        if (fr->ctx->device != f->ctx->device) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_batch_async: frame %u lives on another device", b);
        if (infos[b].numAreaLights > s->numLights) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: SceneInfo.numAreaLights %u > scene lights %u", infos[b].numAreaLights, s->numLights);
        if (infos[b].numAreaLights != infos[0].numAreaLights) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_batch_async: the frames of a launch must use the same number of area lights");
        if (fr->width != p.width || fr->rows != frameRows) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: frame is %ux%u, this shard needs %ux%u", fr->width, fr->rows, p.width, frameRows);
        if (wantHdr && !fr->hdr_ptr()) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: HDR accumulation requested but the frame has no RTR_IMAGE_HDR");
        FrameOut& fo = fb.fo[b];
        k = 0;
        for (int i = 0; i < 8; ++i) {
            fo.img[i] = nullptr;
            if (p.images & RTR_IMG_BIT(i)) {
                if (!fr->image_ptr(i)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: image %d requested but not in the frame", i);
                fo.img[i] = fr->image_ptr(i); ++k;
            }
        }
        fo.hdr = wantHdr ? fr->hdr_ptr() : nullptr;
        RenderArgs& ra = fb.ra[b];
        ra.cam = cams[b]; ra.info = infos[b];
        ra.width = p.width; ra.height = p.height; ra.spp = p.spp; ra.numShadowRays = p.numShadowRays;
        ra.bandRows = p.bandRows; ra.shardIndex = p.shardIndex; ra.shardCount = p.shardCount;
        ra.localRows = rows; ra.tilesPerRow = (p.width + 7u) / 8u;
        ra.images = p.images; ra.accumulate = p.accumulate; ra.accumulatedFrames = p.accumulatedFrames; ra.directRows = direct ? 1u : 0u;
        if (b == 0) for (uint32_t l = 0; l < infos[0].numAreaLights; ++l) maxRays += (uint64_t)s->hostLights[l].numTriangles * p.numShadowRays;
        ra.maxRaysPerSample = (uint32_t)maxRays;
    }
    const RenderArgs& ra = fb.ra[0];
    const FrameOut& fo = fb.fo[0];

    HIP_TRY(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    Counters* dstats = nullptr;
    if (p.collectStats) { HIP_TRY(hipMemsetAsync(f->counters.p, 0, sizeof(Counters), st)); dstats = f->counters.p; }

    const uint64_t paddedPixels = (uint64_t)((rows + 7u) / 8u) * ra.tilesPerRow * 64u;
    const uint64_t blocks = (paddedPixels + 255u) / 256u;
    const uint64_t nPS = blocks * 256u * p.spp * n;          /* pixel-sample slots of the launch */
    const uint64_t nRays = nPS * maxRays;                    /* queue capacity: every pixel-sample issuing every query */
    uint64_t slotStride = 256;                                /* the visibility planes are a power of two apart, so a slot names its pixel-sample with a mask */
    while (slotStride < nPS) slotStride <<= 1;
    const uint64_t nSlots = slotStride * maxRays;
    bool wave = p.pipeline != 1;
    if (wave && (nSlots >= (1ull << 31) || maxRays > 4096)) {
        if (p.pipeline == 2 || n > 1) return fail(RTR_ERR_UNSUPPORTED, "rtr_render: wavefront scratch would need %llu visibility slots", (unsigned long long)nSlots);
        wave = false;
    }
    if (blocks * n >= (1ull << 31)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render: frame too large");

    /* the counting form IS the timed kernel template; the 2-wide comparison kernel has none, and counting another kernel's work
     * under its name would be a wrong number */
    if (wave && p.collectStats && f->ctx->tun.trace_bvh4 == 0u)
        return fail(RTR_ERR_UNSUPPORTED, "rtr_render: collectStats with the tunable trace_bvh4 = 0: the 2-wide comparison kernel has no counting form (set it back to 1, or render with pipeline 1)");
    f->pendingWave = wave; f->pendingCounters = p.collectStats != 0;
    f->pendingImagesK = k; f->pendingHdr = wantHdr; f->pendingAccum = p.accumulate != 0;
    memset(&f->stats, 0, sizeof f->stats);
    f->stats.localRows = rows; f->stats.localPixels = rows * p.width * n;
    f->viaBatch = false; f->pendingSplit = 0;
    /* The launch runs on THIS frame's stream.  What another stream still holds for one of its frames comes first: a launch the frame
     * led on its own stream (ownPending), or a batch on a third stream that wrote it (batchStream).  In steady state — the same
     * frames batched behind the same leader, or frames joined between uses — neither is set and nothing is enqueued here (cross-stream
     * waits on every launch cost 3 % of the frame rate at N = 1 and 25 % on a 1/8 shard: profiles/r03/ab_batch_stream_order.log). */
    if (f->batchStream && f->batchStream != st) HIP_TRY(hipStreamWaitEvent(st, f->evDone, 0));
    f->batchStream = nullptr;
    for (uint32_t b = 1; b < n; ++b) {
        rtr_frame* fr = frames[b];
        if (fr->ownPending && fr->ctx->stream != st) {
            if (!fr->evOwn) HIP_TRY(hipEventCreateWithFlags(&fr->evOwn, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(fr->evOwn, fr->ctx->stream));
            HIP_TRY(hipStreamWaitEvent(st, fr->evOwn, 0));
        }
        if (fr->batchStream && fr->batchStream != st) HIP_TRY(hipStreamWaitEvent(st, fr->evDone, 0));
    }
    hipError_t e;
    if (wave) {
        /* the list of rays left to the redo kernels: the camera rays' needs one entry per pixel-sample at most; the any-hit kernel's
         * gets 1/16 of the queue (at least 2^20 entries) and k_shadow_tail redoes the whole queue should that ever overflow */
        uint64_t ovCap = std::max<uint64_t>(std::max<uint64_t>(nPS, nRays / 16), 1ull << 20);
        if (ovCap > nRays) ovCap = std::max<uint64_t>(nRays, nPS);
        if (f->hitTuvp.n < nPS) { HIP_TRY(f->hitTuvp.alloc(nPS)); HIP_TRY(f->hitCustom.alloc(nPS)); HIP_TRY(f->rayOrigin.alloc(nPS)); }
        if (f->vis.n < nSlots) HIP_TRY(f->vis.alloc(nSlots));
        if (f->rayDT.n < nRays) { HIP_TRY(f->rayDT.alloc(nRays)); HIP_TRY(f->raySlot.alloc(nRays)); }
        /* every array is (re)sized by its OWN need: the redo list is also k_primary's (one entry per pixel-sample at most), and a
         * launch with fewer queries per sample but more samples than an earlier one needs a longer list with a shorter queue */
        if (f->overflow.n < ovCap + 1) HIP_TRY(f->overflow.alloc(ovCap + 1));
        f->overflowCap = (uint32_t)std::min<uint64_t>(ovCap, f->overflow.n - 1);      /* what the any-hit kernel may use of it */
#ifdef RTR_TEST_HOOKS
        if (const char* e = getenv("RTR_TRACE_OVERFLOW_CAP")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1 && v < f->overflowCap) f->overflowCap = (uint32_t)v; }   /* a list short enough to overflow */
#endif
        {   /* batch lists of the binned queue (octant x consumer XCD): a run's batches are dealt round-robin to the eight lists of
             * its octant, so a list holds at most 1/8 of one batch per 64 rays (the smallest batch) + one per k_shadow_gen_oct workgroup */
            const uint64_t ls = nRays / 64 / rtrdev::kQueueRegions + nPS / 256 + 16;
            if ((uint64_t)f->listStride < ls) { HIP_TRY(f->batchLists.alloc((size_t)ls * rtrdev::kQueueLists)); f->listStride = (uint32_t)ls; }
        }
        f->slotStride = (uint32_t)slotStride;
        if (!f->queueCount.p) HIP_TRY(f->queueCount.alloc(rtrdev::kQueueCtrlWords));
        if (!f->clk.p) { HIP_TRY(f->clk.alloc(2 * rtrdev::kQueueRegions)); HIP_TRY(hipMemsetAsync(f->clk.p, 0, 2 * rtrdev::kQueueRegions * sizeof(unsigned long long), st)); }
        if (!f->spill.p) HIP_TRY(f->spill.alloc(rtrdev::kSpillInts));      /* 64 entries x the redo kernels' grid */
        Workspace ws;
        ws.hitTuvp = f->hitTuvp.p; ws.hitCustom = f->hitCustom.p; ws.vis = f->vis.p; ws.visFill = f->visFill;
        ws.visPlaneBytes = (size_t)nPS; ws.visPlanes = (uint32_t)maxRays;      /* what a launch fills: the first nPS bytes of each of the maxRays planes, not the power-of-two pitch between them */
#ifdef RTR_TEST_HOOKS
        if (const char* e = getenv("RTR_TRACE_VIS_FILL")) if ((e[0] == '0' || e[0] == '1') && !e[1]) ws.visFill = (uint32_t)(e[0] - '0');      /* force the pre-fill */
#endif
        ws.rayQueue.dt = f->rayDT.p; ws.rayQueue.slot = f->raySlot.p; ws.rayQueue.origin = f->rayOrigin.p; ws.rayQueue.slotStride = f->slotStride; ws.rayQueue.slotMask = f->slotStride - 1u;
        ws.queueCount = f->queueCount.p; ws.capPixelSamples = nPS; ws.capRays = nRays; ws.spill = f->spill.p; ws.overflow = f->overflow.p; ws.overflowCap = f->overflowCap; ws.batchLists = f->batchLists.p; ws.listStride = f->listStride; ws.clk = f->clk.p;
        e = rtrdev::launch_wavefront(s->dev, fb, ws, f->ctx->tun, (int)s->stats.stackEntries, dstats, st, f->ev, (uint32_t)f->ctx->prop.multiProcessorCount);
    } else {
        (void)hipEventRecord(f->evMega[0], st);
        e = rtrdev::launch_megakernel(s->dev, ra, fo, (int)s->stats.stackEntries, dstats, st);
        (void)hipEventRecord(f->evMega[1], st);
    }
    if (e != hipSuccess) return fail(RTR_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    f->pendingStats = true;
    f->ownPending = true;
    /* ... and each of the other frames gets an event behind the launch: rtr_frame_wait / rtr_frame_download wait on it, and so does
     * the stream of whatever launch writes the frame next (above) */
    for (uint32_t b = 1; b < n; ++b) {
        rtr_frame* fr = frames[b];
        if (!fr->evDone) HIP_TRY(hipEventCreateWithFlags(&fr->evDone, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(fr->evDone, st));
        fr->batchStream = st;
        fr->viaBatch = true; fr->pendingStats = false;
        memset(&fr->stats, 0, sizeof fr->stats);
    }
    return RTR_OK;
}

int rtr_render_batch_limit(const rtr_scene* s, const rtr_render_params* pin, uint32_t numAreaLights, uint32_t* maxFrames) {
    if (!s || !pin || !maxFrames) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_batch_limit: null argument");
    if (numAreaLights > s->numLights) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_batch_limit: %u area lights, the scene has %u", numAreaLights, s->numLights);
    rtr_render_params p = *pin;
    if (p.bandRows == 0) p.bandRows = 8;
    if (p.shardCount == 0) p.shardCount = 1;
    if (p.width == 0 || p.height == 0 || p.spp == 0 || p.bandRows % 8u) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_batch_limit: bad width/height/spp/bandRows");
    *maxFrames = 1;
    if (p.pipeline == 1) return RTR_OK;
    /* the arithmetic of enqueue_render() */
    uint64_t maxRays = 1;
    for (uint32_t l = 0; l < numAreaLights; ++l) maxRays += (uint64_t)s->hostLights[l].numTriangles * p.numShadowRays;
    const uint32_t rows = rtr_shard_rows(p.height, p.bandRows, p.shardCount);
    const uint64_t blocks = ((uint64_t)((rows + 7u) / 8u) * ((p.width + 7u) / 8u) * 64u + 255u) / 256u;
    for (uint32_t n = rtrdev::kMaxBatch; n > 1; --n) {
        const uint64_t nPS = blocks * 256u * p.spp * n;
        uint64_t slotStride = 256;
        while (slotStride < nPS) slotStride <<= 1;
        if (slotStride * maxRays < (1ull << 31) && maxRays <= 4096 && blocks * n < (1ull << 31)) { *maxFrames = n; break; }
    }
    return RTR_OK;
}

/* One frame as `parts` band-shards, each on a stream of its own, all writing their rows of the SAME images: the kernels of one frame
 * are a dependency chain and each ends in a tail, so part k+1's camera rays and queue build run under part k's traversal and its
 * traversal fills the tail of part k's — frames in flight, inside one frame. */
int rtr_render_split_async(rtr_scene* s, const RtrCameraData* cam, const RtrSceneInfo* info, const rtr_render_params* pin, rtr_frame* f, uint32_t parts) {
    if (!s || !cam || !info || !pin || !f) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_split_async: null argument");
    if (parts < 1 || parts > RTR_MAX_SPLIT) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_split_async: %u parts, 1 to %d", parts, RTR_MAX_SPLIT);
    if (pin->shardCount > 1) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_split_async: the frame must be whole (shardCount %u); a shard of a multi-GPU frame is not split again", pin->shardCount);
    if (f->extHdr) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_split_async: not a frame of the caller's");
    rtr_render_params p = *pin;
    if (p.bandRows == 0) p.bandRows = 8;
    if (p.bandRows % 8u) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_split_async: bandRows %u must be a multiple of 8", p.bandRows);
    if (p.height == 0 || p.width == 0) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_split_async: empty frame");
    const uint32_t bands = (p.height + p.bandRows - 1) / p.bandRows;
    if (parts > bands) parts = bands;                 /* never a part without a band */
    if (parts <= 1) { p.shardIndex = 0; p.shardCount = 1; return enqueue_render(s, cam, info, &p, &f, 1); }
    if (f->width != p.width || f->rows != p.height) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_render_split_async: frame is %ux%u, the whole frame is %ux%u", f->width, f->rows, p.width, p.height);
    HIP_TRY(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    /* the parts: internal frame objects (scratch, events, counters; no images) on contexts (streams) of their own */
    while (f->parts.size() < parts) {
        rtr_ctx* pc = nullptr; rtr_frame* pf = nullptr;
        int rc = ctx_create_prio(f->ctx->device, f->ctx->tun.split_priorities ? (int)f->parts.size() : -1, &pc);
        if (rc != RTR_OK) return rc;
        rc = frame_new(pc, f->width, f->rows, f->images, false, &pf);
        if (rc != RTR_OK) { rtr_ctx_destroy(pc); return rc; }
        hipEvent_t ev = nullptr;
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) { rtr_frame_destroy(pf); rtr_ctx_destroy(pc); return fail(RTR_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
        f->partCtx.push_back(pc); f->parts.push_back(pf); f->evPart.push_back(ev);
    }
    for (auto& e : f->evSplit) if (!e) HIP_TRY(hipEventCreate(&e));
    /* what another stream still holds for this frame comes first (enqueue_render's rule) */
    if (f->batchStream && f->batchStream != st) HIP_TRY(hipStreamWaitEvent(st, f->evDone, 0));
    f->batchStream = nullptr;
    HIP_TRY(hipEventRecord(f->evSplit[0], st));          /* the fork */
    /* A failure inside the loop must not leave the frame's stream un-joined from parts that were already enqueued (a download, clear
     * or denoise that follows would race them, and rtr_frame_wait would never collect them): the error is kept, every part that WAS
     * enqueued is joined, the frame is marked pending for exactly those, and then the error is returned. */
    int rc = RTR_OK;
    uint32_t started = 0;
    auto hip_keep = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == RTR_OK) rc = fail(RTR_ERR_HIP, "rtr_render_split_async: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    for (uint32_t k = 0; k < parts && rc == RTR_OK; ++k) {
        rtr_frame* pf = f->parts[k];
        pf->ctx->tun = f->ctx->tun;
        pf->images = f->images;
        for (int i = 0; i < 8; ++i) pf->ext[i] = f->image_ptr(i);
        pf->extHdr = f->hdr.p;
        hipStream_t ps = pf->ctx->stream;
        if (!hip_keep(hipStreamWaitEvent(ps, f->evSplit[0], 0), "hipStreamWaitEvent (fork)")) break;
        p.shardIndex = k; p.shardCount = parts;
        const int prc = enqueue_render(s, cam, info, &p, &pf, 1, true);
        if (prc != RTR_OK) { rc = prc; break; }
        ++started;                                              /* from here on part k has work on its stream: it must be joined */
        if (hip_keep(hipEventRecord(f->evPart[k], ps), "hipEventRecord (part)")) hip_keep(hipStreamWaitEvent(st, f->evPart[k], 0), "hipStreamWaitEvent (join)");
        else hip_keep(hipStreamSynchronize(ps), "hipStreamSynchronize (join of last resort)");      /* no event to order behind: drain the part here */
    }
    hip_keep(hipEventRecord(f->evSplit[1], st), "hipEventRecord (join)");
    f->ownPending = true; f->viaBatch = false;
    f->pendingSplit = started; f->pendingStats = started != 0;
    return rc;
}

int rtr_render_split(rtr_scene* s, const RtrCameraData* cam, const RtrSceneInfo* info, const rtr_render_params* p, rtr_frame* f, uint32_t parts) {
    int rc = rtr_render_split_async(s, cam, info, p, f, parts);
    if (rc != RTR_OK) return rc;
    return rtr_frame_wait(f);
}

/* joins a split render and makes the frame's statistics out of its parts': counters and per-kernel times are SUMS over the parts (the
 * parts' kernels overlap: the sum of their durations exceeds the frame's), totalMs is the frame's own duration, fork to join */
static int wait_split(rtr_frame* f) {
    const uint32_t parts = f->pendingSplit;
    f->pendingSplit = 0; f->pendingStats = false;
    rtr_frame_stats t; memset(&t, 0, sizeof t);
    double clk = 0; uint32_t nclk = 0;
    for (uint32_t k = 0; k < parts; ++k) {
        int rc = rtr_frame_wait(f->parts[k]);
        if (rc != RTR_OK) return rc;
        const rtr_frame_stats& a = f->parts[k]->stats;
        t.numRays += a.numRays; t.numPrimaryRays += a.numPrimaryRays; t.numShadowRays += a.numShadowRays; t.numNodeVisits += a.numNodeVisits;
        t.numTriTests += a.numTriTests; t.numHits += a.numHits; t.numLightFetches += a.numLightFetches; t.numLightTriFetches += a.numLightTriFetches;
        t.numTexFetches += a.numTexFetches; t.numAlphaTests += a.numAlphaTests; t.algorithmicBytes += a.algorithmicBytes;
        t.numShadowNodeVisits += a.numShadowNodeVisits; t.numShadowTriTests += a.numShadowTriTests; t.shadowTraceBytes += a.shadowTraceBytes;
        t.primaryMs += a.primaryMs; t.shadowGenMs += a.shadowGenMs; t.shadowTraceMs += a.shadowTraceMs; t.resolveMs += a.resolveMs; t.shadowTailMs += a.shadowTailMs;
        t.localRows += a.localRows; t.localPixels += a.localPixels; t.pipelineUsed = a.pipelineUsed;
        t.shadowInnerIterations += a.shadowInnerIterations; t.shadowInnerActiveLanes += a.shadowInnerActiveLanes;
        t.shadowTriIterations += a.shadowTriIterations; t.shadowTriActiveLanes += a.shadowTriActiveLanes; t.shadowRefills += a.shadowRefills;
        t.primaryTailRays += a.primaryTailRays; t.shadowTailRays += a.shadowTailRays;
        if (a.shadowTraceClockMHz > 0.f) {
            clk += a.shadowTraceClockMHz; ++nclk;
            t.shadowTraceClockMinMHz = (t.shadowTraceClockMinMHz == 0.f || a.shadowTraceClockMinMHz < t.shadowTraceClockMinMHz) ? a.shadowTraceClockMinMHz : t.shadowTraceClockMinMHz;
            t.shadowTraceClockMaxMHz = a.shadowTraceClockMaxMHz > t.shadowTraceClockMaxMHz ? a.shadowTraceClockMaxMHz : t.shadowTraceClockMaxMHz;
        }
    }
    if (nclk) t.shadowTraceClockMHz = (float)(clk / nclk);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, f->evSplit[0], f->evSplit[1]);
    t.totalMs = ms;
    f->stats = t;
    return RTR_OK;
}

int rtr_frame_wait(rtr_frame* f) {
    if (!f) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_wait: null frame");
    HIP_TRY(hipSetDevice(f->ctx->device));
    HIP_TRY(hipStreamSynchronize(f->ctx->stream));
    f->ownPending = false;
    if (f->pendingSplit) return wait_split(f);
    if (f->viaBatch) { HIP_TRY(hipEventSynchronize(f->evDone)); f->batchStream = nullptr; return RTR_OK; }      /* rendered in another frame's launch: that launch's times and counters are the leading frame's */
    if (!f->pendingStats) return RTR_OK;
    f->pendingStats = false;
    rtr_frame_stats& s = f->stats;
    if (f->pendingWave) {
        float a = 0, b = 0, c = 0, d = 0;
        (void)hipEventElapsedTime(&a, f->ev[0], f->ev[1]);
        (void)hipEventElapsedTime(&b, f->ev[1], f->ev[2]);
        (void)hipEventElapsedTime(&c, f->ev[2], f->ev[5]);
        float tail = 0;
        (void)hipEventElapsedTime(&tail, f->ev[5], f->ev[3]);
        s.shadowTailMs = tail;
        (void)hipEventElapsedTime(&d, f->ev[3], f->ev[4]);
        s.primaryMs = a; s.shadowGenMs = b; s.shadowTraceMs = c; s.resolveMs = d; s.totalMs = a + b + c + tail + d;
        s.pipelineUsed = 2;
        if (f->clk.p) {      /* shader clock held during the any-hit launch: s_memtime ticks over 100-MHz ticks of one wave per XCD; the MEAN over the XCDs — they clock
                              * independently (2.25 ... 2.40 GHz within one launch on a warm part) and the launch's vector-issue capacity is the sum of theirs */
            unsigned long long h[2 * rtrdev::kQueueRegions];
            HIP_TRY(hipMemcpy(h, f->clk.p, sizeof h, hipMemcpyDeviceToHost));
            float mhz[rtrdev::kQueueRegions]; int nv = 0;
            for (uint32_t r = 0; r < rtrdev::kQueueRegions; ++r) if (h[2 * r + 1]) mhz[nv++] = (float)((double)h[2 * r] / (double)h[2 * r + 1] * 100.0);
            for (int i = 1; i < nv; ++i) for (int j = i; j > 0 && mhz[j] < mhz[j - 1]; --j) { const float t = mhz[j]; mhz[j] = mhz[j - 1]; mhz[j - 1] = t; }
            double sum = 0; for (int i = 0; i < nv; ++i) sum += mhz[i];
            s.shadowTraceClockMHz = nv ? (float)(sum / nv) : 0.f;
            s.shadowTraceClockMinMHz = nv ? mhz[0] : 0.f; s.shadowTraceClockMaxMHz = nv ? mhz[nv - 1] : 0.f;
        }
        if (f->queueCount.p) {      /* the next launch pre-fills the visibility array with this one's commoner outcome */
            uint32_t q[4] = {0, 0, 0, 0};
            HIP_TRY(hipMemcpy(q, f->queueCount.p, sizeof q, hipMemcpyDeviceToHost));
            if (q[0]) f->visFill = (2ull * q[3] >= q[0]) ? 1u : 0u;
        }
    } else {
        float a = 0;
        (void)hipEventElapsedTime(&a, f->evMega[0], f->evMega[1]);
        s.primaryMs = a; s.totalMs = a;
        s.pipelineUsed = 1;
    }
    if (f->pendingCounters) {
        Counters h;
        HIP_TRY(hipMemcpy(&h, f->counters.p, sizeof h, hipMemcpyDeviceToHost));
        s.numRays = h.rays; s.numPrimaryRays = h.primary; s.numShadowRays = h.shadow;
        s.numNodeVisits = h.nodes; s.numTriTests = h.tris; s.numHits = h.hits;
        s.numLightFetches = h.lightFetch; s.numLightTriFetches = h.lightTriFetch;
        s.numShadowNodeVisits = h.shadowNodes; s.numShadowTriTests = h.shadowTris;
        s.numTexFetches = h.texFetch; s.numAlphaTests = h.alphaTests;
        const uint64_t shadowNodeBytes = f->pendingWave ? RTR_WIDE_NODE_BYTES : RTR_BVH_NODE_BYTES;     /* the megakernel walks the BVH2 for its shadow rays too */
        s.shadowTraceBytes = shadowNodeBytes * h.shadowNodes + 48ull * h.shadowTris + 37ull * h.shadow;      /* per ray: 20 B of queue record + the 16-B origin of its pixel-sample + its visibility byte */
        s.shadowInnerIterations = h.innerIters; s.shadowInnerActiveLanes = h.innerLanes;
        s.shadowTriIterations = h.triIters; s.shadowTriActiveLanes = h.triLanes; s.shadowRefills = h.refills;
        if (f->overflow.p) { uint32_t ov = 0; HIP_TRY(hipMemcpy(&ov, f->overflow.p, sizeof ov, hipMemcpyDeviceToHost)); s.shadowTailRays = ov; }
        if (f->pendingWave && f->queueCount.p) { uint32_t rd = 0; HIP_TRY(hipMemcpy(&rd, f->queueCount.p + 2, sizeof rd, hipMemcpyDeviceToHost)); s.primaryTailRays = rd; }
        s.algorithmicBytes = (uint64_t)RTR_BVH_NODE_BYTES * (h.nodes - h.shadowNodes) + shadowNodeBytes * h.shadowNodes + 48ull * h.tris + 236ull * (h.hits + h.alphaTests) + 96ull * h.lightFetch + 156ull * h.lightTriFetch +
                             16ull * h.texFetch +
                             4ull * f->pendingImagesK * s.localPixels + (f->pendingHdr ? (f->pendingAccum ? 32ull : 16ull) * s.localPixels : 0ull);
    }
    return RTR_OK;
}

int rtr_render_async(rtr_scene* s, const RtrCameraData* cam, const RtrSceneInfo* info, const rtr_render_params* p, rtr_frame* f) {
    return enqueue_render(s, cam, info, p, &f, 1);
}

int rtr_render_batch_async(rtr_scene* s, const RtrCameraData* cams, const RtrSceneInfo* infos, const rtr_render_params* p, rtr_frame* const* frames, uint32_t n) {
    return enqueue_render(s, cams, infos, p, frames, n);
}

int rtr_render(rtr_scene* s, const RtrCameraData* cam, const RtrSceneInfo* info, const rtr_render_params* p, rtr_frame* f) {
    int rc = enqueue_render(s, cam, info, p, &f, 1);
    if (rc != RTR_OK) return rc;
    return rtr_frame_wait(f);
}

int rtr_frame_get_stats(const rtr_frame* f, rtr_frame_stats* out) {
    if (!f || !out) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_frame_get_stats: null argument");
    *out = f->stats;
    return RTR_OK;
}

int rtr_denoise_combine(rtr_frame* f, int iterations) {
    if (!f) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_denoise_combine: null frame");
    if (iterations < 0 || iterations > 64) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_denoise_combine: iterations %d", iterations);
    for (int i = 0; i < 8; ++i)
        if (!f->image_ptr(i)) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_denoise_combine: the frame lacks image %d (create it with all of images 0-7)", i);
    HIP_TRY(hipSetDevice(f->ctx->device));
    { const int rc = order_on_own_stream(f); if (rc != RTR_OK) return rc; }
    hipStream_t st = f->ctx->stream;
    uint32_t* sh = f->image_ptr(RTR_IMAGE_SHADOWED); uint32_t* un = f->image_ptr(RTR_IMAGE_UNSHADOWED);
    uint32_t* dsh = f->image_ptr(RTR_IMAGE_DENOISED_SHADOWED); uint32_t* dun = f->image_ptr(RTR_IMAGE_DENOISED_UNSHADOWED);
    const uint32_t* nrm = f->image_ptr(RTR_IMAGE_NORMAL); const uint32_t* pos = f->image_ptr(RTR_IMAGE_POSITION);
    int denoisingOutput = 1;                                            /* application.cppm:392 */
    for (int i = 0; i < iterations; ++i) {
        const int step = (i + 1) * 1;                                   /* (i + 1) * DENOISING_STRENGTH */
        hipError_t e;
        if (denoisingOutput == 1) {
            e = rtrdev::launch_denoise_pair(un, dun, sh, dsh, nrm, pos, f->width, f->rows, step, 1.0f, 0.001f, 0.001f, st);
        } else {
            e = rtrdev::launch_denoise_pair(dun, un, dsh, sh, nrm, pos, f->width, f->rows, step, 1.0f, 0.001f, 0.001f, st);
        }
        if (e != hipSuccess) return fail(RTR_ERR_HIP, "denoise launch: %s", hipGetErrorString(e));
        denoisingOutput = 1 - denoisingOutput;
    }
    hipError_t e = rtrdev::launch_combine(f->image_ptr(RTR_IMAGE_ANALYTIC), denoisingOutput == 0 ? sh : dsh, denoisingOutput == 0 ? un : dun,
                                          f->image_ptr(RTR_IMAGE_FINAL), f->width, f->rows, st);
    if (e != hipSuccess) return fail(RTR_ERR_HIP, "combine launch: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(st));
    return RTR_OK;
}

int rtr_deinterleave_bands(rtr_ctx* ctx, const void* gathered, void* dst, uint32_t width, uint32_t height, uint32_t bandRows, uint32_t shardCount) {
    if (!ctx || !gathered || !dst || width == 0 || height == 0) return fail(RTR_ERR_INVALID_ARGUMENT, "rtr_deinterleave_bands: bad argument");
    if (bandRows == 0) bandRows = 8;
    if (shardCount == 0) shardCount = 1;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t localRows = rtr_shard_rows(height, bandRows, shardCount);
    hipError_t e = rtrdev::launch_deinterleave((const uint32_t*)gathered, (uint32_t*)dst, width, height, bandRows, shardCount, localRows, ctx->stream);
    if (e != hipSuccess) return fail(RTR_ERR_HIP, "deinterleave launch: %s", hipGetErrorString(e));
    return RTR_OK;   /* enqueued on the ctx stream; the caller synchronises (stream order is enough for a following copy) */
}

}  // extern "C"
