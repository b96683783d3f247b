/* rtr_types.h — plain-old-data layouts shared by the host scene layer, the C ABI,
 * the HIP kernels and the CPU oracle.  All little-endian fp32 / u32.
 *
 * Every struct here is byte-for-byte the layout the reference hands to its GPU
 * (SURVEY.md Appendix A); the reference file:line each one mirrors is cited.
 * No glm, no Vulkan types: a 3x4 row-major float matrix replaces
 * vk::TransformMatrixKHR, float[3]+pad replaces glm::vec3+pad.
 */
#ifndef RTR_TYPES_H
#define RTR_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: src/scene/geometry/vertex.cppm:11-15, src/shaders/raycommon.glsl:4-8 (48 B) */
typedef struct RtrVertex {
    float position[3]; float pad0;
    float normal[3];   float pad1;
    float uv[2];       float pad2[2];
} RtrVertex;

/* reference: src/scene/camera.cppm:21-26, src/shaders/raycommon.glsl:10-15 (64 B) */
typedef struct RtrCameraData {
    float position[3];                float _pad0;
    float topLeftViewportCorner[3];   float _pad1;
    float horizontalViewportDelta[3]; float _pad2;
    float verticalViewportDelta[3];   float _pad3;
} RtrCameraData;

/* reference: src/scene/scene_info.cppm:10-20, src/shaders/raygen.rgen:35-43 (32 B push constant) */
typedef struct RtrSceneInfo {
    uint32_t frame;
    uint32_t numAreaLights;
    uint32_t _pad0;
    uint32_t _pad1;
    float    camPosition[3];
    float    pad2_;
} RtrSceneInfo;

/* reference: src/scene/object.cppm:21-44, src/shaders/raycommon.glsl:29-51 (80 B) */
typedef struct RtrObjectInfo {
    uint32_t vertexOffset;
    uint32_t indexOffset;
    float    pad0_[2];
    uint32_t usesColorMap;
    uint32_t usesSpecularMap;
    uint32_t usesMetallicMap;
    uint32_t usesOpacityMap;
    uint32_t colorIndex;
    uint32_t specularIndex;
    uint32_t metallicIndex;
    uint32_t opacityIndex;
    float    color[3];
    float    pad1_;
    float    specular;
    float    metallic;
    float    pad3_[2];
} RtrObjectInfo;

/* reference: src/scene/area_light.cppm:23-33, src/shaders/raycommon.glsl:53-63 (96 B).
 * transform is a column-major mat4 (element [c*4+r]), packed as in src/core/utils.cppm:11-39. */
typedef struct RtrAreaLightInfo {
    float    color[3];
    float    intensity;
    uint32_t vertexOffset;
    uint32_t indexOffset;
    uint32_t numTriangles;
    uint32_t isTwoSided;
    float    transform[16];
} RtrAreaLightInfo;

/* reference: src/app/structs.cppm:14-21, src/shaders/denoise.comp:9-16 (24 B) */
typedef struct RtrDenoisingInfo {
    int32_t step_width;
    float   c_phi;
    float   n_phi;
    float   p_phi;
    int32_t denoisedImagesAreOutput;
    int32_t isShadowedImage;
} RtrDenoisingInfo;

/* One mesh = one BLAS in the reference (src/vulkan/raytracing/blas.cppm:75-167;
 * BLASCreateInfo filled at src/app/setup/geometry_builder.cppm:98-112, src/core/file.cppm:254-267).
 * Offsets are element indices into the concatenated vertex / index arrays; indices are mesh-local. */
typedef struct RtrMesh {
    uint32_t vertexOffset;   /* vertexIndexOffset */
    uint32_t indexOffset;    /* indexIndexOffset  */
    uint32_t vertexCount;
    uint32_t indexCount;     /* 3 * triangles */
    uint32_t isOpaque;       /* BLAS geometry OPAQUE flag (blas.cppm:98-100) */
    uint32_t _pad[3];
} RtrMesh;

/* One TLAS instance (src/vulkan/raytracing/tlas.cppm:52-82): lights first, then objects;
 * customIndex = position in that list; transform = 3x4 row-major object->world. */
typedef struct RtrInstance {
    uint32_t meshIndex;
    uint32_t customIndex;
    uint32_t _pad[2];
    float    transform[12];  /* row-major 3x4: m[r*4+c] */
} RtrInstance;

/* ---- device BVH layout, version 3 ("BVH2 / children-in-parent / 32 B, 16-bit planes on a scene grid") ----------
 * One node holds the boxes of BOTH children, each plane quantised OUTWARD to a 16-bit coordinate of the scene-wide
 * grid `RtrBvhGrid` (plane = origin + q * scale per axis), so a visit is one 32-B fetch (2 x dwordx4): the traversal
 * kernels are bound by the bytes they pull through the vector-memory path, and version 2 (fp32 planes, 64 B) cost
 * 21 % more time in the dominant kernel.  Boxes only have to be conservative — the hit a ray reports is decided by the
 * triangle tests alone — so rendered results do not depend on the quantisation.
 *   q[side*4 + isMax*2 + axis]   axis = 0 (x), 1 (y);  side 0 = left child, 1 = right child
 *   q[8 + side*2 + isMax]        axis = 2 (z)
 *   i.e. as 32-bit words: (lminx|lminy<<16) (lmaxx|lmaxy<<16) (rminx|rminy<<16) (rmaxx|rmaxy<<16) (lminz|lmaxz<<16)
 *   (rminz|rmaxz<<16) — the pairing lets one packed FMA decode-and-slab two planes.
 *   child[0], child[1]: >= 0 -> index of an inner node;
 *                       <  0 -> leaf: code = ~child; first = code >> 3; count = (code & 7) + 1
 *   The root is always an inner node: a scene with a single leaf stores that leaf as BOTH children (testing a
 *   triangle twice cannot change the (t, id)-minimal hit); an empty scene holds one degenerate triangle.
 */
#define RTR_BVH_LAYOUT_VERSION 3
#define RTR_BVH_MAX_LEAF 8
#define RTR_BVH_NODE_BYTES 32   /* bytes a node visit fetches (the N_node coefficient of the algorithmic-byte formulas) */
typedef struct RtrBvhNode {
    uint16_t q[12];
    int32_t  child[2];
} RtrBvhNode;
#define RTR_BVH_QSLOT(side, isMax, axis) ((axis) < 2 ? (side) * 4 + (isMax) * 2 + (axis) : 8 + (side) * 2 + (isMax))

/* The grid the planes of every node live on (one per scene; rewritten by a refit). */
typedef struct RtrBvhGrid {
    float origin[3]; uint32_t wideCentreXY;    /* x | y << 16: grid coordinate the planes of the 4-wide records are offsets from (below) */
    float scale[3];  uint32_t wideCentreZ;     /* z */
} RtrBvhGrid;

/* ---- 4-wide view of the same tree (layout W4.0): what the any-hit kernel walks (k_shadow_trace4) ---------------------------
 * One 64-B record = a collapsed subtree of the BVH2 (greedy: open the inner child with the largest box while a slot is free):
 * up to four child boxes and their four child codes, so a ray makes about half as many DEPENDENT visits.  A plane is stored as
 * a HALF FLOAT: its offset in grid steps from the scene's WIDE CENTRE c (q - c; RtrBvhGrid::wideCentreXY / Z: per axis the mean midpoint of
 * the tree's leaf boxes, so the half floats are finest where the geometry is), rounded outward to 11 significant bits
 * (exact within 2048 steps of c, 2^-11 of the distance from it beyond; a magnitude above 65504 becomes infinite), so that its parameter on a ray is ONE
 * instruction, t = fma(f16 plane, ga, gbc) with v_fma_mix_f32, no conversion (ga, gbc: rtr_ray_grid_centre).  Records are in breadth-first order (the first ones are the top levels, which the kernel keeps in LDS); record 0 is
 * the root.  Built on the device after every build / refit (kernels/rtr_bvh.hip: k_wide_nodes, k_permute_wide).
 *   plane[k] : slot k: (xmin | ymin << 16) (xmax | ymax << 16) (zmin | zmax << 16), IEEE binary16 each
 *   child[k] : >= 0 index of a record of this array; < 0 leaf code as in RtrBvhNode (triangles in the same leaf-ordered array);
 *              0x80000000 = empty slot (slots 0 and 1 are never empty); its planes are min = +inf, max = -inf: an inside-out box
 *              no ray enters when the entry / exit planes are picked by the ray's direction signs (the kernel's octant forms)
 */
#define RTR_WIDE_LAYOUT_VERSION 45
#define RTR_WIDE_NODE_BYTES 64
/* entries of the per-ray stack the any-hit kernel keeps in LDS while it walks the wide view; a ray that holds more after a visit is
 * redone over the BVH2 (k_shadow_tail).  Part of the definition of a ray's walk (the counting form and the oracle follow it). */
#ifndef RTR_WIDE_STACK
#define RTR_WIDE_STACK 16
#endif
#define RTR_WIDE_EMPTY ((int32_t)0x80000000)
typedef struct RtrWideNode {
    uint32_t plane[4][3];
    int32_t  child[4];
} RtrWideNode;

/* 48-B world-space Moeller-Trumbore record, in BVH leaf order.
 * v0 / e1 = v1-v0 / e2 = v2-v0 in world space; the three w slots carry the ids the
 * reference's hit shaders read (gl_InstanceCustomIndexEXT, gl_PrimitiveID) and flags. */
typedef struct RtrBvhTri {
    float    v0[3]; uint32_t customIndex;
    float    e1[3]; uint32_t primitiveId;
    float    e2[3]; uint32_t flags;      /* bit0: alpha-tested (any-hit needed) */
} RtrBvhTri;

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
static_assert(sizeof(RtrVertex) == 48, "Vertex must be 48 B");
static_assert(sizeof(RtrCameraData) == 64, "GPUCameraData must be 64 B");
static_assert(sizeof(RtrSceneInfo) == 32, "SceneInfo must be 32 B");
static_assert(sizeof(RtrObjectInfo) == 80, "GPUObjectInfo must be 80 B");
static_assert(sizeof(RtrAreaLightInfo) == 96, "GPUAreaLightInfo must be 96 B");
static_assert(sizeof(RtrDenoisingInfo) == 24, "DenoisingInfo must be 24 B");
static_assert(sizeof(RtrBvhNode) == 32, "BVH node must be 32 B");
static_assert(sizeof(RtrBvhGrid) == 32, "BVH grid record must be 32 B");
static_assert(sizeof(RtrBvhTri) == 48, "BVH triangle must be 48 B");
static_assert(sizeof(RtrWideNode) == 64, "wide node must be 64 B");
#endif

#endif /* RTR_TYPES_H */
