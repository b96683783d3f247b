/* bvh_build.h — host-side BVH builder of the product (replaces the driver-opaque
 * vkCmdBuildAccelerationStructuresKHR of reference src/vulkan/raytracing/blas.cppm:75-167 and
 * tlas.cppm:44-149; PREFER_FAST_TRACE -> full-sweep-quality binned SAH).
 *
 * One flat world-space BVH over every instance's triangles (no TLAS/BLAS split): the scenes of
 * this path replicate no geometry worth instancing and a single level removes one indirection
 * and the per-instance ray transform from the traversal loop.
 */
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/rtr_types.h"

namespace rtr {

struct WorldTriangle {
    float v[3][3];          /* world-space corners */
    uint32_t customIndex;   /* gl_InstanceCustomIndexEXT */
    uint32_t primitiveId;   /* gl_PrimitiveID */
    uint32_t flags;
};

struct BvhResult {
    std::vector<RtrBvhNode> nodes;   /* nodes[0] = root, DFS pre-order */
    std::vector<RtrBvhTri>  tris;    /* leaf order */
    uint32_t maxDepth = 0;           /* inner nodes on the longest root->leaf path = stack bound */
    uint32_t maxLeafSize = 0;
    float    sahCost = 0.f;
    float    boundsMin[3] = {0, 0, 0};
    float    boundsMax[3] = {0, 0, 0};
    float    boxPad = 0.f;
    float    buildMs = 0.f;
};

/* Deterministic binned-SAH build.  Returns false (with *err) on invalid input (NaN/inf corners). */
bool build_bvh(const std::vector<WorldTriangle>& tris, BvhResult& out, std::string* err);

}  // namespace rtr
