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

/* Builder-side node with fp32 planes (what the SAH build, the device LBVH fit and a refit work on); quantize_nodes()
 * turns an array of these into the 32-B RtrBvhNode the traversal reads.
 *   f[0..2] left min xyz, f[3..5] left max xyz, f[6..8] right min xyz, f[9..11] right max xyz */
struct BvhNodeF {
    float   f[12];
    int32_t child[2];
    int32_t _pad[2];
};
static_assert(sizeof(BvhNodeF) == 64, "BvhNodeF is 4 x float4 on the device");

struct BvhResult {
    std::vector<RtrBvhNode> nodes;   /* nodes[0] = root, DFS pre-order; planes on `grid` */
    RtrBvhGrid grid = {};
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

/* Scene grid from the root node's two (padded) child boxes, then every plane quantised outward (rtr_math.h). */
void quantize_nodes(const BvhNodeF* in, size_t count, RtrBvhGrid& grid, RtrBvhNode* out);

}  // namespace rtr
