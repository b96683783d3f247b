/* bvh_build.h — host-side BVH builder of the product (replaces the driver-opaque
 * vkCmdBuildAccelerationStructuresKHR of reference src/vulkan/raytracing/blas.cppm:75-167 and
 * tlas.cppm:44-149; the reference asks its driver for ePreferFastTrace, blas.cppm:115 — here that is: binned SAH,
 * then insertion-based optimisation of the finished tree, then a cost-driven collapse into the 4-wide view).
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

/* How much build time is spent on trace speed.  The defaults are the product's (rtr_scene_create); RTR_BVH_* environment
 * variables override them for experiments (build_options_from_env). */
struct BuildOptions {
    uint32_t bins = 32;               /* SAH bins per axis */
    uint32_t leafTarget = 4;          /* SAH may stop at <= this many triangles */
    uint32_t maxLeaf = 8;             /* hard cap (RTR_BVH_MAX_LEAF: 3 bits in the leaf code) */
    float    costTraverse = 1.0f, costIntersect = 1.0f;
    /* insertion-based optimisation (Bittner, Hapala, Havran 2013): subtrees are taken out and put back where they raise the
     * tree's SAH cost least.  passes x fraction of the nodes, worst first */
    uint32_t reinsertPasses = kReinsertAuto;      /* kReinsertAuto: decided per scene by a probe (build_bvh); 0: never; n: n passes */
    float    reinsertFraction = 1.0f;
    static constexpr uint32_t kReinsertAuto = 0xffffffffu;
    uint32_t wideGreedy = 0;          /* 1: no shapes — the 4-wide view is collapsed by the greedy rule (open the largest box), as device-built trees are */
};
BuildOptions build_options_from_env();

struct BvhResult {
    std::vector<RtrBvhNode> nodes;   /* nodes[0] = root, DFS pre-order; planes on `grid` */
    RtrBvhGrid grid = {};
    std::vector<RtrBvhTri>  tris;    /* leaf order */
    /* per node: how the 4-wide view opens it when the node roots a wide record (wide_shape codes, see collapse_wide); empty = greedy */
    std::vector<uint8_t> wideShape;
    uint32_t maxDepth = 0;           /* inner nodes on the longest root->leaf path = stack bound */
    uint32_t maxLeafSize = 0;
    float    sahCost = 0.f;
    float    sahCostBeforeOpt = 0.f;
    float    wideCost = 0.f, wideCostGreedy = 0.f;   /* SAH-style cost of the 4-wide view: chosen collapse / greedy collapse */
    float    boundsMin[3] = {0, 0, 0};
    float    boundsMax[3] = {0, 0, 0};
    float    boxPad = 0.f;
    float    buildMs = 0.f, optMs = 0.f;
    float    optProbeGain = 0.f;     /* auto mode: relative SAH-cost gain of the probe (the worst tenth of the nodes re-inserted) */
    uint32_t optPasses = 0;          /* full re-insertion passes that were run and kept */
};

/* Deterministic build.  Returns false (with *err) on invalid input (NaN/inf corners). */
bool build_bvh(const std::vector<WorldTriangle>& tris, BvhResult& out, std::string* err, const BuildOptions& opt = build_options_from_env());

/* Scene grid from the root node's two (padded) child boxes, then every plane quantised outward (rtr_math.h). */
void quantize_nodes(const BvhNodeF* in, size_t count, RtrBvhGrid& grid, RtrBvhNode* out);

/* The 4-wide view, decided by cost.  A wide record rooted at BVH2 node n starts from n's two children and may open inner ones into
 * their own children while slots are free; WHICH ones is the `shape` of n:
 *   0 (L R)   1 (LL LR R)   2 (L RL RR)   3 (LL LR RL RR)   4 (LLL LLR LR R)   5 (LL LRL LRR R)   6 (L RLL RLR RR)   7 (L RL RRL RRR)
 * (children listed in slot order).  collapse_wide picks, for EVERY node, the shape that minimises
 *   cost(n) = area(n) * costTraverse + sum over the record's children c of (c inner ? cost(c) : area(c) * tris(c) * costIntersect)
 * by dynamic programming over the slots a subtree may use (Ylitie, Karras, Laine 2017, section 3.1, for 4 slots), and reports the
 * root's cost next to the cost of the greedy rule (open the largest box).  The device kernel k_wide_nodes follows the shapes
 * (kernels/rtr_bvh.hip); without them it falls back to the greedy rule (device-built trees). */
void collapse_wide(const BvhNodeF* nodes, size_t count, float costTraverse, float costIntersect, std::vector<uint8_t>& shape, float* costOpt, float* costGreedy);

/* Host restatement of k_wide_centre_* + k_wide_nodes + the breadth-first order (kernels/rtr_bvh.hip, rtr_api.cpp make_wide_nodes):
 * the RtrWideNode array the any-hit kernel would walk for these quantised nodes, and the wide centre in `grid`.  Used by
 * rtr_host_build_bvh (CPU-only tests walk the wide view with the oracle) and checked against the device's records in the GPU tests. */
void make_wide_host(const RtrBvhNode* nodes, size_t count, const uint8_t* shapeOrNull, RtrBvhGrid& grid, std::vector<RtrWideNode>& wide);

}  // namespace rtr
