/* rtr_bvh.h — device BVH build / refit (internal to librtr_hip.so); see rtr_bvh.hip. */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/rtr_types.h"

namespace rtrdev {

struct PrimRef { uint32_t customIndex, primitiveId, flags, _pad; };                     /* canonical (instance, primitive) order */
struct InstanceRef { float transform[12]; uint32_t vertexOffset, indexOffset, _pad[2]; };  /* indexed by customIndex */

struct BvhInputs {
    const PrimRef* prims;
    const InstanceRef* instances;
    const RtrVertex* vertices;
    const uint32_t* indices;
};

struct BvhDeviceArrays {          /* persistent: the tree + what a refit needs */
    float4* nodesF;               /* numNodes x 4: builder-side nodes with fp32 planes (rtr::BvhNodeF); the fit / refit works here */
    uint4*  nodes;                /* numNodes x 2: RtrBvhNode (16-bit planes on *grid), what the traversal reads */
    RtrBvhGrid* grid;             /* 1 record, rewritten by every build / refit */
    float4* tris;                 /* numPrims x 3, leaf order */
    float4* boxMin; float4* boxMax;   /* per leaf-ordered primitive */
    int32_t* parent;              /* per node: (parentIndex << 1) | slot, -1 root, -2 not part of the tree */
    uint32_t* counters; uint32_t* depth;
    uint32_t* slotOfPrim;         /* canonical primitive -> leaf-order slot */
    uint32_t* red;                /* 8 words: centroid bounds, max |coordinate| (float bits) in [6], max depth in [7] */
};

struct BvhScratch {               /* build only */
    float4* trisCanon; float4* minCanon; float4* maxCanon;
    unsigned long long* keysIn; unsigned long long* keysOut;
    int2* range; int2* rawChild;
    void* sortTemp; size_t sortTempBytes;
};

size_t bvh_sort_temp_bytes(uint32_t numPrims);
/* LBVH build: numPrims >= 16.  Node array has numPrims-1 entries (entries inside collapsed subtrees are unused). */
hipError_t bvh_build_lbvh(const BvhInputs& in, uint32_t numPrims, const BvhDeviceArrays& a, const BvhScratch& t, hipStream_t s);
/* Refit after transforms changed: recompute world records in place (leaf order), re-fit every box, re-derive the grid
 * from the new root bounds and re-quantise. */
hipError_t bvh_refit(const BvhInputs& in, uint32_t numPrims, uint32_t numNodes, const BvhDeviceArrays& a, hipStream_t s);

/* The 4-wide view of a finished (quantised) tree that the any-hit kernel walks: numNodes x 4 uint4, see k_wide_nodes.
 * parentOrNull: the refit parent array (entries outside the tree are skipped) or null. */
/* sets grid->wideCentreXY / Z (k_wide_centre_*; sums4 = bvh_wide_scratch_words() x u64 of scratch) and writes the 4-wide records about it */
size_t bvh_wide_scratch_words();
/* shapeOrNull: per BVH2 node, which entries its wide record opens (bvh_build.h collapse_wide); null = the greedy rule */
hipError_t bvh_make_wide(const uint4* nodes, uint32_t numNodes, const int32_t* parentOrNull, RtrBvhGrid* grid, const uint8_t* shapeOrNull, uint4* wide, unsigned long long* sums4, hipStream_t s);
/* out[remap[i]] = in[i] with inner child codes renumbered through remap (a permutation of 0..numNodes-1, remap[0] == 0) */
hipError_t bvh_permute_wide(const uint4* in, uint32_t numNodes, const uint32_t* remap, uint4* out, hipStream_t stream);

}  // namespace rtrdev
