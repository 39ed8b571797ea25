// bvh_builder.h -- host-side flat BVH2 builder (binned SAH, bounded depth).
//
// Takes the place of the reference's pointer-tree builder `BVHNode::new` (src/geometry/mod.rs:161-199:
// median split on a random axis in {x, y}, one primitive per leaf) and realises the flat layout its
// unfinished src/bvh/mod.rs:14-26 sketches (a node carries both children's boxes and indices).
// The topology is free: closest hit is defined as the minimum of (t, primitive id), so any
// conservative hierarchy returns the same answer.
#ifndef RRT_BVH_BUILDER_H
#define RRT_BVH_BUILDER_H

#include <cstdint>
#include <vector>

#include "../../include/rrt_ffi.h"
#include "rrt_layout.h"

namespace rrt {

struct BuildPrim {
    float bmin[3], bmax[3];  // bounding box (reference formula, plus the builder's conservative pad)
    uint32_t kind;           // KIND_TRI / KIND_SPHERE / KIND_RECT / KIND_INST
    uint32_t index;          // record index in the typed array (triangles: index into the UNORDERED list)
};

// A run of the caller's triangles that the GPU turns into records and boxes itself (host memory, valid until the build
// returns); ids are id0, id0 + 1, ...
struct RawTriRun {
    const rrt_triangle* src;
    uint32_t count;
    uint32_t id0;
};

struct BuildResult {
    int32_t root = 0;      // always an inner node index (relative to node_base)
    uint32_t depth = 0;    // deepest leaf, in node levels
    float root_min[3], root_max[3];
};

// Builds a BVH over prims[0..n) and appends its nodes to `nodes`.  Indices in the produced nodes are absolute: offset
// by `index_base` (the index the first appended node will have in the scene's node array; negative: nodes.size() at
// entry).  Triangle primitives are emitted in leaf
// order into `tri_order` (append): a triangle leaf references tri_base + position in that order.
// max_depth bounds the number of node levels (>= ceil(log2(n)) + 2 required).
BuildResult build_bvh(std::vector<BuildPrim>& prims, std::vector<BvhNode>& nodes, std::vector<uint32_t>& tri_order,
                      uint32_t tri_base, uint32_t max_depth, unsigned threads, int64_t index_base = -1);

}  // namespace rrt
#endif
