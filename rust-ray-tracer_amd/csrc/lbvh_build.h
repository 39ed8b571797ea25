// lbvh_build.h -- BVH construction on the GPU (linear BVH over Morton-sorted primitives).
//
// Takes the place of the reference's recursive host builder `BVHNode::new` (src/geometry/mod.rs:161-199) for BVHs
// that are too large to build on the host in reasonable time; produces the same node format as the host builder
// (rrt_layout.h), so the traversal kernels do not know which builder ran.
#ifndef RRT_LBVH_BUILD_H
#define RRT_LBVH_BUILD_H

#include <hip/hip_runtime_api.h>

#include "bvh_builder.h"
#include "rrt_layout.h"

namespace rrt {

struct LbvhResult {
    uint32_t depth = 0;   // node levels of the built tree; 0: the tree did not fit `max_depth` levels (nothing usable was written)
    float build_ms = 0.f; // device time of the build (sort + tree + boxes + triangle reorder)
    uint32_t clusters = 0; // subtrees the upper levels were rebuilt over (0: the plain radix tree was kept)
    float top_ms = 0.f;    // host time of that rebuild
};

// Builds the BVH of h_prims[0..n) (n >= 2) into d_nodes[node_base .. node_base + n - 1): the root is node_base.
// Triangle primitives carry their ordinal among the BVH's triangles in `index`; their records are expected at
// d_tris[tri_base + ordinal] and are reordered in place into leaf order.  Child boxes are put on `grid`.
// cluster_size > 0: the radix tree is cut where subtrees hold at most that many primitives, and the levels above the
// cut -- where splitting at Morton code boundaries costs the most -- are rebuilt on the host with the surface-area
// heuristic over the subtrees' boxes (a few per cent of the primitives' number; `threads` host threads).
// runs / n_runs (may be empty): triangles of the BVH that the host has NOT expanded.  The GPU reads the caller's
// vertices, writes their records (reference src/geometry/triangle.rs:17-18,28-29: v0 and the two edges) to
// d_tris[tri_base + ordinal] and their boxes (triangle.rs:82-94, widened by `pad` like the host's) behind the n
// host-expanded primitives; their ordinals follow those of the host-expanded triangles, in run order.  tri_count counts
// both kinds.
// Blocks until the build is complete.
hipError_t lbvh_build(const BuildPrim* h_prims, uint32_t n, const Grid& grid, QNode* d_nodes, uint32_t node_base,
                      TriRec* d_tris, uint32_t tri_base, uint32_t tri_count, uint32_t max_depth, uint32_t cluster_size,
                      unsigned threads, LbvhResult* out, const RawTriRun* runs = nullptr, size_t n_runs = 0, float pad = 0.f);

}  // namespace rrt
#endif
