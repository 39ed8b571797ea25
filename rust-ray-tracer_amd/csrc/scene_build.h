// scene_build.h -- turns the caller's rrt_scene_desc into the flat records + BVH that are uploaded to HBM.
#ifndef RRT_SCENE_BUILD_H
#define RRT_SCENE_BUILD_H

#include <string>
#include <vector>

#include "../../include/rrt_ffi.h"
#include "bvh_builder.h"
#include "rrt_layout.h"

namespace rrt {

// A BVH whose construction is left to the GPU builder: its primitives (in submission order; a triangle's `index`
// is its ordinal among this BVH's triangles, whose records start at tri_base in the same order), the node range
// reserved for it (prims.size() nodes from node_base; the root is the first of them) and its grid.
struct DeferredBuild {
    std::vector<BuildPrim> prims;
    uint32_t node_base = 0;
    uint32_t tri_base = 0, tri_count = 0;
    uint32_t budget = 0;   // node levels the BVH may have
    int mesh = -1;         // mesh index, -1: the top-level BVH
    Grid grid{};
    // Triangle runs the host has not expanded (top-level BVH only): the GPU makes their records and boxes from the
    // caller's vertices, which must stay valid until the upload is complete.  They are primitives prims.size() ..
    // prims.size() + raw_tris - 1 of the BVH and its last raw_tris triangle ordinals; `pad` widens their boxes.
    std::vector<RawTriRun> raw_runs;
    uint32_t raw_tris = 0;
    float pad = 0.f;
};

struct HostScene {
    std::vector<QNode> qnodes;    // every BVH's nodes on its 16-bit grid: what is uploaded (ranges left to the GPU builder stay blank)
    Grid grid{};                  // grid of the top-level BVH
    std::vector<TriRec> tris;
    std::vector<SphereRec> spheres;
    std::vector<RectRec> rects;
    std::vector<InstRec> insts;
    std::vector<MatRec> mats;
    std::vector<LightRec> lights;
    int32_t root = 0;
    uint32_t prim_count = 0;
    uint32_t depth = 0;            // node levels of the deepest path (top level + deepest mesh)
    uint32_t top_depth = 0;        // node levels of the top-level BVH (0 while its build is deferred)
    uint32_t deepest_mesh = 0;     // node levels of the deepest host-built mesh BVH
    std::vector<DeferredBuild> deferred;
    // What exists on the device only, behind the arrays above: the node range of a deferred top-level BVH over raw
    // triangle runs, and those triangles' records (both written by the GPU builder).
    uint32_t tail_nodes = 0, tail_tris = 0;
    bool has_instances = false;    // instance leaves in the top-level BVH (two-level walk)
    uint32_t unroll_shift = 0;     // != 0: instances unrolled into the top-level tree (SceneDev::unroll_shift)
    uint32_t unrolled_instances = 0;
    CameraRec cam{};
    float background[3] = {0.f, 0.f, 0.f};
};

// Returns RRT_OK or RRT_ERR_INVALID with a message.  `threads` host threads are used for the BVH build.
// device_build_min > 0: BVHs over at least that many primitives are not built here but listed in out.deferred.
// unroll: instances may be written out as world-space copies of their mesh's tree inside the top-level BVH (done when
// every instanced mesh is host-built and triangle-only and the copies fit the node budget); false: always two-level.
rrt_status build_host_scene(const rrt_scene_desc* desc, HostScene& out, std::string& err, unsigned threads,
                            uint32_t device_build_min = 0, bool unroll = true);

// Grid of 65536 planes per axis over the box [mn, mx] of a BVH's root.
Grid make_grid(const float* mn, const float* mx);

}  // namespace rrt
#endif
