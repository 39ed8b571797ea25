// scene_build.h -- turns the caller's rrt_scene_desc into the flat records + BVH that are uploaded to HBM.
#ifndef RRT_SCENE_BUILD_H
#define RRT_SCENE_BUILD_H

#include <string>
#include <vector>

#include "../../include/rrt_ffi.h"
#include "rrt_layout.h"

namespace rrt {

struct HostScene {
    std::vector<BvhNode> nodes;   // builder output (full-precision boxes), host only
    std::vector<QNode> qnodes;    // the same nodes on their BVH's 16-bit grid: what is uploaded
    Grid grid{};                  // grid of the top-level BVH
    std::vector<TriRec> tris;
    std::vector<SphereRec> spheres;
    std::vector<RectRec> rects;
    std::vector<InstRec> insts;
    std::vector<MatRec> mats;
    std::vector<LightRec> lights;
    int32_t root = 0;
    uint32_t prim_count = 0;
    uint32_t depth = 0;
    bool has_instances = false;
    CameraRec cam{};
    float background[3] = {0.f, 0.f, 0.f};
};

// Returns RRT_OK or RRT_ERR_INVALID with a message.  `threads` host threads are used for the BVH build.
rrt_status build_host_scene(const rrt_scene_desc* desc, HostScene& out, std::string& err, unsigned threads);

}  // namespace rrt
#endif
