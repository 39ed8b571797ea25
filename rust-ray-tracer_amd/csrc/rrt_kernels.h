// rrt_kernels.h -- host-callable launchers of the gfx950 kernels.
#ifndef RRT_KERNELS_H
#define RRT_KERNELS_H

#include <hip/hip_runtime_api.h>

#include "../../include/rrt_ffi.h"
#include "rrt_layout.h"

namespace rrt {

// Number of sample chunks a frame of `spp` samples per pixel is cut into (work item = tile x chunk).
uint32_t render_chunks(uint32_t width, uint32_t height, uint32_t spp, uint32_t samples_per_item);

// Share (in 64ths) of the walking lanes that must be descending for a node pass to run; 40/64 measured best
// on MI355X for both the 968-triangle and the 1M-triangle scene.
static constexpr uint32_t DEFAULT_NODE_SHARE_64THS = 40u;
// Idle lanes of a wavefront that make the streaming ray-batch kernel draw new rays.
static constexpr uint32_t DEFAULT_REFILL_MIN = 16u;

// Scheduling parameters of the traversal kernels.
struct RenderTuning {
    uint32_t node_share_64ths = DEFAULT_NODE_SHARE_64THS;
    uint32_t refill_min = DEFAULT_REFILL_MIN;
    // Render kernel: a trip's walk phase ends with at most leave_walk lanes still walking if at least leave_ready
    // lanes wait to shade; the stragglers finish their walk in the next trip.  Measured on MI355X: 10-20 % faster
    // on the large scenes (optimum 12 on the 1M-triangle scene, 20 on the 10M-triangle ones, both within 2 % of
    // each other), 1 % slower on the 968-triangle one, whose walks are short against a shading pass -- the
    // library therefore switches it off (leave_walk = 0) for BVHs of fewer than 4096 nodes.
    uint32_t leave_walk = 12u;
    uint32_t leave_ready = 16u;
    // Render kernel: the pipelined form (1: a wavefront holds two work items, and a lane that has finished its pixel
    // of the current one starts on the next one at once) or the item-at-a-time form (0).  Measured on MI355X: the
    // pipelined form is 5-9 % faster on the large scenes and 6 % slower on the 968-triangle one.
    uint32_t overlap_items = 1u;
    // Render kernel: tiles are numbered in vertical bands of this many tiles' width (0: row by row over the whole
    // frame).  The items in flight at one time then cover a compact block of the image, whose rays share BVH nodes
    // in the L1 and L2 caches.  Measured on MI355X (16): 6-7 % faster on the 1M- and 10M-triangle scenes, 5 % slower
    // on the 968-triangle one, where a band gathers the few expensive tiles.
    uint32_t band_tiles = 16u;
    // Render kernel: the scaled form of the slab arithmetic (plane indices used as denormal floats: no converts) on
    // scenes for which the upload found it valid (SceneDev::sigma != 0); 0 keeps the convert form.  Images are the
    // same bit for bit either way (both box tests are conservative; the primitive tests decide).  Measured on MI355X:
    // 3 % faster on the 1M-triangle scene, 6-7 % on the 968-triangle and the unrolled instanced one.
    uint32_t scaled_slab = 1u;
    // Render kernel, scenes without instances: 1: a work item is a few pixels times as many of their sample chunks as make 64 lanes
    // (all of the frame's chunks when there are at most 32); 0: 64 pixels times one chunk.  Same image either way.
    uint32_t split_items = 0u;
    // Render kernel, split items of at most eight pixels: 1: the primary rays of a pixel are resolved against a list of
    // the primitives the pixel's pyramid of camera rays can meet, collected by a pass over the frame's pixels before
    // the render kernel, instead of walking the tree.  Same image either way.
    uint32_t pixel_lists = 0u;
    // ... a pixel with more candidates than this (at most PIXEL_LIST_MAX) gets no list
    uint32_t pixel_list_len = PIXEL_LIST_MAX;
    // ... the pipelined form with lists as compiled for 7 workgroups per CU (72 registers, some of them spilled) or for
    // 6 (84 registers): same image, and which one is faster depends on the scene
    uint32_t list_waves = 7u;
    // Persistent kernels: fewer resident workgroups per CU than the kernel allows (0: all of them; occupancy experiments).
    uint32_t blocks_per_cu = 0u;
    // Ray batches: 1 runs the plain one-lane-per-ray kernel instead of the streaming one (experiments).
    uint32_t intersect_static = 0u;
};

// d_counters: 32 x u64 in device memory, zero before the launch: [0..6] {rays, node visits, primitive tests, wave-level
//             node passes, primitive passes, shading passes, refill passes}, [8..11] the work-queue heads (d_heads),
//             [12..13] idle lane slots of node passes, [14] pixels that got a candidate leaf list; the kernels ADD to them.
// d_heads:    8 x u32 work-queue heads, zero before the launch.
// n_chunks (1 .. spp) sample chunks whose lengths differ by at most one (RenderGeom).
// n_chunks > 1: the kernel writes n_chunks partial images (each width*height*3 floats) to d_partial and a second
// kernel sums them into d_accum in chunk order; n_chunks == 1: the kernel writes d_accum directly.
hipError_t launch_render(const SceneDev& S, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                         float* d_accum, float* d_partial, uint32_t n_chunks, unsigned long long* d_counters,
                         unsigned int* d_heads, int cu_count, bool count_visits, const RenderTuning& tuning,
                         hipStream_t stream, uint32_t* d_lists = nullptr);
// Whether a launch with these parameters resolves primary rays against per-pixel leaf lists; it then needs d_lists:
// width * height * PIXEL_LIST_DWORDS dwords of device memory (written by a pass that launch_render runs first).
bool render_takes_pixel_lists(const SceneDev& S, uint32_t width, uint32_t height, uint32_t n_chunks, const RenderTuning& tuning);
// d_head: one u32, zero before the launch (the batch's ray counter).
hipError_t launch_intersect(const SceneDev& S, const rrt_ray* d_rays, size_t n, float tmin, float tmax, rrt_hit* d_out,
                            unsigned long long* d_counters, unsigned int* d_head, int cu_count, bool count_visits,
                            const RenderTuning& tuning, hipStream_t stream);
hipError_t launch_tonemap(const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_rgb8,
                          hipStream_t stream);

// d_dst[i] += d_src[i] for n floats (both 16-byte aligned).
hipError_t launch_add(float* d_dst, const float* d_src, size_t n, hipStream_t stream);

// d_bad: 2 x u64, zero before the launch; receives the number of float bit patterns (of all 2^32) on which the
// kernels' short reciprocal / square root differ from the IEEE operations.
hipError_t launch_check_arith(unsigned long long* d_bad, hipStream_t stream);

}  // namespace rrt
#endif
