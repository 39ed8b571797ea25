// rrt_kernels.h -- host-callable launchers of the gfx950 kernels.
#ifndef RRT_KERNELS_H
#define RRT_KERNELS_H

#include <hip/hip_runtime_api.h>

#include "../../include/rrt_ffi.h"
#include "rrt_layout.h"

namespace rrt {

// d_counters: 3 x u64 in device memory {rays, node visits, primitive tests}; the kernel ADDS to them.
hipError_t launch_render(const SceneDev& S, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                         float* d_accum, unsigned long long* d_counters, bool count_visits, hipStream_t stream);
hipError_t launch_intersect(const SceneDev& S, const rrt_ray* d_rays, size_t n, float tmin, float tmax, rrt_hit* d_out,
                            hipStream_t stream);
hipError_t launch_tonemap(const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_rgb8,
                          hipStream_t stream);

}  // namespace rrt
#endif
