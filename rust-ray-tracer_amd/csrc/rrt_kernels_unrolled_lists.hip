// rrt_kernels_unrolled_lists.hip -- the instantiations of the render kernel for unrolled scenes (rrt_device.h:
// MODE_UNROLLED) whose primary rays take per-pixel candidate lists.  A translation unit of its own because these
// kernels are fastest under a third scheduling strategy (Makefile; measured on MI355X, 100k-triangle mesh x 100
// instances at 7680x4320 x 256 spp: iterative-maxocc 2190 ms, the default and iterative-ilp 2244 ms, max-ilp 2263 ms,
// while the kernels without lists lose 3.6 % under iterative-maxocc).
#include "rrt_render_kernel.h"

namespace rrt {

// (split items of a few pixels, G.lists)
hipError_t launch_render_unrolled_lists(bool count_visits, bool pipelined, bool scaled, const SceneDev& S, const RenderGeom& G, float* out,
                                               unsigned long long* d_counters, unsigned int* d_heads, int cu_count, uint32_t max_blocks_per_cu,
                                               hipStream_t stream) {
    switch ((scaled ? 4u : 0u) | (count_visits ? 2u : 0u) | (pipelined ? 1u : 0u)) {
        case 7u: return launch_render_variant<MODE_UNROLLED, true, true, true, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 6u: return launch_render_variant<MODE_UNROLLED, true, false, true, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 5u: return launch_render_variant<MODE_UNROLLED, false, true, true, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 4u: return launch_render_variant<MODE_UNROLLED, false, false, true, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 3u: return launch_render_variant<MODE_UNROLLED, true, true, false, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 2u: return launch_render_variant<MODE_UNROLLED, true, false, false, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 1u: return launch_render_variant<MODE_UNROLLED, false, true, false, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        default: return launch_render_variant<MODE_UNROLLED, false, false, false, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
    }
}

}  // namespace rrt
