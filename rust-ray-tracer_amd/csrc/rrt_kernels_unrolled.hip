// rrt_kernels_unrolled.hip -- the instantiations of the render kernel for scenes whose instances are unrolled into one
// world-space tree (rrt_device.h: MODE_UNROLLED).  A translation unit of its own because these kernels are faster under
// the compiler's default scheduling strategy and the others under max-ilp (Makefile).
#include "rrt_render_kernel.h"

namespace rrt {

hipError_t launch_render_unrolled(bool count_visits, bool pipelined, bool scaled, const SceneDev& S, const RenderGeom& G, float* out,
                                  unsigned long long* d_counters, unsigned int* d_heads, int cu_count, uint32_t max_blocks_per_cu,
                                  hipStream_t stream) {
    switch ((scaled ? 4u : 0u) | (count_visits ? 2u : 0u) | (pipelined ? 1u : 0u)) {
        case 7u: return launch_render_variant<MODE_UNROLLED, true, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 6u: return launch_render_variant<MODE_UNROLLED, true, false, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 5u: return launch_render_variant<MODE_UNROLLED, false, true, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 4u: return launch_render_variant<MODE_UNROLLED, false, false, true>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 3u: return launch_render_variant<MODE_UNROLLED, true, true, false>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 2u: return launch_render_variant<MODE_UNROLLED, true, false, false>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        case 1u: return launch_render_variant<MODE_UNROLLED, false, true, false>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
        default: return launch_render_variant<MODE_UNROLLED, false, false, false>(S, G, out, d_counters, d_heads, cu_count, max_blocks_per_cu, stream);
    }
}

}  // namespace rrt
