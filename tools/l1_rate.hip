// l1_rate.hip -- how fast does a CU's vector L1 serve 16-byte-per-lane loads, as a function of how many distinct
// 128-byte lines the 64 lanes of one load touch?  The traversal kernels fetch a 32-byte node as two such loads per
// lane, and lanes that walk different parts of the tree touch a line each.
// Every wavefront loads from a small table (16 KB: L1-resident) in an unrolled loop of independent loads; the lane
// stride decides the lines per instruction: 0 -> 1 line, 1 (16 B) -> 8 lines, 2 -> 16, 4 -> 32, 8 (128 B) -> 64.
//   hipcc --offload-arch=gfx950 -O2 tools/l1_rate.hip -o tools/l1_rate && tools/l1_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

static constexpr uint32_t TABLE_U4 = 1024;  // 16 KB of uint4

__global__ __launch_bounds__(256) void k_loads(const uint4* __restrict__ table, uint32_t lane_stride, int trips, uint32_t active_lanes,
                                               uint4* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (lane < active_lanes) {
        uint32_t base = (lane * lane_stride + (threadIdx.x >> 6) * 17u) & (TABLE_U4 - 1u);
        for (int i = 0; i < trips; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint4 v = table[(base + (uint32_t)k * 64u) & (TABLE_U4 - 1u)];
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
            base = (base + 8u * 64u + 8u) & (TABLE_U4 - 1u);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// The same loop with narrower loads (W dwords per lane): is a load instruction's cost its bytes or its lanes?
template <int W>
__global__ __launch_bounds__(256) void k_width(const uint32_t* __restrict__ table, int trips, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t acc = 0;
    uint32_t base = (lane * 4u + (threadIdx.x >> 6) * 68u) & (TABLE_U4 * 4u - 1u);  // lanes 16 bytes apart
    for (int i = 0; i < trips; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t* p = table + ((base + (uint32_t)k * 256u) & (TABLE_U4 * 4u - 1u));
            if (W == 1) { acc ^= *p; }
            if (W == 2) { uint2 v = *reinterpret_cast<const uint2*>(p); acc ^= v.x ^ v.y; }
            if (W == 3) { typedef uint32_t u3 __attribute__((ext_vector_type(3))); const u3 v = *reinterpret_cast<const u3*>(p); acc ^= v.x ^ v.y ^ v.z; }
            if (W == 4) { uint4 v = *reinterpret_cast<const uint4*>(p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
        base = (base + 8u * 256u + 32u) & (TABLE_U4 * 4u - 1u);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int W>
static void run_width(const uint32_t* table, uint32_t* out, int blocks, int blocks_per_cu, double ghz, hipEvent_t e0, hipEvent_t e1) {
    const int trips = 2048;
    hipLaunchKernelGGL(k_width<W>, dim3(blocks), dim3(256), 0, 0, table, 64, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_width<W>, dim3(blocks), dim3(256), 0, 0, table, trips, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_cu = (double)blocks_per_cu * 4.0 * trips * 8.0;
    printf("waves/SIMD %d  %d dword(s) per lane, lanes 16 B apart : %7.2f cycles per load instruction per CU\n", blocks_per_cu, W,
           ms * 1e-3 * ghz * 1e9 / insts_per_cu);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = 2.4;
    uint4 *table, *out;
    CHECK(hipMalloc(&table, TABLE_U4 * sizeof(uint4)));
    CHECK(hipMemset(table, 1, TABLE_U4 * sizeof(uint4)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("# %d CUs; cycles are wall time x %.1f GHz; one load instruction = 64 lanes x 16 bytes\n", cus, ghz);
    for (int blocks_per_cu : {4, 8}) {
        const int blocks = cus * blocks_per_cu;
        CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(uint4)));
        for (uint32_t active : {64u, 32u, 16u}) {
            for (uint32_t stride : {0u, 1u, 2u, 4u, 8u}) {
                const int trips = 2048;
                hipLaunchKernelGGL(k_loads, dim3(blocks), dim3(256), 0, 0, table, stride, 64, active, out);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_loads, dim3(blocks), dim3(256), 0, 0, table, stride, trips, active, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double insts_per_cu = (double)blocks_per_cu * 4.0 * trips * 8.0;  // wave-level load instructions per CU
                const double cyc = ms * 1e-3 * ghz * 1e9 / insts_per_cu;
                const uint32_t span = stride * 16u * (active - 1u) + 16u;
                const uint32_t lines = stride == 0u ? 1u : (span + 127u) / 128u;
                printf("waves/SIMD %d  active lanes %2u  lane stride %3u B  lines/inst %2u : %7.2f cycles per load instruction per CU  (%5.2f per line, %6.1f B/clk/CU)\n",
                       blocks_per_cu, active, stride * 16u, lines, cyc, cyc / lines, active * 16.0 / cyc);
            }
        }
        run_width<1>((const uint32_t*)table, (uint32_t*)out, blocks, blocks_per_cu, ghz, e0, e1);
        run_width<2>((const uint32_t*)table, (uint32_t*)out, blocks, blocks_per_cu, ghz, e0, e1);
        run_width<3>((const uint32_t*)table, (uint32_t*)out, blocks, blocks_per_cu, ghz, e0, e1);
        run_width<4>((const uint32_t*)table, (uint32_t*)out, blocks, blocks_per_cu, ghz, e0, e1);
        CHECK(hipFree(out));
    }
    return 0;
}
