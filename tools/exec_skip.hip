// exec_skip.hip -- does a vector-ALU instruction of a 64-lane wavefront cost less when part of the lanes is switched
// off?  Runs the same instruction stream under different EXEC masks (all lanes, one half, one quarter, every second
// lane, one lane) and prints the issue cost per wave-instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/exec_skip.hip -o tools/exec_skip && tools/exec_skip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                    \
        }                                                               \
    } while (0)

#define REP8(x) x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k_masked(float* out, int trips, float a, float b, unsigned long long mask) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    const unsigned lane = threadIdx.x & 63u;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < trips; ++i) {
            if (MODE == 0) {
                REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                  : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                                  : "v"(a), "v"(b));)
            } else if (MODE == 1) {
                REP8(asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_min3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                                  "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_min3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                                  : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                                  : "v"(a), "v"(b));)
            } else {
                REP8(asm volatile("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_add_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9\n"
                                  "v_add_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9\n"
                                  : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                                  : "v"(a), "v"(b));)
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

template <int MODE>
static void run(const char* name, float* d_out, int cus, double ghz) {
    const int trips = 4096, wps = 4;
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFFFFF00000000ull, 0xFFFFull, 0xFFFF0000FFFF0000ull,
                                        0x5555555555555555ull, 0x00000000FFFF0000ull, 1ull};
    const char* names[] = {"all 64", "lanes 0-31", "lanes 32-63", "lanes 0-15", "lanes 16-31 + 48-63", "even lanes", "lanes 16-31", "lane 0"};
    for (int m = 0; m < 8; ++m) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_masked<MODE>, dim3(cus * wps), dim3(256), 0, 0, d_out, 64, 1.0001f, 0.5f, masks[m]);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_masked<MODE>, dim3(cus * wps), dim3(256), 0, 0, d_out, trips, 1.0001f, 0.5f, masks[m]);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-18s %-22s %.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz, %d waves/SIMD)\n", name, names[m], ms,
               ms * 1e-3 * ghz * 1e9 / ((double)trips * 64.0 * wps), ghz, wps);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    printf("%s  CUs %d  clock %.2f GHz\n", p.gcnArchName, p.multiProcessorCount, ghz);
    float* d_out;
    CHECK(hipMalloc(&d_out, (size_t)p.multiProcessorCount * 8 * 256 * sizeof(float)));
    run<0>("v_fma_f32", d_out, p.multiProcessorCount, ghz);
    run<1>("min/max/min3/max3", d_out, p.multiProcessorCount, ghz);
    run<2>("v_add/v_mul", d_out, p.multiProcessorCount, ghz);
    CHECK(hipFree(d_out));
    return 0;
}
