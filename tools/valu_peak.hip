// valu_peak.hip -- measures the vector-ALU issue rate of one SIMD for the instruction mixes the traversal loop uses,
// at 1..8 wavefronts per SIMD.  Gives the denominator for "VALU utilisation" figures.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_peak.hip -o tools/valu_peak && tools/valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

#define REP8(x) x x x x x x x x

// 64 independent-ish instructions per loop trip (8 accumulators, each touched 8 times; dependent distance 8)
template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float* out, int trips, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    unsigned u0 = threadIdx.x * 2654435761u;
    for (int i = 0; i < trips; ++i) {
        if (MODE == 0) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                              : "v"(a), "v"(b));)
        } else if (MODE == 1) {  // min/max mix
            REP8(asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_min3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                              "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_min3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                              : "v"(a), "v"(b));)
        } else if (MODE == 2) {  // u16 -> f32 conversions with sub-dword operand select
            REP8(asm volatile("v_cvt_f32_u32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                              "v_cvt_f32_u32_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                              "v_cvt_f32_u32_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                              "v_cvt_f32_u32_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                              "v_cvt_f32_u32_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                              "v_cvt_f32_u32_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                              "v_cvt_f32_u32_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                              "v_cvt_f32_u32_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                              : "v"(u0));)
        } else if (MODE == 3) {  // reciprocal (quarter-rate class)
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (MODE == 4) {  // dependent chain: every instruction needs the previous result
            REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                              "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                              : "+v"(r0)
                              : "v"(a), "v"(b));)
        } else if (MODE == 5) {  // compare + select pairs
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %3, %9\n v_cndmask_b32 %4, %4, %5, vcc\n"
                              "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc\n v_cmp_lt_f32 vcc, %2, %9\n v_cndmask_b32 %3, %3, %6, vcc\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                              : "v"(a), "v"(b)
                              : "vcc");)
        } else if (MODE == 6) {  // packed f32 fma: two fmas per instruction
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, pa = {a, a}, pb = {b, b};
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                              : "v"(pa), "v"(pb));)
            r0 = p0.x + p0.y; r2 = p1.x + p1.y; r4 = p2.x + p2.y; r6 = p3.x + p3.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

template <int MODE>
static void run(const char* name, float* d_out, int cus, double ghz) {
    const int trips = 4096;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;  // 256 threads = one wave per SIMD; wps blocks per CU
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 64, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, trips, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double insts_per_simd = (double)trips * 64.0 * wps;
        const double cycles = ms * 1e-3 * ghz * 1e9;
        printf("%-28s waves/SIMD %d  %.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, wps, ms,
               cycles / insts_per_simd, ghz);
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
    run<2>("v_cvt_f32_u32_sdwa", d_out, p.multiProcessorCount, ghz);
    run<3>("v_rcp_f32", d_out, p.multiProcessorCount, ghz);
    run<4>("v_fma_f32 dependent chain", d_out, p.multiProcessorCount, ghz);
    run<5>("v_cmp + v_cndmask", d_out, p.multiProcessorCount, ghz);
    run<6>("v_pk_fma_f32", d_out, p.multiProcessorCount, ghz);
    CHECK(hipFree(d_out));
    return 0;
}
