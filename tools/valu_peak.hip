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

// One instruction form repeated on 8 registers (dependent distance 8).
#define ONE_OP_KERNEL(NAME, ASM)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(float* out, int trips, float a, float b) {                            \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        for (int i = 0; i < trips; ++i) {                                                                             \
            REP8(asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)                          \
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)        \
                              : "v"(a), "v"(b));)                                                                     \
        }                                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                           \
    }
#define A_CVT(r) "v_cvt_f32_u32 " #r ", " #r "\n"
#define A_CVTB(r) "v_cvt_f32_ubyte1 " #r ", " #r "\n"
#define A_ALIGN(r) "v_alignbit_b32 " #r ", " #r ", " #r ", %8\n"
#define A_AND(r) "v_and_b32 " #r ", %8, " #r "\n"
#define A_LSHR(r) "v_lshrrev_b32 " #r ", 16, " #r "\n"
#define A_MIN(r) "v_min_f32 " #r ", " #r ", %8\n"
#define A_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9\n"
#define A_MUL(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define A_ADD(r) "v_add_f32 " #r ", " #r ", %8\n"
#define A_ADDU(r) "v_add_u32 " #r ", " #r ", %8\n"
#define A_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9\n"
#define A_BFE(r) "v_bfe_u32 " #r ", " #r ", 4, 16\n"
#define A_XOR(r) "v_xor_b32 " #r ", %8, " #r "\n"
#define A_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n"
#define A_MOV(r) "v_mov_b32 " #r ", %8\n"
#define A_LSHLOR(r) "v_lshl_or_b32 " #r ", " #r ", 16, %8\n"
#define A_DOT2I(r) "v_dot2_i32_i16 " #r ", " #r ", %8, %9\n"
#define A_DOT2U(r) "v_dot2_u32_u16 " #r ", " #r ", %8, %9\n"
#define A_MADI16(r) "v_mad_i32_i16 " #r ", " #r ", %8, %9\n"
#define A_MAXI(r) "v_max_i32 " #r ", " #r ", %8\n"
#define A_MAX3I(r) "v_max3_i32 " #r ", " #r ", %8, %9\n"
#define A_MAD24(r) "v_mad_i32_i24 " #r ", " #r ", %8, %9\n"
#define A_FMAMIX(r) "v_fma_mix_f32 " #r ", " #r ", %8, %9 op_sel_hi:[1,0,0]\n"
ONE_OP_KERNEL(k_dot2i, A_DOT2I)
ONE_OP_KERNEL(k_dot2u, A_DOT2U)
ONE_OP_KERNEL(k_madi16, A_MADI16)
ONE_OP_KERNEL(k_maxi, A_MAXI)
ONE_OP_KERNEL(k_max3i, A_MAX3I)
ONE_OP_KERNEL(k_mad24, A_MAD24)
ONE_OP_KERNEL(k_fmamix, A_FMAMIX)
ONE_OP_KERNEL(k_cvt, A_CVT)
ONE_OP_KERNEL(k_cvtb, A_CVTB)
ONE_OP_KERNEL(k_align, A_ALIGN)
ONE_OP_KERNEL(k_and, A_AND)
ONE_OP_KERNEL(k_lshr, A_LSHR)
ONE_OP_KERNEL(k_min, A_MIN)
ONE_OP_KERNEL(k_max3, A_MAX3)
ONE_OP_KERNEL(k_mul, A_MUL)
ONE_OP_KERNEL(k_add, A_ADD)
ONE_OP_KERNEL(k_addu, A_ADDU)
ONE_OP_KERNEL(k_perm, A_PERM)
ONE_OP_KERNEL(k_bfe, A_BFE)
ONE_OP_KERNEL(k_xor, A_XOR)
ONE_OP_KERNEL(k_mullo, A_MULLO)
ONE_OP_KERNEL(k_mov, A_MOV)
ONE_OP_KERNEL(k_lshlor, A_LSHLOR)

typedef void (*one_op_kernel)(float*, int, float, float);
static void run_one(const char* name, one_op_kernel k, float* d_out, int cus, double ghz) {
    const int trips = 4096, wps = 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(cus * wps), dim3(256), 0, 0, d_out, 64, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(cus * wps), dim3(256), 0, 0, d_out, trips, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s waves/SIMD %d  %.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, wps, ms,
           ms * 1e-3 * ghz * 1e9 / ((double)trips * 64.0 * wps), ghz);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
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
    const int cus = p.multiProcessorCount;
    run_one("v_dot2_i32_i16", k_dot2i, d_out, cus, ghz);
    run_one("v_dot2_u32_u16", k_dot2u, d_out, cus, ghz);
    run_one("v_mad_i32_i16", k_madi16, d_out, cus, ghz);
    run_one("v_max_i32", k_maxi, d_out, cus, ghz);
    run_one("v_max3_i32", k_max3i, d_out, cus, ghz);
    run_one("v_mad_i32_i24", k_mad24, d_out, cus, ghz);
    run_one("v_fma_mix_f32 (f16 src0)", k_fmamix, d_out, cus, ghz);
    run_one("v_cvt_f32_u32", k_cvt, d_out, cus, ghz);
    run_one("v_cvt_f32_ubyte1", k_cvtb, d_out, cus, ghz);
    run_one("v_alignbit_b32", k_align, d_out, cus, ghz);
    run_one("v_and_b32", k_and, d_out, cus, ghz);
    run_one("v_lshrrev_b32", k_lshr, d_out, cus, ghz);
    run_one("v_min_f32", k_min, d_out, cus, ghz);
    run_one("v_max3_f32", k_max3, d_out, cus, ghz);
    run_one("v_mul_f32", k_mul, d_out, cus, ghz);
    run_one("v_add_f32", k_add, d_out, cus, ghz);
    run_one("v_add_u32", k_addu, d_out, cus, ghz);
    run_one("v_perm_b32", k_perm, d_out, cus, ghz);
    run_one("v_bfe_u32", k_bfe, d_out, cus, ghz);
    run_one("v_xor_b32", k_xor, d_out, cus, ghz);
    run_one("v_mul_lo_u32", k_mullo, d_out, cus, ghz);
    run_one("v_mov_b32", k_mov, d_out, cus, ghz);
    run_one("v_lshl_or_b32", k_lshlor, d_out, cus, ghz);
    CHECK(hipFree(d_out));
    return 0;
}
