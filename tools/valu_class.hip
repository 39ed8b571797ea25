// valu_class.hip -- issue cost of single vector-instruction forms in SHADER CYCLES (s_memtime), so that the figure does
// not depend on the clock the chip holds under the test's own load.  Four wavefronts per SIMD run the same stream; a
// wavefront's elapsed ticks / (instructions x wavefronts per SIMD) = cycles per wave-instruction per SIMD.
// Also checks that a 16-bit integer reinterpreted as a float (a denormal, value q * 2^-149) goes through v_fma_f32
// at the same cost and with the exact product.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_class.hip -o tools/valu_class && tools/valu_class
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

#define REP8(x) x x x x x x x x

// 64 instructions per trip on 8 registers (dependent distance 8); %8, %9 inputs
#define ONE_OP_KERNEL(NAME, ASM)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(float* out, unsigned long long* ticks, int trips, float a, float b) {   \
        float r0 = threadIdx.x * 1e-40f, r1 = r0 * 2, r2 = r0 * 3, r3 = r0 * 4, r4 = r0 * 5, r5 = r0 * 6, r6 = r0 * 7,  \
              r7 = r0 * 8;                                                                                              \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                     \
        for (int i = 0; i < trips; ++i) {                                                                               \
            REP8(asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7)                            \
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)          \
                              : "v"(a), "v"(b));)                                                                       \
        }                                                                                                               \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                             \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                              \
    }

#define A_MOV(r) "v_mov_b32 " #r ", %8\n"
#define A_AND(r) "v_and_b32 " #r ", %8, " #r "\n"
#define A_LSHR(r) "v_lshrrev_b32 " #r ", 16, " #r "\n"
#define A_ADD(r) "v_add_f32 " #r ", " #r ", %8\n"
#define A_MUL(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define A_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define A_FMAC(r) "v_fmac_f32 " #r ", %8, %9\n"
#define A_FMA_DEN(r) "v_fma_f32 " #r ", %9, %8, " #r "\n"   /* %9 is a denormal in the run that uses this */
#define A_FMA_CLAMP(r) "v_fma_f32 " #r ", " #r ", %8, %9 clamp\n"
#define A_CVT_SDWA(r) "v_cvt_f32_u32_sdwa " #r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define A_MOV_SDWA(r) "v_mov_b32_sdwa " #r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define A_MAX(r) "v_max_f32 " #r ", " #r ", %8\n"
#define A_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9\n"
#define A_ALIGN(r) "v_alignbit_b32 " #r ", " #r ", " #r ", %8\n"
#define A_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define A_CMP(r) "v_cmp_lt_f32 vcc, " #r ", %8\n"
#define A_XOR(r) "v_xor_b32 " #r ", %8, " #r "\n"
#define A_SUB(r) "v_sub_f32 " #r ", " #r ", %8\n"
#define A_MED3(r) "v_med3_f32 " #r ", " #r ", %8, %9\n"
#define A_MAXU(r) "v_max_u32 " #r ", " #r ", %8\n"
#define A_MINI16(r) "v_pk_min_i16 " #r ", " #r ", %8\n"
#define A_PKADD16(r) "v_pk_add_u16 " #r ", " #r ", %8\n"
#define A_PKFMAF16(r) "v_pk_fma_f16 " #r ", " #r ", %8, %9\n"
#define A_PKMULF16(r) "v_pk_mul_f16 " #r ", " #r ", %8\n"
#define A_PKMAXF16(r) "v_pk_max_f16 " #r ", " #r ", %8\n"
#define A_MADMIX(r) "v_fma_mix_f32 " #r ", %8, %9, " #r " op_sel_hi:[1,0,0]\n"
#define A_SUBREV_U(r) "v_subrev_u32 " #r ", %8, " #r "\n"
#define A_LSHL_ADD(r) "v_lshl_add_u32 " #r ", " #r ", 3, %8\n"
#define A_ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n"
#define A_OR3(r) "v_or3_b32 " #r ", " #r ", %8, %9\n"
#define A_BFI(r) "v_bfi_b32 " #r ", %8, " #r ", %9\n"

ONE_OP_KERNEL(k_mov, A_MOV)
ONE_OP_KERNEL(k_and, A_AND)
ONE_OP_KERNEL(k_lshr, A_LSHR)
ONE_OP_KERNEL(k_add, A_ADD)
ONE_OP_KERNEL(k_mul, A_MUL)
ONE_OP_KERNEL(k_fma, A_FMA)
ONE_OP_KERNEL(k_fmac, A_FMAC)
ONE_OP_KERNEL(k_fma_den, A_FMA_DEN)
ONE_OP_KERNEL(k_fma_clamp, A_FMA_CLAMP)
ONE_OP_KERNEL(k_cvt_sdwa, A_CVT_SDWA)
ONE_OP_KERNEL(k_mov_sdwa, A_MOV_SDWA)
ONE_OP_KERNEL(k_max, A_MAX)
ONE_OP_KERNEL(k_max3, A_MAX3)
ONE_OP_KERNEL(k_align, A_ALIGN)
ONE_OP_KERNEL(k_cndmask, A_CNDMASK)
ONE_OP_KERNEL(k_cmp, A_CMP)
ONE_OP_KERNEL(k_xor, A_XOR)
ONE_OP_KERNEL(k_sub, A_SUB)
ONE_OP_KERNEL(k_med3, A_MED3)
ONE_OP_KERNEL(k_maxu, A_MAXU)
ONE_OP_KERNEL(k_mini16, A_MINI16)
ONE_OP_KERNEL(k_pkadd16, A_PKADD16)
ONE_OP_KERNEL(k_pkfmaf16, A_PKFMAF16)
ONE_OP_KERNEL(k_pkmulf16, A_PKMULF16)
ONE_OP_KERNEL(k_pkmaxf16, A_PKMAXF16)
ONE_OP_KERNEL(k_madmix, A_MADMIX)
ONE_OP_KERNEL(k_lshl_add, A_LSHL_ADD)
ONE_OP_KERNEL(k_add3, A_ADD3)
ONE_OP_KERNEL(k_or3, A_OR3)
ONE_OP_KERNEL(k_bfi, A_BFI)

// exactness of the denormal route: fma(asfloat(q), a * 2^85, b * 2^-64) against fma((float)q, a, b) * 2^-64
__global__ void k_exact(unsigned* bad, float a, float b) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 65535
    const float ref = __builtin_fmaf((float)q, a, b) * 0x1p-64f;
    const float got = __builtin_fmaf(__uint_as_float(q), a * 0x1p85f, b * 0x1p-64f);
    if (ref != got) atomicAdd(bad, 1u);
}

typedef void (*kern)(float*, unsigned long long*, int, float, float);
static void run(const char* name, kern k, float* d_out, unsigned long long* d_ticks, int cus, float a, float b) {
    const int trips = 2048, wps = 4;
    const int blocks = cus * wps;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, 16, a, b);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, trips, a, b);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * blocks * 4);
    CHECK(hipMemcpy(h, d_ticks, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < blocks * 4; ++i) sum += (double)h[i];
    free(h);
    const double ticks = sum / (blocks * 4);
    const double insts = (double)trips * 64.0;
    printf("%-34s %7.3f ms   wall x 2.4 GHz: %5.2f   counter ticks per instruction per SIMD: %6.3f\n", name, ms,
           ms * 1e-3 * 2.4e9 / (insts * wps), ticks / (insts * wps));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    int wall_khz = 0;
    CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    printf("%s  CUs %d  clockRate %.2f GHz  counter rate %.1f MHz (ticks are this counter's; x 2400/rate = shader cycles at 2.4 GHz)\n",
           p.gcnArchName, cus, p.clockRate * 1e-6, wall_khz * 1e-3);
    float* d_out;
    unsigned long long* d_ticks;
    CHECK(hipMalloc(&d_out, (size_t)cus * 4 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_ticks, (size_t)cus * 4 * 4 * sizeof(unsigned long long)));
    float den;
    const unsigned qbits = 40000u;
    memcpy(&den, &qbits, 4);
    run("v_mov_b32", k_mov, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_and_b32", k_and, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_xor_b32", k_xor, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_lshrrev_b32", k_lshr, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_add_f32", k_add, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_sub_f32", k_sub, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_mul_f32", k_mul, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_fma_f32", k_fma, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_fmac_f32", k_fmac, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_fma_f32 denormal multiplicand", k_fma_den, d_out, d_ticks, cus, 0x1p100f, den);
    run("v_fma_f32 clamp", k_fma_clamp, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_cvt_f32_u32_sdwa", k_cvt_sdwa, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_mov_b32_sdwa", k_mov_sdwa, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_max_f32", k_max, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_max3_f32", k_max3, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_med3_f32", k_med3, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_max_u32", k_maxu, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_alignbit_b32", k_align, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_cndmask_b32", k_cndmask, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_cmp_lt_f32", k_cmp, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_pk_min_i16", k_mini16, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_pk_add_u16", k_pkadd16, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_pk_fma_f16", k_pkfmaf16, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_pk_mul_f16", k_pkmulf16, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_pk_max_f16", k_pkmaxf16, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_fma_mix_f32 (f16 multiplicand)", k_madmix, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_lshl_add_u32", k_lshl_add, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_add3_u32", k_add3, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_or3_b32", k_or3, d_out, d_ticks, cus, 1.0001f, 0.5f);
    run("v_bfi_b32", k_bfi, d_out, d_ticks, cus, 1.0001f, 0.5f);

    unsigned* d_bad;
    CHECK(hipMalloc(&d_bad, 4));
    const float as[4] = {0.37f, -1234.5f, 3.1e-4f, -7.7e3f}, bs[4] = {-5.25f, 9.9e4f, 0.125f, -3.3e-2f};
    for (int i = 0; i < 4; ++i) {
        CHECK(hipMemset(d_bad, 0, 4));
        hipLaunchKernelGGL(k_exact, dim3(256), dim3(256), 0, 0, d_bad, as[i], bs[i]);
        unsigned bad = 0;
        CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        printf("denormal route, a = %g, b = %g: %u of 65536 plane indices differ from convert + fma\n", as[i], bs[i], bad);
    }
    return 0;
}
