// salu_mix.hip -- how many scalar instructions can ride along with the vector instructions of a SIMD for free?
// Every loop trip issues 32 v_fma_f32 (3.06 cycles each when alone) and M scalar adds between them, M = 0 .. 64,
// on all four SIMDs of every CU at 4 and 6 waves per SIMD.  Also the cost of a compare + ballot + scalar branch
// chain of the kind the traversal loop uses.
//   hipcc --offload-arch=gfx950 -O2 tools/salu_mix.hip -o tools/salu_mix && tools/salu_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

#define V4 "v_fma_f32 %[r0], %[r0], %[a], %[b]\n v_fma_f32 %[r1], %[r1], %[a], %[b]\n v_fma_f32 %[r2], %[r2], %[a], %[b]\n v_fma_f32 %[r3], %[r3], %[a], %[b]\n"
#define S1 "s_add_u32 %[s], %[s], 1\n"
#define S2 S1 S1
#define S4 S2 S2
#define S8 S4 S4

// M scalar adds per 4 vector instructions: MODE 0: 0, 1: 1, 2: 2, 3: 4, 4: 8
template <int MODE>
__global__ __launch_bounds__(256) void k_mix(float* out, int trips, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
    unsigned s = 0;
    for (int i = 0; i < trips; ++i) {
#define GROUP(SS) asm volatile(V4 SS : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [s] "+s"(s) : [a] "v"(a), [b] "v"(b) : "scc");
#define REP8G(SS) GROUP(SS) GROUP(SS) GROUP(SS) GROUP(SS) GROUP(SS) GROUP(SS) GROUP(SS) GROUP(SS)
        if (MODE == 0) { REP8G("") }
        else if (MODE == 1) { REP8G(S1) }
        else if (MODE == 2) { REP8G(S2) }
        else if (MODE == 3) { REP8G(S4) }
        else { REP8G(S8) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + (float)s;
}

// 4 vector instructions, then a data-dependent compare, saveexec and a branch that is never taken for the whole
// wave (the shape of "if (cond) { rare }" in the traversal loop).  MODE = number of such chains per 4 fma.
template <int MODE>
__global__ __launch_bounds__(256) void k_branchy(float* out, int trips, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
    for (int i = 0; i < trips; ++i) {
#define CHAIN "v_cmp_gt_f32 vcc, 0, %[r0]\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_mov_b32 %[r1], 0\n 1:\n s_or_b64 exec, exec, s[20:21]\n"
#define BG(CC) asm volatile(V4 CC : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3) : [a] "v"(a), [b] "v"(b) : "vcc", "scc", "s20", "s21");
#define REP8B(CC) BG(CC) BG(CC) BG(CC) BG(CC) BG(CC) BG(CC) BG(CC) BG(CC)
        if (MODE == 0) { REP8B("") }
        else if (MODE == 1) { REP8B(CHAIN) }
        else { REP8B(CHAIN CHAIN) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
}

// The render kernel's own mix, per group: 3 v_fma_f32, 3 of the convert / min / max class, 2 plain adds, 3 scalar
// logic instructions (about its 72 % vector share), no memory.
__global__ __launch_bounds__(256) void k_kernel_mix(float* out, int trips, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
    for (int i = 0; i < trips; ++i) {
#define KM asm volatile("v_fma_f32 %[r0], %[r0], %[a], %[b]\n s_and_b64 s[20:21], s[22:23], exec\n v_min_f32 %[r1], %[r1], %[a]\n v_fma_f32 %[r2], %[r2], %[a], %[b]\n" \
                        "v_add_f32 %[r3], %[r3], %[a]\n s_or_b64 s[22:23], s[20:21], exec\n v_max3_f32 %[r0], %[r0], %[a], %[b]\n v_fma_f32 %[r1], %[r1], %[a], %[b]\n" \
                        "v_cvt_f32_u32 %[r2], %[r2]\n s_xor_b64 s[20:21], s[22:23], exec\n v_add_u32 %[r3], %[r3], %[r0]\n" \
                        : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3) : [a] "v"(a), [b] "v"(b) : "scc", "s20", "s21", "s22", "s23");
        KM KM KM KM KM KM KM KM
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
}

typedef void (*kern)(float*, int, float, float);
static void run_total(const char* name, kern k, float* d_out, int cus, double ghz, int insts_per_trip) {
    const int trips = 4096;
    for (int wps = 4; wps <= 8; ++wps) {
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
        printf("%-44s waves/SIMD %d  %.3f ms  %.2f cycles per instruction (vector or scalar) per SIMD\n", name, wps, ms,
               ms * 1e-3 * ghz * 1e9 / ((double)trips * insts_per_trip * wps));
        fflush(stdout);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}
static void run(const char* name, kern k, float* d_out, int cus, double ghz, int valu_per_trip) {
    const int trips = 4096;
    for (int wps = 4; wps <= 6; wps += 2) {
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
        printf("%-44s waves/SIMD %d  %.3f ms  %.2f cycles per v_fma_f32 per SIMD\n", name, wps, ms,
               ms * 1e-3 * ghz * 1e9 / ((double)trips * valu_per_trip * wps));
        fflush(stdout);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    const int cus = p.multiProcessorCount;
    float* d_out;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float)));
    run("4 fma + 0 s_add", k_mix<0>, d_out, cus, ghz, 32);
    run("4 fma + 1 s_add", k_mix<1>, d_out, cus, ghz, 32);
    run("4 fma + 2 s_add", k_mix<2>, d_out, cus, ghz, 32);
    run("4 fma + 4 s_add", k_mix<3>, d_out, cus, ghz, 32);
    run("4 fma + 8 s_add", k_mix<4>, d_out, cus, ghz, 32);
    run("4 fma, no branch chain", k_branchy<0>, d_out, cus, ghz, 32);
    run("4 fma + 1 (cmp, saveexec, branch, or exec)", k_branchy<1>, d_out, cus, ghz, 32);
    run("4 fma + 2 (cmp, saveexec, branch, or exec)", k_branchy<2>, d_out, cus, ghz, 32);
    run_total("kernel-like mix (8 vector + 3 scalar)", k_kernel_mix, d_out, cus, ghz, 88);
    CHECK(hipFree(d_out));
    return 0;
}
