// sqrt_check.hip -- exhaustive check of short correctly-rounded square roots: for EVERY positive normal float x
// each candidate sequence is compared bit for bit with the IEEE sqrtf(x).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/sqrt_check.hip -o tools/sqrt_check && tools/sqrt_check
#include <hip/hip_runtime.h>

#include <cstdint>
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

template <int V>
__device__ __forceinline__ float sqrt_short(float x) {
    if (V == 0) {  // rsq, one Newton step on s = x * y
        const float y = __builtin_amdgcn_rsqf(x);
        const float s = x * y;
        const float e = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(e, 0.5f * y, s);
    } else if (V == 1) {  // hardware sqrt + rsq for the correction factor
        const float s = __builtin_amdgcn_sqrtf(x);
        const float y = __builtin_amdgcn_rsqf(x);
        const float e = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(e, 0.5f * y, s);
    } else {  // two steps
        const float y = __builtin_amdgcn_rsqf(x);
        float s = x * y;
        const float h = 0.5f * y;
        float e = __builtin_fmaf(-s, s, x);
        s = __builtin_fmaf(e, h, s);
        e = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(e, h, s);
    }
}

template <int V>
__global__ void k_check(unsigned long long* bad, unsigned int* first_bad) {
    const uint32_t bits = blockIdx.x * blockDim.x + threadIdx.x;  // 2^31 positive patterns
    const float x = __uint_as_float(bits);
    const uint32_t ex = bits >> 23;
    if (ex == 0u || ex == 255u) return;  // zero, denormals, inf, NaN
    const float want = sqrtf(x);
    const float got = sqrt_short<V>(x);
    if (__float_as_uint(want) != __float_as_uint(got)) {
        atomicAdd(&bad[ex], 1ull);
        atomicMin(&first_bad[ex], bits);
    }
}

template <int V>
static void run(const char* name) {
    unsigned long long* d_bad;
    unsigned int* d_first;
    CHECK(hipMalloc(&d_bad, 256 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_first, 256 * sizeof(unsigned int)));
    CHECK(hipMemset(d_bad, 0, 256 * sizeof(unsigned long long)));
    CHECK(hipMemset(d_first, 0xFF, 256 * sizeof(unsigned int)));
    hipLaunchKernelGGL(k_check<V>, dim3(1u << 23), dim3(256), 0, 0, d_bad, d_first);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad[256];
    unsigned int first[256];
    CHECK(hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(first, d_first, sizeof first, hipMemcpyDeviceToHost));
    unsigned long long total = 0;
    int shown = 0;
    for (int e = 1; e < 255; ++e) {
        total += bad[e];
        if (bad[e] != 0 && shown < 6) {
            printf("  %s: biased exponent %3d: %llu mismatches, first at bits 0x%08x\n", name, e, bad[e], first[e]);
            shown++;
        }
    }
    int best_lo = 0, best_hi = -1, run_lo = -1;
    for (int e = 1; e <= 255; ++e) {
        const bool clean = e < 255 && bad[e] == 0;
        if (clean && run_lo < 0) run_lo = e;
        if (!clean && run_lo >= 0) {
            if (e - 1 - run_lo > best_hi - best_lo) { best_lo = run_lo; best_hi = e - 1; }
            run_lo = -1;
        }
    }
    printf("%s: %llu mismatches over all positive normal floats; clean for x in [2^%d, 2^%d)\n", name, total, best_lo - 127, best_hi - 126);
    CHECK(hipFree(d_bad));
    CHECK(hipFree(d_first));
}

int main() {
    run<0>("rsq + one step");
    run<1>("sqrt + rsq + one step");
    run<2>("rsq + two steps");
    return 0;
}
