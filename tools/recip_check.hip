// recip_check.hip -- exhaustive check of the short correctly-rounded reciprocal used by the triangle test:
// for EVERY positive float x the sequence  r = v_rcp_f32(x); one Newton step in fused arithmetic
// is compared bit for bit with the IEEE division 1.0f / x.  Prints the number of mismatches per binade range.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/recip_check.hip -o tools/recip_check && tools/recip_check
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

__device__ __forceinline__ float recip_short(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}

// bad[k]: mismatches among the floats with biased exponent k
__global__ void k_check(unsigned long long* bad, unsigned int* first_bad) {
    const uint32_t bits = blockIdx.x * blockDim.x + threadIdx.x;  // 2^31 positive patterns
    const float x = __uint_as_float(bits);
    const uint32_t ex = bits >> 23;
    if (ex == 0u || ex == 255u) return;  // zero, denormals, inf, NaN
    const float want = 1.0f / x;
    const float got = recip_short(x);
    if (__float_as_uint(want) != __float_as_uint(got)) {
        atomicAdd(&bad[ex], 1ull);
        atomicMin(&first_bad[ex], bits);
    }
}

int main() {
    unsigned long long* d_bad;
    unsigned int* d_first;
    CHECK(hipMalloc(&d_bad, 256 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_first, 256 * sizeof(unsigned int)));
    CHECK(hipMemset(d_bad, 0, 256 * sizeof(unsigned long long)));
    CHECK(hipMemset(d_first, 0xFF, 256 * sizeof(unsigned int)));
    hipLaunchKernelGGL(k_check, dim3(1u << 23), dim3(256), 0, 0, d_bad, d_first);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad[256];
    unsigned int first[256];
    CHECK(hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(first, d_first, sizeof first, hipMemcpyDeviceToHost));
    unsigned long long total = 0;
    int lo = -1, hi = -1;
    for (int e = 1; e < 255; ++e) {
        total += bad[e];
        if (bad[e] == 0) {
            if (lo < 0) lo = e;
            hi = e;
        } else {
            printf("biased exponent %3d (x in [2^%d, 2^%d)): %llu mismatches, first at bits 0x%08x\n", e, e - 127, e - 126, bad[e], first[e]);
        }
    }
    printf("total mismatches over all normal positive floats: %llu\n", total);
    // widest run of clean binades
    int best_lo = 0, best_hi = -1, run_lo = -1;
    for (int e = 1; e <= 255; ++e) {
        const bool clean = e < 255 && bad[e] == 0;
        if (clean && run_lo < 0) run_lo = e;
        if (!clean && run_lo >= 0) {
            if (e - 1 - run_lo > best_hi - best_lo) { best_lo = run_lo; best_hi = e - 1; }
            run_lo = -1;
        }
    }
    printf("clean for x in [2^%d, 2^%d)\n", best_lo - 127, best_hi - 126);
    return 0;
}
