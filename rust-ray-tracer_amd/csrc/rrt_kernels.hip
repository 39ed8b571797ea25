// rrt_kernels.hip -- gfx950 kernels: ray-batch closest hit, tone map, chunk sum, and the launchers of all kernels;
// instantiates the render kernel (rrt_render_kernel.h) for scenes without instances and for the two-level walk.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include "rrt_device.h"
#include "rrt_kernels.h"
#include "rrt_render_kernel.h"

namespace rrt {

// The caller-visible record of a closest hit.  Sphere uv follows the reference's get_sphere_uv
// (src/geometry/sphere.rs:34-40), including its far-root call with the un-normalised p.
__device__ __forceinline__ rrt_hit hit_record(const SceneDev& S, const Best& best, V3 o, V3 d) {
    rrt_hit h;
    if (best.what == WHAT_NONE) {
        h.prim_id = -1; h.t = __builtin_inff(); h.u = 0.f; h.v = 0.f;
    } else {
        h.prim_id = (int32_t)prim_id_of(S, best.what, best.inst); h.t = best.t; h.u = best.u; h.v = best.v;
        if ((best.what >> 28) == KIND_SPHERE) {
            const uint32_t idx = best.what & REF_OFFSET_MASK;
            V3 ro = o, rd = d;
            if (best.inst >= 0) to_object_space(S.insts[best.inst], o, d, ro, rd);
            const float4 q0 = *reinterpret_cast<const float4*>(&S.spheres[idx].c[0]);
            const V3 c = v3(q0.x, q0.y, q0.z);
            const V3 p = ro + best.t * rd;
            // which root was it?  recompute the near root exactly as sphere_root does
            const V3 oc = ro - c;
            const float a = length_squared(rd), b = dot(oc, rd), cc = length_squared(oc) - q0.w * q0.w;
            const float sq = sqrtf(b * b - a * cc);
            const float near_root = (-b - sq) / a;
            const V3 q = (near_root == best.t) ? (p - c) / q0.w : p;
            const float theta = acosf(-q.y);
            const float phi = atan2f(-q.z, q.x) + RRT_PI;
            h.u = phi / (2.0f * RRT_PI);
            h.v = theta / RRT_PI;
        }
    }
    return h;
}

// Closest hit for a batch of rays, one lane per ray from start to end (kept as the plain form of the walk; the
// streaming kernel below is the one the library launches).
template <int MODE>
__global__ __launch_bounds__(WAVE) void k_intersect(SceneDev S, const rrt_ray* __restrict__ rays, size_t n, float tmin,
                                                    float tmax, rrt_hit* __restrict__ out) {
    constexpr bool INST = MODE == MODE_TWO_LEVEL;
    __shared__ int s_stack[STACK_DEPTH * WAVE];
    __shared__ float s_wray[INST ? 6 * WAVE : 1];
    const size_t i = (size_t)blockIdx.x * WAVE + threadIdx.x;
    if (i >= n) return;
    const rrt_ray r = rays[i];
    const V3 o = v3(r.origin[0], r.origin[1], r.origin[2]);
    const V3 d = v3(r.direction[0], r.direction[1], r.direction[2]);
    Best best;
    TraceCounters cnt = {0u, 0u, 0u, 0u, 0u};
    trace<MODE, false>(S, o, d, tmin, tmax, (lds_int*)(s_stack + threadIdx.x), WAVE, best, cnt,
                       INST ? (lds_float*)(s_wray + threadIdx.x) : nullptr);
    out[i] = hit_record(S, best, o, d);
}

// Closest hit for a batch of rays, streaming form: a persistent grid in which every lane takes the next ray of
// the batch as soon as its own walk is complete, so that the lanes of a wavefront stay busy although their rays
// need different numbers of steps.  Rays are handed out through one counter (`head`, zero before the launch) in
// blocks of TRACE_BLOCK indices per wavefront (one atomic add per block: a counter shared by all eight XCDs takes
// about 10 ns per add, which would otherwise bound the kernel); whenever at least `refill_min` lanes of a
// wavefront are idle, they take the next indices of the wavefront's block.  counters (COUNT only): {-, node visits, primitive tests, node passes, primitive
// passes, -, refill passes}.
// The streaming kernel keeps the bottom RRT_STREAM_K stack entries of a lane in LDS and the rest in private memory, like
// the render kernel: with the whole 39-entry stack in LDS a CU held four workgroups, and the walk is a chain of
// dependent loads that four wavefronts per SIMD do not cover (tile-ordered primary rays of the 1M-triangle scene:
// 10.3 Grays/s at four wavefronts, 12.6 at seven; 134M-ray batches, where the tail no longer shows: 14.2).
#ifndef RRT_STREAM_STEPS
#define RRT_STREAM_STEPS 5  // node steps between two continuation votes of a node pass, streaming kernel
#endif
#ifndef RRT_STREAM_K
#define RRT_STREAM_K 16
#endif
#ifndef RRT_STREAM_WAVES
#define RRT_STREAM_WAVES 7
#endif
#ifndef RRT_STREAM_WAVES_INST
#define RRT_STREAM_WAVES_INST 6
#endif
template <int MODE, bool COUNT>
__global__ __launch_bounds__(RENDER_BLOCK, MODE == MODE_TWO_LEVEL ? RRT_STREAM_WAVES_INST : RRT_STREAM_WAVES) void k_trace_stream(SceneDev S, const rrt_ray* __restrict__ rays, uint32_t n,
                                                                  float tmin, float tmax, rrt_hit* __restrict__ out,
                                                                  unsigned int* __restrict__ head, uint32_t refill_min,
                                                                  uint32_t node_share_64ths,
                                                                  unsigned long long* __restrict__ counters) {
    constexpr bool INST = MODE == MODE_TWO_LEVEL;
    constexpr int K = RRT_STREAM_K;
    __shared__ int s_stack[(K > 0 ? K : STACK_DEPTH) * RENDER_BLOCK];
    __shared__ float s_wray[INST ? 6 * RENDER_BLOCK : 1];
    lds_int* stack = (lds_int*)(s_stack + threadIdx.x);
    lds_int* lim = (lds_int*)s_stack + K * RENDER_BLOCK;
    int deep[K > 0 ? STACK_DEPTH - K : 1];
    lds_float* wray = INST ? (lds_float*)(s_wray + threadIdx.x) : nullptr;
    // unrolled scenes: the instance transforms in LDS, as in the render kernel
    __shared__ vec4f s_xf[MODE == MODE_UNROLLED ? 2 * XF_LDS_SLOTS : 1];
    const lds_float4* xf_lds = nullptr;
    if (MODE == MODE_UNROLLED && S.inst_count + 1u <= (uint32_t)XF_LDS_SLOTS) {
        const vec4f* src = reinterpret_cast<const vec4f*>(S.insts - 1);
        for (uint32_t i = threadIdx.x; i < 2u * (S.inst_count + 1u); i += RENDER_BLOCK) s_xf[i] = src[(i >> 1) * 4u + (i & 1u)];
        __syncthreads();
        xf_lds = (const lds_float4*)s_xf;
    }
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t mine = 0xFFFFFFFFu;  // index of the ray this lane is walking (or has walked and not yet written)
    V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f);
    Trav T;
    T.cur = REF_DONE;
    T.top = stack;
    TraceCounters cnt = {0u, 0u, 0u, 0u, 0u};
    bool exhausted = false;  // wave-uniform: the batch has no rays left to hand out
    uint32_t blk = 0u, blk_end = 0u;  // wave-uniform: the block of ray indices this wavefront hands to its lanes
    for (;;) {
        const bool idle = T.cur == REF_DONE;
        const unsigned long long idle_m = __ballot(idle);
        const uint32_t n_idle = (uint32_t)__popcll(idle_m);
        if ((!exhausted && n_idle >= refill_min) || n_idle == 64u) {
            if (COUNT && lane == 0u) cnt.wave_shade++;
            if (idle && mine != 0xFFFFFFFFu) {
                out[mine] = hit_record(S, T.best, o, d);
                mine = 0xFFFFFFFFu;
            }
            if (exhausted) break;  // every lane is idle and nothing is left
            if (blk == blk_end) {  // the wavefront's private block is used up: draw the next one
                unsigned int b = 0;
                if (lane == 0u) b = atomicAdd(head, TRACE_BLOCK);
                b = __shfl(b, 0, WAVE);
                blk = b < n ? b : n;
                blk_end = (unsigned long long)b + TRACE_BLOCK < n ? b + TRACE_BLOCK : n;
                if (blk == blk_end) {
                    exhausted = true;
                    continue;
                }
            }
            const uint32_t take = min(n_idle, blk_end - blk);
            const uint32_t rank = (uint32_t)__popcll(idle_m & ((1ull << lane) - 1ull));
            if (idle && rank < take) {
                mine = blk + rank;
                const rrt_ray r = rays[mine];
                o = v3(r.origin[0], r.origin[1], r.origin[2]);
                d = v3(r.direction[0], r.direction[1], r.direction[2]);
                trav_begin<MODE>(S, o, d, tmax, stack, RENDER_BLOCK, T);
            }
            blk += take;
            continue;
        }
        const uint32_t n_walk = 64u - n_idle;
        const uint32_t n_desc = (uint32_t)__popcll(__ballot(T.cur >= 0 || (INST && trav_due(T.cur))));
        const uint32_t share = (n_walk * node_share_64ths) >> 6;
        if (n_desc >= share) trav_nodes<MODE, COUNT, K, RRT_STREAM_STEPS>(S, o, d, tmin, RENDER_BLOCK, T, cnt, share, lim, deep, wray);
        if (T.cur < 0 && T.cur != REF_DONE && !(INST && trav_due(T.cur))) trav_leaf<MODE, COUNT, K, true>(S, o, d, tmin, RENDER_BLOCK, T, cnt, lim, deep, wray, xf_lds);
    }
    if (COUNT) {
        const unsigned long long nn = wave_sum((unsigned long long)cnt.nodes);
        const unsigned long long np = wave_sum((unsigned long long)cnt.prims);
        const unsigned long long wn = wave_sum((unsigned long long)cnt.wave_nodes);
        const unsigned long long wp = wave_sum((unsigned long long)cnt.wave_prims);
        const unsigned long long wr = wave_sum((unsigned long long)cnt.wave_shade);
        if (lane == 0u) {
            atomicAdd(&counters[1], nn); atomicAdd(&counters[2], np);
            atomicAdd(&counters[3], wn); atomicAdd(&counters[4], wp); atomicAdd(&counters[6], wr);
        }
    }
}

// get_color + flipped put_pixel -- reference src/vec3.rs:30-53, src/main.rs:104-109
__global__ void k_tonemap(const float* __restrict__ accum, uint32_t width, uint32_t height, uint32_t spp,
                          uint8_t* __restrict__ rgb8) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t x = i % width, y = i / width;
    const float scale = 1.0f / (float)spp;
    uint8_t* px = rgb8 + 3ull * ((size_t)(height - 1u - y) * width + x);
    for (int c = 0; c < 3; ++c) {
        const float v = accum[3ull * i + c];
        float g = (v != v) ? 0.0f : sqrtf(v * scale);
        // f32::clamp keeps NaN; `as u8` saturates and maps NaN to 0
        float cl = g;
        if (cl < 0.0f) cl = 0.0f;
        if (cl > 0.999f) cl = 0.999f;
        const float q = 256.0f * cl;
        px[c] = (q != q) ? (uint8_t)0 : (uint8_t)(int)q;
    }
}

// Self-test of the short correctly rounded reciprocal and square root: every float bit pattern is run through
// recip_rn / sqrt_rn and through the IEEE operations; bad[0] / bad[1] count the patterns whose results differ in
// any bit (NaN results are compared as NaN == NaN).
__global__ void k_check_arith(unsigned long long* __restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long n_recip = 0ull, n_sqrt = 0ull;
    // 2^20 workgroups x 256 threads x 16 patterns
    for (uint32_t k = 0; k < 16u; ++k) {
        const uint32_t bits = ((blockIdx.x * 16u + k) << 8) | threadIdx.x;
        const float x = __uint_as_float(bits);
        const float r0 = 1.0f / x, r1 = recip_rn(x);
        const float s0 = sqrtf(x), s1 = sqrt_rn(x);
        if (__float_as_uint(r0) != __float_as_uint(r1) && !(r0 != r0 && r1 != r1)) n_recip++;
        if (__float_as_uint(s0) != __float_as_uint(s1) && !(s0 != s0 && s1 != s1)) n_sqrt++;
    }
    n_recip = wave_sum(n_recip);
    n_sqrt = wave_sum(n_sqrt);
    if (lane == 0u && (n_recip | n_sqrt) != 0ull) {
        atomicAdd(&bad[0], n_recip);
        atomicAdd(&bad[1], n_sqrt);
    }
}

// Sums the per-chunk partial images in chunk order (deterministic, unlike float atomics).
__global__ void k_reduce_chunks(const float* __restrict__ partial, uint32_t n_floats, uint32_t n_chunks,
                                float* __restrict__ accum) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    float s = 0.0f;
    for (uint32_t c = 0; c < n_chunks; ++c) s += partial[(size_t)c * n_floats + i];
    accum[i] = s;
}

// dst[i] += src[i]: the sum step of a multi-GPU group whose members' accumulation buffers are gathered on one GPU.
__global__ void k_add(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n4 = n / 4u;
    if (i < n4) {
        float4 a = reinterpret_cast<float4*>(dst)[i];
        const float4 b = reinterpret_cast<const float4*>(src)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        reinterpret_cast<float4*>(dst)[i] = a;
    } else if (i == n4) {
        for (size_t k = n4 * 4u; k < n; ++k) dst[k] += src[k];
    }
}

// The candidate lists of every pixel's primary rays, one lane per pixel, written to `lists` (PIXEL_LIST_DWORDS dwords
// per pixel, row-major): per pixel the primitives a camera ray of the pixel can hit, as leaf references (a triangle as
// a triangle leaf of one) -- the spheres and rectangles of the leaves whose stored boxes the pixel's pyramid touches,
// and those of their triangles that frustum_touches_triangle keeps -- in ascending order of the lower bound of the ray
// parameter at which a ray of the pixel can reach their leaf's box; 0xFFFFFFFF in place of the count when there are
// more than max_len.
// The 64 pixels of a wavefront (an 8x8 block) walk the tree TOGETHER: one depth-first walk over the nodes that any of
// their pyramids touches, with one stack for the wavefront (a node reference and the mask of the lanes it concerns)
// and every node and triangle record fetched once, through the scalar cache; a lane only tests what concerns it.
// Neighbouring pyramids touch nearly the same nodes, so the shared walk is hardly longer than a single pixel's, and
// it has no divergent control flow.  Leaves are handled where their parent is visited (the parent holds their box),
// so the stack holds inner nodes only.
// listed (one u64): the kernel adds the number of pixels that got a list.
// (Six workgroups per CU, 80 registers; measured on MI355X at 4 / 5 / 6 / 7 / 8: 2.55 / 2.50 / 2.16 / 2.60 / 4.26 ms for
// the 1M-triangle scene at 3840x2160.  One lane per pixel walking by itself took 3.3 ms.)
// UNROLLED: a leaf's reference names its instance slot above the record index (SceneDev::unroll_shift); the records of
// an instance's triangles are in the mesh's space and are moved to world space (the point map of transform.rs:113-124
// and 27) before the pyramid's tests; entries keep the slot, so that the render kernel tests them as its walk would.
template <bool UNROLLED>
__global__ __launch_bounds__(256, 6) void k_pixel_lists(SceneDev S, uint32_t width, uint32_t height, uint32_t max_len, uint32_t* __restrict__ lists,
                                                        unsigned long long* __restrict__ listed) {
    __shared__ int s_ref[4][STACK_DEPTH];
    __shared__ unsigned long long s_mask[4][STACK_DEPTH];
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t wv = threadIdx.x >> 6, wave = blockIdx.x * 4u + wv, lane = threadIdx.x & 63u;
    const uint32_t x = (wave % tiles_x) * 8u + (lane & 7u), y = (wave / tiles_x) * 8u + (lane >> 3);
    const bool inside = x < width && y < height;
    const PixelFrustum F = pixel_frustum(S.cam, min(x, width - 1u), min(y, height - 1u), width, height, UNROLLED ? 3.0f : 1.0f);
    uint32_t refs[PIXEL_LIST_MAX];
    float entry[PIXEL_LIST_MAX];
    uint32_t n = inside && F.ok ? 0u : 0xFFFFFFFFu;
    const V3 g0 = v3(S.grid.origin[0], S.grid.origin[1], S.grid.origin[2]), gs = v3(S.grid.step[0], S.grid.step[1], S.grid.step[2]);
    // A leaf child of the node being visited, for the lanes in `mine`: its candidates go into their lists.
    auto take_leaf = [&](uint32_t leaf, bool mine, float leaf_t) {
        const uint32_t kind = (leaf >> 28) & 7u;
        if (kind == KIND_TRI) {
            const uint32_t count = ((leaf >> 25) & 7u) + 1u, first = leaf & REF_OFFSET_MASK;
            uint32_t rec0 = first;
            float sn = 0.0f, cs = 1.0f, slack = 1.0f;
            V3 off = v3(0.0f, 0.0f, 0.0f);
            if (UNROLLED) {
                rec0 = first & ((1u << S.unroll_shift) - 1u);
                const uint32_t slot = first >> S.unroll_shift;  // uniform, like the leaf
                if (slot != 0u) {
                    const const_u32x4* ir = (const const_u32x4*)(S.insts + (slot - 1u));
                    const u32x4 x0 = ir[0], x1 = ir[1];
                    off = v3(__uint_as_float(x0.x), __uint_as_float(x0.y), __uint_as_float(x0.z));
                    sn = __uint_as_float(x1.x); cs = __uint_as_float(x1.y);
                    slack = 2.0f;
                }
            }
            for (uint32_t j = 0; j < count; ++j) {
                const const_u32x4* rec = (const const_u32x4*)(reinterpret_cast<const char*>(S.tris) + (size_t)(rec0 + j) * 48u);
                const u32x4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
                V3 v0 = v3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
                V3 e1 = v3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z));
                V3 e2 = v3(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z));
                if (UNROLLED) {
                    v0 = v3(cs * v0.x + sn * v0.z, v0.y, -sn * v0.x + cs * v0.z) + off;
                    e1 = v3(cs * e1.x + sn * e1.z, e1.y, -sn * e1.x + cs * e1.z);
                    e2 = v3(cs * e2.x + sn * e2.z, e2.y, -sn * e2.x + cs * e2.z);
                }
                if (mine && n != 0xFFFFFFFFu && frustum_touches_triangle(F, v0, e1, e2, slack)) {
                    if (n == max_len) n = 0xFFFFFFFFu;  // one more than a list may hold: no list
                    else list_insert(refs, entry, n, (KIND_TRI << 28) | (first + j), leaf_t);
                }
            }
        } else if (kind <= KIND_RECT) {  // a sphere, a rectangle (an empty child slot is skipped)
            if (mine && n != 0xFFFFFFFFu) {
                if (n == max_len) n = 0xFFFFFFFFu;
                else list_insert(refs, entry, n, leaf, leaf_t);
            }
        }
    };
    int cur = S.root;  // uniform
    unsigned long long cur_mask = __ballot(n != 0xFFFFFFFFu);
    int sp = 0;
    if (cur < 0) {  // a scene of one leaf
        take_leaf((uint32_t)cur, n != 0xFFFFFFFFu, 0.0f);
        cur_mask = 0ull;
    }
    while (cur_mask != 0ull) {
        const const_u32x4* np = (const const_u32x4*)(reinterpret_cast<const char*>(S.nodes) + (uint32_t)cur);
        const u32x4 a = np[0], b = np[1];  // a = (left, right, L.x, L.y)  b = (L.z, R.x, R.y, R.z)
        const bool mine = ((cur_mask >> lane) & 1ull) != 0ull && n != 0xFFFFFFFFu;
        auto lo = [&](uint32_t wx, uint32_t wy, uint32_t wz) {
            return v3(__builtin_fmaf((float)(wx & 0xFFFFu), gs.x, g0.x), __builtin_fmaf((float)(wy & 0xFFFFu), gs.y, g0.y), __builtin_fmaf((float)(wz & 0xFFFFu), gs.z, g0.z));
        };
        auto hi = [&](uint32_t wx, uint32_t wy, uint32_t wz) {
            return v3(__builtin_fmaf((float)(wx >> 16), gs.x, g0.x), __builtin_fmaf((float)(wy >> 16), gs.y, g0.y), __builtin_fmaf((float)(wz >> 16), gs.z, g0.z));
        };
        const V3 lmn = lo(a.z, a.w, b.x), lmx = hi(a.z, a.w, b.x), rmn = lo(b.y, b.z, b.w), rmx = hi(b.y, b.z, b.w);
        const bool hl = mine && frustum_touches_box(F, lmn, lmx), hr = mine && frustum_touches_box(F, rmn, rmx);
        unsigned long long ml = __ballot(hl), mr = __ballot(hr);
        const int left = (int)a.x, right = (int)a.y;
        if (left < 0) {
            if (ml != 0ull) take_leaf((uint32_t)left, hl, hl ? frustum_box_entry(F, lmn, lmx) : 0.0f);
            ml = 0ull;
        }
        if (right < 0) {
            if (mr != 0ull) take_leaf((uint32_t)right, hr, hr ? frustum_box_entry(F, rmn, rmx) : 0.0f);
            mr = 0ull;
        }
        if (ml != 0ull && mr != 0ull) {
            if (sp == STACK_DEPTH) {  // (deeper than any tree the builders make: nobody gets a list)
                n = 0xFFFFFFFFu;
                break;
            }
            if (lane == 0u) { s_ref[wv][sp] = right; s_mask[wv][sp] = mr; }
            sp++;
            cur = left; cur_mask = ml;
        } else if (ml != 0ull) {
            cur = left; cur_mask = ml;
        } else if (mr != 0ull) {
            cur = right; cur_mask = mr;
        } else {
            // the next node on the stack that still concerns a lane with a list
            cur_mask = 0ull;
            const unsigned long long alive = __ballot(n != 0xFFFFFFFFu);
            while (sp > 0 && cur_mask == 0ull) {
                --sp;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                cur = __builtin_amdgcn_readfirstlane(s_ref[wv][sp]);
                const unsigned long long m = s_mask[wv][sp];
                cur_mask = (((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)m)) & alive;
            }
        }
    }
    const unsigned long long got = __ballot(inside && n != 0xFFFFFFFFu);
    if (lane == 0u && got != 0ull) atomicAdd(listed, (unsigned long long)__popcll(got));
    if (!inside) return;
    uint32_t* dst = lists + (size_t)(y * width + x) * PIXEL_LIST_DWORDS;
    dst[0] = n;
    dst[1] = 0u;
    for (uint32_t i = 0; i < PIXEL_LIST_MAX; ++i) {
        const bool have = n != 0xFFFFFFFFu && i < n;
        dst[2u + 2u * i] = have ? refs[i] : 0u;
        dst[3u + 2u * i] = have ? __float_as_uint(entry[i]) : 0u;
    }
}

// ---- launchers --------------------------------------------------------------------------------

uint32_t render_chunks(uint32_t width, uint32_t height, uint32_t spp, uint32_t samples_per_item) {
    // Measured on MI355X: items of about 16 trips (16 samples where most paths end at the first ray, 8 where they
    // bounce) amortise the queue ticket and the 12-byte-per-pixel write of an item and still spread an expensive
    // tile over many CUs; small frames get more, smaller items so that every resident wave finds a few
    // (target: 4 items per wavefront slot of the chip, 256 CUs x 20 wavefronts).  Never fewer than 4
    // samples per item, never more than 32 partial images.
    const uint64_t n_tiles = (uint64_t)((width + TILE_W - 1u) / TILE_W) * ((height + TILE_H - 1u) / TILE_H);
    if (samples_per_item == 0u) samples_per_item = 16u;
    uint64_t n = (spp + samples_per_item - 1u) / samples_per_item;
    const uint64_t for_balance = (4ull * 256ull * 20ull + n_tiles - 1u) / n_tiles;
    if (n < for_balance) n = for_balance;
    if (n > spp / 4u) n = spp / 4u;
    if (n > 32u) n = 32u;
    if (n < 1u) n = 1u;
    return (uint32_t)n;
}

bool render_takes_pixel_lists(const SceneDev& S, uint32_t width, uint32_t height, uint32_t n_chunks, const RenderTuning& tuning) {
    if (tuning.pixel_lists == 0u || tuning.split_items == 0u || width >= 65536u || height >= 65536u) return false;
    // (a scene whose instances are unrolled is walked as one world-space tree like a scene without instances; a
    // two-level scene's instance leaves hold whole meshes)
    if ((S.unroll_shift == 0u && S.has_instances != 0u) || n_chunks < 64u / (uint32_t)LIST_PIXELS) return false;
    return true;
}

hipError_t launch_render(const SceneDev& S, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                         float* d_accum, float* d_partial, uint32_t n_chunks, unsigned long long* d_counters,
                         unsigned int* d_heads, int cu_count, bool count_visits, const RenderTuning& tuning,
                         hipStream_t stream, uint32_t* d_lists) {
    // the pipelined form packs a lane's pixel coordinates into 16 bits each
    const bool pipelined = tuning.overlap_items != 0u && width < 65536u && height < 65536u;
    // Scenes without instances, where the caller asks for it: an item is a few pixels times as many of their sample chunks as make 64 lanes -- all chunks of the
    // frame when there are at most 32 (an item is then everything the frame asks of its pixels).  The lanes of an item
    // render the same few pixels, so they have the same work ahead of them (no item whose lanes are half sky and half
    // ground, with the sky lanes out of work long before the item retires), and their primary rays are nearly the
    // same ray.  Measured on MI355X against 64 pixels x one chunk: 1M-triangle scene at 1024 spp 1177 -> 1124 ms
    // (154.9 -> 151.9 ms at 128 spp), MetalRoughSpheres 358 -> 331 ms; ten-million-triangle scenes lose 1-3 % and keep
    // whole items (the caller's choice: RenderTuning::split_items).
    if (n_chunks == 0u || n_chunks > spp) return hipErrorInvalidValue;  // every chunk holds at least one sample
    uint32_t pix_log2 = 6u;
    const int mode = S.unroll_shift != 0u ? MODE_UNROLLED : (S.has_instances != 0u ? MODE_TWO_LEVEL : MODE_FLAT);
    // (unrolled scenes are only built with split items together with lists)
    const bool lists_wanted = d_lists != nullptr && render_takes_pixel_lists(S, width, height, n_chunks, tuning);
    if (tuning.split_items != 0u && (mode == MODE_FLAT || (mode == MODE_UNROLLED && lists_wanted))) {
#ifdef RRT_PIPE_PIX_LOG2
        pix_log2 = RRT_PIPE_PIX_LOG2;
#else
        uint32_t sub_log2 = 0u;
        while (sub_log2 < 5u && (1u << sub_log2) < n_chunks) ++sub_log2;  // chunks per item: the power of two that holds them all, at most 32
        pix_log2 = 6u - sub_log2;
#endif
        // split items are numbered with 31 bits like the others: whole tiles otherwise
        const uint32_t w_log2 = pix_log2 / 2u, h_log2 = pix_log2 - w_log2;
        const uint64_t items = (uint64_t)((width + (1u << w_log2) - 1u) >> w_log2) * ((height + (1u << h_log2) - 1u) >> h_log2) *
                               ((n_chunks + (64u >> pix_log2) - 1u) >> (6u - pix_log2));
        if (items > 0x7FFFFFFFull) pix_log2 = 6u;
        if (mode == MODE_UNROLLED && (1u << pix_log2) > (uint32_t)LIST_PIXELS) pix_log2 = 6u;
    }
    const uint32_t tw_log2 = pix_log2 == 6u ? (uint32_t)RRT_TILE_W_LOG2 : pix_log2 / 2u;
    const uint32_t tw = 1u << tw_log2, th = (1u << pix_log2) / tw, sub = 64u >> pix_log2;
    RenderGeom G;
    G.width = width; G.height = height; G.spp = spp; G.spp_offset = spp_offset;
    G.tiles_x = (width + tw - 1u) / tw;
    G.n_tiles = G.tiles_x * ((height + th - 1u) / th);
    // (band widths are given in tiles of 8 pixels' width)
    const uint32_t band = tuning.band_tiles * (8u / std::min(tw, 8u));
    G.band_tiles = tuning.band_tiles == 0u ? G.tiles_x : std::min(band, G.tiles_x);
    G.band_items = G.band_tiles * ((height + th - 1u) / th);
    G.full_bands = G.tiles_x / G.band_tiles;
    G.n_chunks = n_chunks;
    G.chunk_q = spp / n_chunks;
    G.chunk_r = spp % n_chunks;
    G.n_groups = (n_chunks + sub - 1u) / sub;
    G.pix_log2 = pix_log2;
    G.tw_log2 = tw_log2;
    G.lists = nullptr;
    G.n_work = G.n_tiles * G.n_groups;
    G.node_share_64ths = tuning.node_share_64ths;
    G.leave_walk = tuning.leave_walk;
    G.leave_ready = tuning.leave_ready;
    float* out = n_chunks > 1u ? d_partial : d_accum;
    hipError_t e;
    // the scaled slab arithmetic where the host found it valid for this scene (S.sigma != 0) and the walk stays on one grid
    const bool scaled = S.sigma != 0.0f && tuning.scaled_slab != 0u && mode != MODE_TWO_LEVEL;
    const unsigned variant = (scaled ? 16u : 0u) | ((unsigned)mode << 2) | (count_visits ? 2u : 0u) | (pipelined ? 1u : 0u);
    const bool lists = pix_log2 != 6u && lists_wanted && (1u << pix_log2) <= (uint32_t)LIST_PIXELS;
    if (lists) {
        // split items of a few pixels: primary rays resolved against per-pixel candidate lists, which a pass over the
        // frame's pixels collects first
        const uint32_t waves = ((width + 7u) / 8u) * ((height + 7u) / 8u);
        const uint32_t max_len = std::min(tuning.pixel_list_len, PIXEL_LIST_MAX);
        if (mode == MODE_UNROLLED)
            hipLaunchKernelGGL(k_pixel_lists<true>, dim3((waves + 3u) / 4u), dim3(256), 0, stream, S, width, height, max_len, d_lists, d_counters + 14);
        else
            hipLaunchKernelGGL(k_pixel_lists<false>, dim3((waves + 3u) / 4u), dim3(256), 0, stream, S, width, height, max_len, d_lists, d_counters + 14);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        G.lists = d_lists;
    }
    if (mode == MODE_UNROLLED) {
        e = lists ? launch_render_unrolled_lists(count_visits, pipelined, scaled, S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream)
                  : launch_render_unrolled(count_visits, pipelined, scaled, S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream);
    } else if (lists) {
        const bool roomy = pipelined && tuning.list_waves == 6u;
        switch ((roomy ? 8u : 0u) | (scaled ? 4u : 0u) | (count_visits ? 2u : 0u) | (pipelined ? 1u : 0u)) {
            case 15u: e = launch_render_variant<MODE_FLAT, true, true, true, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 13u: e = launch_render_variant<MODE_FLAT, false, true, true, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 11u: e = launch_render_variant<MODE_FLAT, true, true, false, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 9u: e = launch_render_variant<MODE_FLAT, false, true, false, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 7u: e = launch_render_variant<MODE_FLAT, true, true, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 6u: e = launch_render_variant<MODE_FLAT, true, false, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 5u: e = launch_render_variant<MODE_FLAT, false, true, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 4u: e = launch_render_variant<MODE_FLAT, false, false, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 3u: e = launch_render_variant<MODE_FLAT, true, true, false, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 2u: e = launch_render_variant<MODE_FLAT, true, false, false, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 1u: e = launch_render_variant<MODE_FLAT, false, true, false, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            default: e = launch_render_variant<MODE_FLAT, false, false, false, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        }
    } else if (pix_log2 != 6u) {  // (scenes without instances: the items are split, their primary rays walk)
        switch ((scaled ? 4u : 0u) | (count_visits ? 2u : 0u) | (pipelined ? 1u : 0u)) {
            case 7u: e = launch_render_variant<MODE_FLAT, true, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 6u: e = launch_render_variant<MODE_FLAT, true, false, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 5u: e = launch_render_variant<MODE_FLAT, false, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 4u: e = launch_render_variant<MODE_FLAT, false, false, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 3u: e = launch_render_variant<MODE_FLAT, true, true, false, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 2u: e = launch_render_variant<MODE_FLAT, true, false, false, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            case 1u: e = launch_render_variant<MODE_FLAT, false, true, false, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
            default: e = launch_render_variant<MODE_FLAT, false, false, false, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        }
    } else
    switch (variant) {
        case 19u: e = launch_render_variant<MODE_FLAT, true, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 18u: e = launch_render_variant<MODE_FLAT, true, false, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 17u: e = launch_render_variant<MODE_FLAT, false, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 16u: e = launch_render_variant<MODE_FLAT, false, false, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 7u: e = launch_render_variant<MODE_TWO_LEVEL, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 6u: e = launch_render_variant<MODE_TWO_LEVEL, true, false>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 5u: e = launch_render_variant<MODE_TWO_LEVEL, false, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 4u: e = launch_render_variant<MODE_TWO_LEVEL, false, false>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 3u: e = launch_render_variant<MODE_FLAT, true, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 2u: e = launch_render_variant<MODE_FLAT, true, false>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        case 1u: e = launch_render_variant<MODE_FLAT, false, true>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
        default: e = launch_render_variant<MODE_FLAT, false, false>(S, G, out, d_counters, d_heads, cu_count, tuning.blocks_per_cu, stream); break;
    }
    if (e != hipSuccess || n_chunks <= 1u) return e;
    const uint32_t n_floats = width * height * 3u;
    hipLaunchKernelGGL(k_reduce_chunks, dim3((n_floats + 255u) / 256u), dim3(256), 0, stream, d_partial, n_floats, n_chunks, d_accum);
    return hipGetLastError();
}

template <int MODE, bool COUNT>
static hipError_t launch_trace_variant(const SceneDev& S, const rrt_ray* d_rays, uint32_t n, float tmin, float tmax,
                                       rrt_hit* d_out, unsigned long long* d_counters, unsigned int* d_head, int cu_count,
                                       const RenderTuning& tuning, hipStream_t stream) {
    static std::atomic<int> cache{0};  // per template instantiation
    int blocks_per_cu = 1;
    if (hipError_t e = resident_blocks_per_cu(k_trace_stream<MODE, COUNT>, cache, blocks_per_cu); e != hipSuccess) return e;
    if (tuning.blocks_per_cu != 0u && (int)tuning.blocks_per_cu < blocks_per_cu) blocks_per_cu = (int)tuning.blocks_per_cu;
    unsigned long long blocks = (unsigned long long)blocks_per_cu * (unsigned)cu_count;
    const unsigned long long needed = ((unsigned long long)n + RENDER_BLOCK - 1ull) / RENDER_BLOCK;
    if (blocks > needed) blocks = needed;
    hipLaunchKernelGGL((k_trace_stream<MODE, COUNT>), dim3((unsigned)blocks), dim3(RENDER_BLOCK), 0, stream, S, d_rays, n, tmin,
                       tmax, d_out, d_head, tuning.refill_min, tuning.node_share_64ths, d_counters);
    return hipGetLastError();
}

hipError_t launch_intersect(const SceneDev& S, const rrt_ray* d_rays, size_t n, float tmin, float tmax, rrt_hit* d_out,
                            unsigned long long* d_counters, unsigned int* d_head, int cu_count, bool count_visits,
                            const RenderTuning& tuning, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int mode = S.unroll_shift != 0u ? MODE_UNROLLED : (S.has_instances != 0u ? MODE_TWO_LEVEL : MODE_FLAT);
    if (tuning.intersect_static != 0u) {  // experiment knob: the plain one-lane-per-ray kernel
        dim3 grid((unsigned)((n + WAVE - 1) / WAVE)), block(WAVE);
        if (mode == MODE_UNROLLED)
            hipLaunchKernelGGL((k_intersect<MODE_UNROLLED>), grid, block, 0, stream, S, d_rays, n, tmin, tmax, d_out);
        else if (mode == MODE_TWO_LEVEL)
            hipLaunchKernelGGL((k_intersect<MODE_TWO_LEVEL>), grid, block, 0, stream, S, d_rays, n, tmin, tmax, d_out);
        else
            hipLaunchKernelGGL((k_intersect<MODE_FLAT>), grid, block, 0, stream, S, d_rays, n, tmin, tmax, d_out);
        return hipGetLastError();
    }
    const uint32_t m = (uint32_t)n;
    switch (mode * 2 + (count_visits ? 1 : 0)) {
        case 5: return launch_trace_variant<MODE_UNROLLED, true>(S, d_rays, m, tmin, tmax, d_out, d_counters, d_head, cu_count, tuning, stream);
        case 4: return launch_trace_variant<MODE_UNROLLED, false>(S, d_rays, m, tmin, tmax, d_out, d_counters, d_head, cu_count, tuning, stream);
        case 3: return launch_trace_variant<MODE_TWO_LEVEL, true>(S, d_rays, m, tmin, tmax, d_out, d_counters, d_head, cu_count, tuning, stream);
        case 2: return launch_trace_variant<MODE_TWO_LEVEL, false>(S, d_rays, m, tmin, tmax, d_out, d_counters, d_head, cu_count, tuning, stream);
        case 1: return launch_trace_variant<MODE_FLAT, true>(S, d_rays, m, tmin, tmax, d_out, d_counters, d_head, cu_count, tuning, stream);
        default: return launch_trace_variant<MODE_FLAT, false>(S, d_rays, m, tmin, tmax, d_out, d_counters, d_head, cu_count, tuning, stream);
    }
}

hipError_t launch_add(float* d_dst, const float* d_src, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t threads = n / 4u + 1u;
    hipLaunchKernelGGL(k_add, dim3((unsigned)((threads + 255u) / 256u)), dim3(256), 0, stream, d_dst, d_src, n);
    return hipGetLastError();
}

hipError_t launch_check_arith(unsigned long long* d_bad, hipStream_t stream) {
    hipLaunchKernelGGL(k_check_arith, dim3(1u << 20), dim3(256), 0, stream, d_bad);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_rgb8,
                          hipStream_t stream) {
    const uint32_t n = width * height;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tonemap, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_accum, width, height, spp, d_rgb8);
    return hipGetLastError();
}

}  // namespace rrt
