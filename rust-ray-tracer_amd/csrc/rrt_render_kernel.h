// rrt_render_kernel.h -- the persistent path-tracing megakernel k_render and the launcher of one instantiation of it.
// Included by the three translation units that instantiate it: rrt_kernels.hip (scenes without instances, two-level
// walk), rrt_kernels_unrolled.hip (unrolled instances) and rrt_kernels_unrolled_lists.hip (unrolled instances, primary
// rays from per-pixel lists), which are compiled with different scheduling strategies (the Makefile says which and why).
//
// k_render is a persistent-wavefront kernel: the grid is sized to what the chip holds at once, and every
// 64-lane wavefront pulls work items from eight queues (one per group of workgroups that share an XCD).
// A work item is (8x8 pixel tile, chunk of samples).  Inside an item one lane owns one pixel and runs the
// reference's per-pixel loop (src/main.rs:87-103) flattened over (sample, bounce): every loop trip
// traces exactly one ray, and a lane whose path ended starts its next sample at once, so lanes are never
// parked waiting for the longest path in the wave.  The radiance sum stays in registers and is written
// once per item (12 bytes per pixel).  Traversal stacks live in LDS (STACK_DEPTH dwords per lane, laid
// out [entry][thread] so that a wave access touches 64 consecutive banks).
// The kernel has two forms of the loop over items (template parameter PIPELINED): one item at a time, or two
// items in flight per wavefront with every lane moving from one to the next by itself.  The two forms are kept as
// separate texts on purpose: written over shared helpers the pipelined form compiled to 3-20 % slower code
// (register allocation at the 128-VGPR limit of four waves per SIMD).
#ifndef RRT_RENDER_KERNEL_H
#define RRT_RENDER_KERNEL_H

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "rrt_device.h"
#include "rrt_kernels.h"

namespace rrt {

static constexpr int WAVE = 64;
#ifndef RRT_TILE_W_LOG2
#define RRT_TILE_W_LOG2 3  // a work item's 64 pixels form a tile of 2^RRT_TILE_W_LOG2 pixels' width (8x8 by default)
#endif
static constexpr uint32_t TILE_W = 1u << RRT_TILE_W_LOG2, TILE_H = 64u / TILE_W;
// Split items (SPLIT instantiations): the 64 lanes of a work item are 2^G.pix_log2 pixels (a tile of 2^G.tw_log2 pixels' width) times
// 64 / pixels consecutive sample chunks of each pixel; the launcher picks the split (launch_render).  RRT_PIPE_PIX_LOG2
// (build knob, 1 .. 6) fixes it instead.
static constexpr int RENDER_BLOCK = 256;  // 4 independent wavefronts per workgroup
static constexpr uint32_t TRACE_BLOCK = 512;  // ray indices a wavefront of the streaming kernel draws at a time
#ifndef RRT_RENDER_STACK
#define RRT_RENDER_STACK 16
#endif
static constexpr int RENDER_STACK = RRT_RENDER_STACK;   // traversal stack entries per lane the render kernel keeps in LDS (short stack)
// (With 20 the private remainder is 19 entries, small enough for the compiler to turn it into registers: the one-item
// form then spills 95 of them and runs eight times slower.  Larger LDS parts bought nothing anyway: walks seldom
// leave the first 16 entries.)
#ifndef RRT_INST_WAVES
#define RRT_INST_WAVES 7  // workgroups per CU (= wavefronts per SIMD) the instanced forms of k_render are compiled for
#endif
#ifndef RRT_INST_STACK
#define RRT_INST_STACK 16  // ... and the traversal stack entries per lane they keep in LDS
#endif
static constexpr int XF_LDS_SLOTS = 128;  // unrolled scenes: instance transforms (identity + instances) a workgroup of k_render keeps in LDS
static constexpr uint32_t N_QUEUES = 8;   // workgroup ids congruent mod 8 are dealt to one XCD and share its L2

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// Tile number -> tile coordinates (bands of G.band_tiles tiles' width, row-major inside a band).
__device__ __forceinline__ void tile_coords(const RenderGeom& G, uint32_t tile, uint32_t& tx, uint32_t& ty) {
    const uint32_t band = min(tile / G.band_items, G.full_bands);
    const uint32_t r = tile - band * G.band_items;
    const uint32_t w = band < G.full_bands ? G.band_tiles : G.tiles_x - G.full_bands * G.band_tiles;
    ty = r / w;
    tx = band * G.band_tiles + r % w;
}

// First work item of queue q: the items are split into N_QUEUES contiguous ranges, i.e. contiguous image
// regions, so that the 4 MiB L2 of an XCD sees one eighth of the frame.
__device__ __forceinline__ uint32_t queue_begin(uint32_t q, uint32_t n_work) {
    return (uint32_t)(((unsigned long long)q * n_work) / N_QUEUES);
}

#ifndef RRT_PIPE_STEPS
#define RRT_PIPE_STEPS 5  // node steps between two continuation votes of a node pass, pipelined form
#endif
#ifndef RRT_UNROLLED_STEPS
#define RRT_UNROLLED_STEPS 6  // ... of the pipelined form on unrolled scenes (measured: 4 / 5 / 6 / 7 / 8 -> 168.4 / 168.7 / 163.3 / 165.3 / 168.7 ms)
#endif
#ifndef RRT_ONE_STEPS
#define RRT_ONE_STEPS 2  // ... and in the one-item form
#endif
#ifndef RRT_UNROLLED_WAVES
#define RRT_UNROLLED_WAVES 7  // workgroups per CU the pipelined form is compiled for on unrolled scenes
#endif
#ifndef RRT_WALK_PRIO
#define RRT_WALK_PRIO 0  // wavefront priority during the walk phase of the pipelined form (0: none set)
#endif
#ifndef RRT_SHADE_PRIO
#define RRT_SHADE_PRIO 0  // ... and during the shade phase
#endif
#ifndef RRT_LANE_FLIP
#define RRT_LANE_FLIP 63  // pipelined form: lane l takes pixel l ^ RRT_LANE_FLIP of the tile in every second item (0: always pixel l)
#endif
#ifndef RRT_ITEM_RING
#define RRT_ITEM_RING 2  // pipelined form: work items a wavefront holds at a time (2 or 3)
#endif
#ifndef RRT_FLAT_WAVES
#define RRT_FLAT_WAVES 7  // workgroups per CU the pipelined form is compiled for on scenes without instances
#endif
#ifndef RRT_RESET_RAY
#define RRT_RESET_RAY 1  // pipelined form with pixel lists: 1 = slab constants recomputed at the start of every walk phase
#endif
#ifndef RRT_ONE_LIST_ROUNDS
#define RRT_ONE_LIST_ROUNDS 2  // one-item form with pixel lists: rounds of a shade phase (the new samples of all but the last take their lists)
#endif
#ifndef RRT_UL_STEPS
#define RRT_UL_STEPS 5  // unrolled scenes with lists: node steps between two votes of the pipelined form (the walks are bounce rays only)
#endif
#ifndef RRT_UL_STACK_CUT
#define RRT_UL_STACK_CUT 4  // unrolled scenes with lists: entries of the traversal stack moved from LDS to private memory
#endif
#ifndef RRT_UL_WAVES
#define RRT_UL_WAVES RRT_UNROLLED_WAVES  // unrolled scenes with lists: workgroups per CU the pipelined form is compiled for
#endif
#ifndef RRT_UL_XF
#define RRT_UL_XF 1  // unrolled scenes with lists: 1 = the table of instance transforms is copied to LDS
#endif
#ifndef RRT_ONE_ITEM_WAVES
#define RRT_ONE_ITEM_WAVES 6  // workgroups per CU the one-item form (small scenes) is compiled for
#endif
// SCALED: the node steps use the scaled form of the slab arithmetic (rrt_device.h: trav_set_ray_scaled), which the host
// selects when the scene is walked on one grid and its camera's ray directions are of ordinary magnitude.
// SPLIT (scenes without instances; unrolled scenes together with LISTS): an item's 64 lanes are 2^G.pix_log2 pixels times several of their
// sample chunks instead of 64 pixels times one (launch_render picks the split); a template parameter because the
// general index arithmetic costs the other scenes 1-3 % when it is merely switched off at run time.
// LISTS (split items of at most LIST_PIXELS pixels, scenes walked as one world-space tree): a kernel run before this one
// (k_pixel_lists) has collected, for every pixel, the primitives a camera ray of the pixel can hit.  When a
// wavefront draws an item it copies the lists of the item's pixels to LDS; a lane that starts a sample of a pixel with a
// list resolves its primary ray against that list inside the shading phase -- a couple of primitive tests instead of a
// walk -- and shades it in a second round of the same phase, so that the walk phase of a trip carries bounce rays
// only and a sample costs one trip less.
static constexpr int LIST_PIXELS = 8;   // pixels of an item that get a list
// ROOMY (pipelined form with lists): compiled for six workgroups per CU instead of seven -- 84 registers, which the
// shade phase with a list's primitive tests in it needs to stay out of scratch memory.  Which of the two is faster
// depends on the scene (measured on MI355X: the 1M-triangle scene 764 -> 740 ms with six, MetalRoughSpheres 258 -> 262):
// the caller chooses (RenderTuning::list_waves).
template <int MODE, bool COUNT, bool PIPELINED, bool SCALED, bool SPLIT = false, bool LISTS = false, bool ROOMY = false>
__global__ __launch_bounds__(RENDER_BLOCK, MODE == MODE_TWO_LEVEL ? RRT_INST_WAVES : (PIPELINED ? (MODE == MODE_UNROLLED ? (LISTS ? RRT_UL_WAVES : RRT_UNROLLED_WAVES) : (ROOMY ? RRT_FLAT_WAVES - 1 : RRT_FLAT_WAVES)) : RRT_ONE_ITEM_WAVES)) void k_render(SceneDev S, RenderGeom G, float* __restrict__ out,
                                                         unsigned long long* __restrict__ counters,
                                                         unsigned int* __restrict__ heads) {
    constexpr bool INST = MODE == MODE_TWO_LEVEL;
    static_assert(!(SCALED && INST), "the scaled slab arithmetic is for scenes walked on one grid");
    static_assert(!LISTS || (SPLIT && MODE != MODE_TWO_LEVEL), "pixel lists: split items, one world-space tree");
    static_assert(!ROOMY || (LISTS && PIPELINED), "the six-workgroup build is for the pipelined form with lists");
    __shared__ __align__(16) uint32_t s_lists[LISTS ? (RENDER_BLOCK / WAVE) * (PIPELINED ? RRT_ITEM_RING : 1) * LIST_PIXELS * PIXEL_LIST_DWORDS : 1];
    // RENDER_STACK entries per lane of the traversal stack live in LDS, the rest in private memory (six workgroups per
    // CU, seven in the pipelined form, whose large scenes gain more from the extra wavefront than they lose to its
    // spills).  Instanced scenes add six floats per lane for the world-space ray of a lane that walks inside a mesh.
    // (an unrolled scene's kernel with lists keeps RRT_UL_STACK_CUT entries fewer in LDS: its table of instance transforms
    // and the lists would otherwise leave room for six workgroups per CU instead of seven)
    constexpr int K = INST ? RRT_INST_STACK : (MODE == MODE_UNROLLED && LISTS ? RENDER_STACK - RRT_UL_STACK_CUT : RENDER_STACK);
    __shared__ int s_stack[K * RENDER_BLOCK];
    __shared__ float s_wray[INST ? 6 * RENDER_BLOCK : 1];
    __shared__ uint32_t s_items[(RENDER_BLOCK / WAVE) * 8 * RRT_ITEM_RING];  // per wavefront: the records of its work items in flight
    // Unrolled scenes: the instance transforms a triangle leaf needs (offset, sin, cos: the first 32 bytes of a record)
    // are copied to LDS once per workgroup when they fit, slot 0 = the identity record in front of the table.
    __shared__ vec4f s_xf[MODE == MODE_UNROLLED && !(LISTS && RRT_UL_XF == 0) ? 2 * XF_LDS_SLOTS : 1];
    const lds_float4* xf_lds = nullptr;
    if (MODE == MODE_UNROLLED && !(LISTS && RRT_UL_XF == 0) && S.inst_count + 1u <= (uint32_t)XF_LDS_SLOTS) {
        const vec4f* src = reinterpret_cast<const vec4f*>(S.insts - 1);  // four 16-byte quarters per 64-byte record
        for (uint32_t i = threadIdx.x; i < 2u * (S.inst_count + 1u); i += RENDER_BLOCK) s_xf[i] = src[(i >> 1) * 4u + (i & 1u)];
        __syncthreads();
        xf_lds = (const lds_float4*)s_xf;
    }
    const uint32_t lane = threadIdx.x & 63u;
    lds_int* stack = (lds_int*)(s_stack + threadIdx.x);
    lds_float* wray = INST ? (lds_float*)(s_wray + threadIdx.x) : nullptr;
    lds_int* lim = (lds_int*)s_stack + K * RENDER_BLOCK;  // where entry K of the first lane would lie: the end of the LDS part
    int deep[STACK_DEPTH - K];  // stack entries below the LDS window (private memory; rarely touched)
    const uint32_t width = G.width, height = G.height;
    const V3 bg = v3(S.background[0], S.background[1], S.background[2]);

    uint32_t n_rays = 0;
    TraceCounters cnt = {0u, 0u, 0u, 0u, 0u};
    unsigned long long slots_done = 0ull, slots_leaf = 0ull;  // lane 0, COUNT only
    uint32_t q = blockIdx.x % N_QUEUES;
    uint32_t exhausted = 0;
    // A ticket is always drawn one item ahead, so that the round trip of the atomic overlaps rendering.
    // Every drawn ticket is resolved by the wave that drew it before that wave exits.
    unsigned int drawn = 0;
    if (lane == 0) drawn = atomicAdd(&heads[q], 1u);
    // Next work item of this wave (own queue first, then the others' leftovers); NO_WORK once every queue is empty.
    constexpr uint32_t NO_WORK = 0xFFFFFFFFu;
    auto draw = [&]() -> uint32_t {
        uint32_t work = NO_WORK;
        while (exhausted < N_QUEUES) {
            const unsigned int ticket = __builtin_amdgcn_readfirstlane(drawn);
            const uint32_t b = queue_begin(q, G.n_work), e = queue_begin(q + 1u, G.n_work);
            if (ticket < e - b) {
                work = b + ticket;
                break;
            }
            q = (q + 1u) % N_QUEUES;
            if (++exhausted >= N_QUEUES) break;
            if (lane == 0) drawn = atomicAdd(&heads[q], 1u);
        }
        if (work != NO_WORK && lane == 0) drawn = atomicAdd(&heads[q], 1u);
        return work;
    };

    // LISTS: the wavefront copies the lists of an item's pixels (tile at (x0, y0), 2^pix_log2 pixels in rows of 2^tw_log2)
    // from the frame's list buffer to slot `slot` of its part of s_lists: one 16-byte piece per lane.
    [[maybe_unused]] auto stage_lists = [&](uint32_t slot, uint32_t slots, uint32_t x0, uint32_t y0, uint32_t pix_log2, uint32_t tw_log2) {
        constexpr uint32_t QUADS = PIXEL_LIST_DWORDS / 4u;
        if (lane < (QUADS << pix_log2)) {
            const uint32_t pix = lane / QUADS, quad = lane % QUADS;
            const uint32_t px_x = x0 + (pix & ((1u << tw_log2) - 1u)), px_y = y0 + (pix >> tw_log2);
            uint4 v = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);  // (a pixel outside the frame is never taken)
            if (px_x < width && px_y < height)
                v = reinterpret_cast<const uint4*>(G.lists + (size_t)(px_y * width + px_x) * PIXEL_LIST_DWORDS)[quad];
            uint32_t* dst = s_lists + (((threadIdx.x >> 6) * slots + slot) * (uint32_t)LIST_PIXELS + pix) * PIXEL_LIST_DWORDS + quad * 4u;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    };
    // LISTS: the closest hit of a primary ray over the primitives of its pixel's list (at dword `lst` of s_lists), in
    // ascending order of the earliest ray parameter at which the pixel's rays can reach them: nothing behind the hit
    // already found needs a test.  Leaves the lane as a completed walk does.
    [[maybe_unused]] auto resolve_listed = [&](uint32_t lst, V3 o, V3 d, Trav& T) {
        T.best.t = __builtin_inff(); T.best.u = 0.f; T.best.v = 0.f; T.best.what = WHAT_NONE; T.best.inst = -1;
        T.cur = REF_DONE;
        const lds_u32* list = (const lds_u32*)s_lists + lst;
        const uint32_t n_listed = list[0];
        if (COUNT && first_active_lane()) cnt.wave_prims++;
        for (uint32_t i = 0; i < n_listed; ++i) {
            if (__uint_as_float(list[3u + 2u * i]) > T.best.t) break;
            test_listed_leaf<COUNT, MODE>(S, list[2u + 2u * i], o, d, S.t_min, T.best, cnt, xf_lds);
        }
    };

    if (!PIPELINED) {
        // One item at a time: the lanes start an item together and wait for its last path.  The form for
        // scenes whose trips are short (a handful of node steps, mostly first-ray misses), where the
        // bookkeeping of the pipelined form below costs more than the waiting it removes.
        for (;;) {
            const uint32_t work = draw();
            if (work == NO_WORK) break;  // every queue is empty: this wave is done
            uint32_t tile, chunk, x, y;
            uint32_t tx, ty;
            bool inside;
            if constexpr (SPLIT) {
                // a few pixels x as many consecutive chunks of each as make 64 lanes (see the pipelined form below):
                // the lane's pixel is the low part of its number, its chunk the item's first chunk + the high part
                const uint32_t pix = lane & ((1u << G.pix_log2) - 1u);
                tile = work / G.n_groups;
                chunk = ((work % G.n_groups) << (6u - G.pix_log2)) + (lane >> G.pix_log2);
                tile_coords(G, tile, tx, ty);
                x = (tx << G.tw_log2) + (pix & ((1u << G.tw_log2) - 1u));
                y = (ty << (G.pix_log2 - G.tw_log2)) + (pix >> G.tw_log2);
                inside = x < width && y < height && chunk < G.n_chunks;
            } else {
                tile = work / G.n_chunks;
                chunk = work % G.n_chunks;
                tile_coords(G, tile, tx, ty);
                x = tx * TILE_W + (lane & (TILE_W - 1u));
                y = ty * TILE_H + (lane >> RRT_TILE_W_LOG2);
                inside = x < width && y < height;
            }
            // samples [first, first + spp) of this chunk, relative to the launch's first sample
            const uint32_t first = chunk_first(chunk, G.chunk_q, G.chunk_r);
            const uint32_t spp = chunk_len(chunk, G.chunk_q, G.chunk_r);  // (a lane whose chunk does not exist is not `inside`)
            const uint32_t spp_offset = G.spp_offset + first;
            const uint32_t pixel = y * width + x;

            V3 sum = v3(0.f, 0.f, 0.f);
            uint32_t s = inside ? 0u : spp;  // sample counter of this lane
            bool has_ray = s < spp;          // this lane has a ray in flight (being walked, or waiting to be shaded)
            V3 o, d, beta;
            uint32_t depth = 0;
            Rng rng;
            Trav T;
            T.cur = REF_DONE;
            T.top = stack;
            [[maybe_unused]] uint32_t lst = 0u;      // LISTS: where the list of the lane's pixel starts in s_lists
            [[maybe_unused]] bool has_list = false;  // ... and whether the pixel has one
            if constexpr (LISTS) {
                // (the lists of the last item are free: every lane of the wavefront is through with it)
                __builtin_amdgcn_wave_barrier();
                stage_lists(0u, 1u, tx << G.tw_log2, ty << (G.pix_log2 - G.tw_log2), G.pix_log2, G.tw_log2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                lst = ((threadIdx.x >> 6) * (uint32_t)LIST_PIXELS + (lane & ((1u << G.pix_log2) - 1u))) * PIXEL_LIST_DWORDS;
                has_list = inside && s_lists[lst] != 0xFFFFFFFFu;
            }
            if (has_ray) {
                rng = rng_init(S.seed, pixel, spp_offset + s);
                const float u = (float)x + rng_f32(rng);  // main.rs:92-93
                const float v = (float)y + rng_f32(rng);
                rng_skip(rng);                                // camera.rs:67: the ray time draw (unused by static scenes)
                camera_ray(S.cam, u, v, width, height, o, d);
                beta = v3(1.f, 1.f, 1.f);
                depth = S.max_depth;
                if (LISTS && has_list) resolve_listed(lst, o, d, T);
                else trav_begin<MODE, SCALED>(S, o, d, __builtin_inff(), stack, RENDER_BLOCK, T);
            }

            for (;;) {
                // ---- walk phase: node passes and leaf passes alternate until every lane's walk is complete.
                // A pass costs the same however many lanes take part, so a node pass only runs (and only keeps
                // running) while a fair share of the walking lanes is descending; below that share the lanes
                // waiting at leaves get their primitive test first.  (Measured on MI355X: the share rule raises the
                // lanes active per node pass from 28 % to 41 % on the 1M-triangle scene.  Leaving the walk phase
                // early to shade finished lanes did not pay: shading passes with few lanes cost more than they save.)
                for (;;) {
                    const uint32_t n_walk = (uint32_t)__popcll(__ballot(T.cur != REF_DONE));
                    if (n_walk == 0u) break;
                    // the last few walkers of a trip are left to finish in the next one, once enough lanes wait to shade
                    if (n_walk <= G.leave_walk && (uint32_t)__popcll(__ballot(has_ray && T.cur == REF_DONE)) >= G.leave_ready) break;
                    // (instanced scenes: a lane due for a change of space counts as descending, the node pass resolves it)
                    const uint32_t n_desc = (uint32_t)__popcll(__ballot(T.cur >= 0 || (INST && trav_due(T.cur))));
                    const uint32_t share = (n_walk * G.node_share_64ths) >> 6;
                    if (n_desc >= share) {  // share <= n_walk, so a pass in which every walker descends always runs
                        if (COUNT) {
                            // idle lane slots of this node pass, by what the idle lanes are waiting for
                            // (the pass is as long as the lane that stayed in it longest)
                            const uint32_t nodes0 = cnt.nodes;
                            trav_nodes<MODE, COUNT, K, RRT_ONE_STEPS, SCALED>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, share, lim, deep, wray);
                            uint32_t steps = cnt.nodes - nodes0;
                            for (int off = 32; off > 0; off >>= 1) steps = max(steps, (uint32_t)__shfl_xor((int)steps, off, WAVE));
                            if (lane == 0u) {
                                slots_done += (unsigned long long)steps * (64u - n_walk);
                                slots_leaf += (unsigned long long)steps * (n_walk - n_desc);
                            }
                        } else {
                            trav_nodes<MODE, COUNT, K, RRT_ONE_STEPS, SCALED>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, share, lim, deep, wray);
                        }
                    }
                    if (T.cur < 0 && T.cur != REF_DONE && !(INST && trav_due(T.cur))) trav_leaf<MODE, COUNT, K>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, lim, deep, wray, xf_lds);
                }
                // ---- shade phase: every lane whose walk is complete shades its hit and starts its next ray.
                // LISTS: a lane that starts a sample of a pixel with a list resolves the primary ray against the list on
                // the spot and is shaded in the phase's next round; the new samples of the last round walk the tree.
                for (int round = 0; round < (LISTS ? RRT_ONE_LIST_ROUNDS : 1); ++round) {
                [[maybe_unused]] bool listed = false;
                if (LISTS && round != 0 && __ballot(has_ray && T.cur == REF_DONE) == 0ull) break;
                if (has_ray && T.cur == REF_DONE) {
                    n_rays++;
                    if (COUNT && first_active_lane()) cnt.wave_shade++;
                    bool alive;
                    // What the path emits towards the camera.  Only the last surface of a path emits (a light or the
                    // background; neither scatters), so the sample's radiance is this one term: no sum is carried
                    // from bounce to bounce.
                    V3 radiance = v3(0.f, 0.f, 0.f);
                    if (T.best.what == WHAT_NONE) {
                        radiance = radiance + mulv(beta, bg);  // ray.rs:126
                        alive = false;
                    } else {
                        const Surface sf = surface_at(S, T.best, o, d);
                        alive = shade(S, sf, o, d, beta, radiance, rng);
                    }
                    depth--;
                    if (depth == 0u) alive = false;  // ray.rs:85-87: depth 0 contributes black
                    if (!alive) {
                        sum = sum + radiance;  // main.rs:95
                        s++;
                        if (s < spp) {
                            rng = rng_init(S.seed, pixel, spp_offset + s);
                            const float u = (float)x + rng_f32(rng);
                            const float v = (float)y + rng_f32(rng);
                            rng_skip(rng);
                            camera_ray(S.cam, u, v, width, height, o, d);
                            beta = v3(1.f, 1.f, 1.f);
                            depth = S.max_depth;
                            if constexpr (LISTS) listed = has_list && round + 1 < RRT_ONE_LIST_ROUNDS;
                        } else {
                            has_ray = false;
                        }
                    }
                    if (has_ray && !(LISTS && listed)) trav_begin<MODE, SCALED>(S, o, d, __builtin_inff(), stack, RENDER_BLOCK, T);
                }
                if constexpr (LISTS) {
                    if (__ballot(listed) == 0ull) break;
                    if (listed) resolve_listed(lst, o, d, T);
                }
                }
                if (__ballot(has_ray) == 0ull) break;  // every lane has rendered all samples of this item
            }

            if (inside) {
                float* px = out + 3ull * ((size_t)chunk * width * height + pixel);
                px[0] = sum.x; px[1] = sum.y; px[2] = sum.z;
            }
        }
    } else {
        // The wave holds two items, the current one and the next one.  A lane that has finished its pixel of the
        // current item moves on to the next one by itself, so the lanes whose pixels took few rays do not wait for the
        // others (a lane that is done with both does).  When no lane is left on the current item, the next one takes
        // its place and another is drawn.  Every (item, pixel) sum is still formed by one lane in sample order,
        // whatever the interleaving, so the image does not depend on it.
        // The two items' records live in LDS (x0, y0, chunk, sample count, first sample index; count 0 = no item): a
        // lane reads the one it moves to, the wave writes one when it draws an item.
        [[maybe_unused]] const uint32_t pix_log2 = SPLIT ? G.pix_log2 : 6u, tw_log2 = SPLIT ? G.tw_log2 : (uint32_t)RRT_TILE_W_LOG2;
        constexpr uint32_t RING = RRT_ITEM_RING;
        static_assert(RING >= 2u && RING <= 3u, "two or three items in flight");
        uint32_t* items = s_items + (threadIdx.x >> 6) * (8u * RING);
        uint32_t cur_slot = 0u;    // record of the current item; the item p places ahead has record (cur_slot + p) % RING
        uint32_t ahead_valid = 0u; // items behind the current one that exist (they are drawn in order: a prefix of the ring)
        uint32_t flip = 0u;        // alternates from item to item: which pixel of the tile a lane takes (RRT_LANE_FLIP)
        auto slot_of = [&](uint32_t pos) -> uint32_t {
            const uint32_t t = cur_slot + pos;
            return t >= RING ? t - RING : t;
        };
        auto draw_into = [&](uint32_t slot) -> bool {
            const uint32_t work = draw();
            // record: x0, y0, first chunk, samples of a short chunk (one chunk per item: of this chunk), first sample
            // index of the first chunk, pixel permutation, chunks of the item that exist, how many of them (the item's
            // first ones) hold one sample more
            uint32_t x0 = 0u, y0 = 0u, chunk = 0u, n_spp = 0u, first_sample = 0u;
            [[maybe_unused]] uint32_t n_valid = 0u, n_long = 0u;
            if (work != NO_WORK) {
                if constexpr (SPLIT) {
                    const uint32_t tile = work / G.n_groups;
                    chunk = (work % G.n_groups) << (6u - pix_log2);  // the item's first chunk; lane group s renders chunk + s
                    uint32_t tx, ty;
                    tile_coords(G, tile, tx, ty);
                    x0 = tx << tw_log2;
                    y0 = ty << (pix_log2 - tw_log2);
                    n_spp = G.chunk_q;
                    first_sample = G.spp_offset + chunk_first(chunk, G.chunk_q, G.chunk_r);
                    n_valid = min(64u >> pix_log2, G.n_chunks - chunk);
                    n_long = min(n_valid, G.chunk_r - min(chunk, G.chunk_r));
                } else {
                    const uint32_t tile = work / G.n_chunks;
                    chunk = work % G.n_chunks;
                    uint32_t tx, ty;
                    tile_coords(G, tile, tx, ty);
                    x0 = tx * TILE_W;
                    y0 = ty * TILE_H;
                    n_spp = chunk_len(chunk, G.chunk_q, G.chunk_r);
                    first_sample = G.spp_offset + chunk_first(chunk, G.chunk_q, G.chunk_r);
                }
            }
            if (lane == 0u) {
                uint32_t* rec = items + slot * 8u;
                rec[0] = x0; rec[1] = y0; rec[2] = chunk; rec[3] = n_spp; rec[4] = first_sample; rec[5] = flip;
                if constexpr (SPLIT) { rec[6] = n_valid; rec[7] = n_long; }
            }
            if constexpr (LISTS) {
                // (the slot's old lists are free: nobody is on the item that held them)
                if (work != NO_WORK) stage_lists(slot, RING, x0, y0, pix_log2, tw_log2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            flip ^= (uint32_t)RRT_LANE_FLIP;
            return work != NO_WORK;
        };
        draw_into(0u);
        for (uint32_t p = 1u; p < RING; ++p) ahead_valid += draw_into(p) ? 1u : 0u;

        // per lane: `stage` = which of the items in flight it is on (0: current, 1: next, ..., RING: done with all),
        // + 4 once it has taken its pixel of that item; then its place in the item
        uint32_t stage = 0u;
        // (kept small: the kernel runs at the register budget of six waves per SIMD)
        uint32_t xy = 0;           // the lane's pixel, x | y << 16
        uint32_t next_sample = 0;  // absolute index of the next sample to start
        uint32_t left = 0;         // samples of the item not yet finished
        [[maybe_unused]] uint32_t lst = 0;  // LISTS: where the list of the lane's pixel starts in s_lists
        bool has_ray = false;    // a ray is in flight (being walked, or waiting to be shaded)
        V3 sum = v3(0.f, 0.f, 0.f);
        V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f), beta = v3(0.f, 0.f, 0.f);
        uint32_t depth = 0;
        Rng rng = {0u, 0u};
        Trav T;
        T.cur = REF_DONE;
        T.top = stack;

        for (;;) {
            // ---- item rotation (touches nothing of a lane but its stage)
            bool out_of_work = false;
            while (__ballot((stage & 3u) == 0u) == 0ull) {
                if (ahead_valid == 0u) {
                    out_of_work = true;  // nobody on the current item and no next item: every lane is done
                    break;
                }
                stage--;
                const bool drew = draw_into(cur_slot);  // the old current item's record is free: it becomes the last one's
                ahead_valid = ahead_valid - 1u + (drew ? 1u : 0u);
                cur_slot = slot_of(1u);
            }
            if (out_of_work) break;

#if RRT_WALK_PRIO || RRT_SHADE_PRIO
            __builtin_amdgcn_s_setprio(RRT_WALK_PRIO);
#endif
#if RRT_RESET_RAY
            // LISTS: the slab constants of every lane's ray are formed anew from the ray at the start of each walk
            // phase (a straggler's are the same values again): six registers that are then free during the shade phase,
            // where the primitive tests of the lists make the kernel's register pressure peak (30 -> 26 spilled
            // registers; 1M-triangle scene 802 -> 795 ms.  Without lists the recomputation costs more than the spills:
            // the ten-million-triangle scenes 160 -> 166 ms.)
            if constexpr (LISTS) {
                if (SCALED) trav_set_ray_scaled(T, o, d, S);
                else trav_set_ray(T, o, d, S.grid.origin, S.grid.step);
            }
#endif
            // ---- walk phase: node passes and leaf passes alternate until every lane's walk is complete.
            // A pass costs the same however many lanes take part, so a node pass only runs (and only keeps
            // running) while a fair share of the walking lanes is descending; below that share the lanes
            // waiting at leaves get their primitive test first.  (Measured on MI355X: the share rule raises the
            // lanes active per node pass from 28 % to 41 % on the 1M-triangle scene.)
            for (;;) {
                const uint32_t n_walk = (uint32_t)__popcll(__ballot(T.cur != REF_DONE));
                if (n_walk == 0u) break;
                // the last few walkers of a trip are left to finish in the next one, once enough lanes wait to shade
                if (n_walk <= G.leave_walk && (uint32_t)__popcll(__ballot(has_ray && T.cur == REF_DONE)) >= G.leave_ready) break;
                // (instanced scenes: a lane due for a change of space counts as descending, the node pass resolves it)
                const uint32_t n_desc = (uint32_t)__popcll(__ballot(T.cur >= 0 || (INST && trav_due(T.cur))));
                const uint32_t share = (n_walk * G.node_share_64ths) >> 6;
                if (n_desc >= share) {  // share <= n_walk, so a pass in which every walker descends always runs
                    if (COUNT) {
                        // idle lane slots of this node pass, by what the idle lanes are waiting for
                        // (the pass is as long as the lane that stayed in it longest)
                        const uint32_t nodes0 = cnt.nodes;
                        trav_nodes<MODE, COUNT, K, (MODE == MODE_UNROLLED ? (LISTS ? RRT_UL_STEPS : RRT_UNROLLED_STEPS) : RRT_PIPE_STEPS), SCALED>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, share, lim, deep, wray);
                        uint32_t steps = cnt.nodes - nodes0;
                        for (int off = 32; off > 0; off >>= 1) steps = max(steps, (uint32_t)__shfl_xor((int)steps, off, WAVE));
#ifdef RRT_COUNT_NORAY  // experiment: the second idle counter counts the lanes that have no ray at all instead
#if RRT_COUNT_NORAY == 2  // ... only those that could take a pixel of an item in flight and wait for the next shading pass to do it
                        const uint32_t n_noray = (uint32_t)__popcll(__ballot(!has_ray && stage < RING));
#else
                        const uint32_t n_noray = (uint32_t)__popcll(__ballot(!has_ray));
#endif
                        if (lane == 0u) {
                            slots_done += (unsigned long long)steps * (64u - n_walk);
                            slots_leaf += (unsigned long long)steps * n_noray;
                        }
#else
                        if (lane == 0u) {
                            slots_done += (unsigned long long)steps * (64u - n_walk);
                            slots_leaf += (unsigned long long)steps * (n_walk - n_desc);
                        }
#endif
                    } else {
                        trav_nodes<MODE, COUNT, K, (MODE == MODE_UNROLLED ? (LISTS ? RRT_UL_STEPS : RRT_UNROLLED_STEPS) : RRT_PIPE_STEPS), SCALED>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, share, lim, deep, wray);
                    }
                }
                if (T.cur < 0 && T.cur != REF_DONE && !(INST && trav_due(T.cur))) trav_leaf<MODE, COUNT, K, true>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, lim, deep, wray, xf_lds);
            }

#if RRT_WALK_PRIO || RRT_SHADE_PRIO
            __builtin_amdgcn_s_setprio(RRT_SHADE_PRIO);
#endif
            // ---- shade phase: every lane whose walk is complete shades its hit; a lane whose path ended starts its
            // next sample, a lane whose item ended (or that has none yet) takes its pixel of the item it is due for.
            // LISTS: a lane that starts a sample of a pixel with a list resolves the primary ray against the list on
            // the spot; those lanes are shaded in a second round of the phase (whose own new samples walk the tree).
            for (int round = 0; round < (LISTS ? 2 : 1); ++round) {
            const bool shade_now = has_ray && T.cur == REF_DONE;
            bool take = stage < RING;  // on an item without having taken its pixel of it
            [[maybe_unused]] bool listed = false;  // the lane's new ray is resolved against its pixel's list
            if (LISTS && round == 1 && __ballot(shade_now) == 0ull) break;
            if (shade_now || take) {
                bool alive = false;
                V3 radiance = v3(0.f, 0.f, 0.f);  // what the path emits: one term, at its last surface (see above)
                if (shade_now) {
                    n_rays++;
                    if (COUNT && first_active_lane()) cnt.wave_shade++;
                    if (T.best.what == WHAT_NONE) {
                        radiance = radiance + mulv(beta, bg);  // ray.rs:126
                    } else {
                        const Surface sf = surface_at(S, T.best, o, d);
                        alive = shade(S, sf, o, d, beta, radiance, rng);
                    }
                    depth--;
                    if (depth == 0u) alive = false;  // ray.rs:85-87: depth 0 contributes black
                }
                if (!alive) {  // the path ended, or the lane had none
                    bool start = shade_now;
                    if (shade_now) {
                        sum = sum + radiance;  // main.rs:95
                        left--;
                        if (left == 0u) {  // this lane's part of the item is complete: one 12-byte store
                            const uint32_t on = stage & 3u;
                            uint32_t chunk = items[slot_of(on) * 8u + 2u];
                            if constexpr (SPLIT) chunk += (lane ^ items[slot_of(on) * 8u + 5u]) >> pix_log2;
                            const uint32_t pixel = (xy >> 16) * width + (xy & 0xFFFFu);
                            float* px = out + 3ull * ((size_t)chunk * width * height + pixel);
                            px[0] = sum.x; px[1] = sum.y; px[2] = sum.z;
                            has_ray = false;
                            start = false;
                            stage = on + 1u;
                            take = stage < RING;
                        }
                    }
                    if (take) {
                        const uint32_t* rec = items + slot_of(stage) * 8u;
                        // which pixel of the tile the lane takes: mirrored (RRT_LANE_FLIP) in every second item of the
                        // wavefront, so that a lane whose pixel was cheap in one item (sky above a horizon, say) gets
                        // the opposite corner of the tile in the next
                        const uint32_t pl = lane ^ rec[5];
                        uint32_t px_x, px_y, n_spp = rec[3];
                        [[maybe_unused]] uint32_t first_sample = 0u;
                        if constexpr (SPLIT) {
                            // the lane's chunk of the pixel: the item's first chunk + the lane's group
                            const uint32_t pix = pl & ((1u << pix_log2) - 1u), sub = pl >> pix_log2, n_valid = rec[6], n_long = rec[7];
                            px_x = rec[0] + (pix & ((1u << tw_log2) - 1u));
                            px_y = rec[1] + (pix >> tw_log2);
                            first_sample = rec[4] + sub * n_spp + min(sub, n_long);
                            n_spp = sub < n_valid ? n_spp + (sub < n_long ? 1u : 0u) : 0u;
                        } else {
                            px_x = rec[0] + (pl & (TILE_W - 1u));
                            px_y = rec[1] + (pl >> RRT_TILE_W_LOG2);
                        }
                        if (px_x < width && px_y < height && n_spp != 0u) {
                            xy = px_x | (px_y << 16);
                            left = n_spp;
                            next_sample = SPLIT ? first_sample : rec[4];
                            sum = v3(0.f, 0.f, 0.f);
                            if constexpr (LISTS)
                                lst = (((threadIdx.x >> 6) * RING + slot_of(stage)) * (uint32_t)LIST_PIXELS + (pl & ((1u << pix_log2) - 1u))) * PIXEL_LIST_DWORDS;
                            stage += 4u;
                            start = true;
                        } else {
                            stage++;  // nothing for this lane in that item
                        }
                    }
                    if (start) {
                        const uint32_t x = xy & 0xFFFFu, y = xy >> 16;
                        rng = rng_init(S.seed, y * width + x, next_sample);
                        next_sample++;
                        const float u = (float)x + rng_f32(rng);  // main.rs:92-93
                        const float v = (float)y + rng_f32(rng);
                        rng_skip(rng);                                // camera.rs:67: the ray time draw (unused by static scenes)
                        camera_ray(S.cam, u, v, width, height, o, d);
                        beta = v3(1.f, 1.f, 1.f);
                        depth = S.max_depth;
                        has_ray = true;
                        if constexpr (LISTS) listed = round == 0 && s_lists[lst] != 0xFFFFFFFFu;
                    }
                }
                if (has_ray && !(LISTS && listed)) trav_begin<MODE, SCALED>(S, o, d, __builtin_inff(), stack, RENDER_BLOCK, T);
            }
            if constexpr (LISTS) {
                if (round == 1 || __ballot(listed) == 0ull) break;
                // the primary rays of the lanes that just started a sample of a listed pixel: closest hit over the
                // primitives of the pixel's candidate leaves
                if (listed) resolve_listed(lst, o, d, T);
            }
            }
        }
    }

    const unsigned long long rays = wave_sum((unsigned long long)n_rays);
    if (COUNT) {
        const unsigned long long nn = wave_sum((unsigned long long)cnt.nodes);
        const unsigned long long np = wave_sum((unsigned long long)cnt.prims);
        const unsigned long long wn = wave_sum((unsigned long long)cnt.wave_nodes);
        const unsigned long long wp = wave_sum((unsigned long long)cnt.wave_prims);
        const unsigned long long ws = wave_sum((unsigned long long)cnt.wave_shade);
        if (lane == 0) {
            atomicAdd(&counters[1], nn); atomicAdd(&counters[2], np);
            atomicAdd(&counters[3], wn); atomicAdd(&counters[4], wp); atomicAdd(&counters[5], ws);
            atomicAdd(&counters[12], slots_done); atomicAdd(&counters[13], slots_leaf);
        }
    }
    if (lane == 0) atomicAdd(&counters[0], rays);
}

// Workgroups of `kernel` a CU holds at once, asked of the runtime once per kernel and kept in `cache` (one per template
// instantiation; threads that race on the first call all store the same value).
template <typename Kernel>
static hipError_t resident_blocks_per_cu(Kernel kernel, std::atomic<int>& cache, int& out) {
    int n = cache.load(std::memory_order_relaxed);
    if (n == 0) {
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, RENDER_BLOCK, 0);
        if (e != hipSuccess) return e;
        n = n > 0 ? n : 1;
        cache.store(n, std::memory_order_relaxed);
    }
    out = n;
    return hipSuccess;
}

// max_blocks_per_cu: 0, or fewer resident workgroups per CU than the kernel allows (RenderTuning::blocks_per_cu).
template <int MODE, bool COUNT, bool PIPELINED, bool SCALED = false, bool SPLIT = false, bool LISTS = false, bool ROOMY = false>
static hipError_t launch_render_variant(const SceneDev& S, const RenderGeom& G, float* out, unsigned long long* d_counters,
                                        unsigned int* d_heads, int cu_count, uint32_t max_blocks_per_cu, hipStream_t stream) {
    // persistent grid: as many workgroups as the chip holds at once (never more than there are work items)
    static std::atomic<int> cache{0};
    int blocks_per_cu = 1;
    if (hipError_t e = resident_blocks_per_cu(k_render<MODE, COUNT, PIPELINED, SCALED, SPLIT, LISTS, ROOMY>, cache, blocks_per_cu); e != hipSuccess) return e;
    if (max_blocks_per_cu != 0u && (int)max_blocks_per_cu < blocks_per_cu) blocks_per_cu = (int)max_blocks_per_cu;
    unsigned long long blocks = (unsigned long long)blocks_per_cu * (unsigned)cu_count;
    const unsigned long long needed = ((unsigned long long)G.n_work + 3ull) / 4ull;
    if (blocks > needed) blocks = needed;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL((k_render<MODE, COUNT, PIPELINED, SCALED, SPLIT, LISTS, ROOMY>), dim3((unsigned)blocks), dim3(RENDER_BLOCK), 0, stream, S, G, out, d_counters, d_heads);
    return hipGetLastError();
}

// The instantiations for unrolled scenes (rrt_kernels_unrolled.hip).
hipError_t launch_render_unrolled(bool count_visits, bool pipelined, bool scaled, const SceneDev& S, const RenderGeom& G, float* out,
                                  unsigned long long* d_counters, unsigned int* d_heads, int cu_count, uint32_t max_blocks_per_cu,
                                  hipStream_t stream);
hipError_t launch_render_unrolled_lists(bool count_visits, bool pipelined, bool scaled, const SceneDev& S, const RenderGeom& G, float* out,
                                  unsigned long long* d_counters, unsigned int* d_heads, int cu_count, uint32_t max_blocks_per_cu,
                                  hipStream_t stream);

}  // namespace rrt
#endif
