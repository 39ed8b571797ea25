// rrt_kernels.hip -- gfx950 kernels: persistent path-tracing megakernel, ray-batch closest hit, tone map.
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
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include "rrt_device.h"
#include "rrt_kernels.h"

namespace rrt {

static constexpr int WAVE = 64;
#ifndef RRT_TILE_W_LOG2
#define RRT_TILE_W_LOG2 3  // a work item's 64 pixels form a tile of 2^RRT_TILE_W_LOG2 pixels' width (8x8 by default)
#endif
static constexpr uint32_t TILE_W = 1u << RRT_TILE_W_LOG2, TILE_H = 64u / TILE_W;
static constexpr int RENDER_BLOCK = 256;  // 4 independent wavefronts per workgroup
static constexpr uint32_t TRACE_BLOCK = 512;  // ray indices a wavefront of the streaming kernel draws at a time
static constexpr int RENDER_STACK = 16;   // traversal stack entries per lane the render kernel keeps in LDS (short stack)
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
#ifndef RRT_FLAT_WAVES
#define RRT_FLAT_WAVES 7  // workgroups per CU the pipelined form is compiled for on scenes without instances
#endif
#ifndef RRT_ONE_ITEM_WAVES
#define RRT_ONE_ITEM_WAVES 6  // workgroups per CU the one-item form (small scenes) is compiled for
#endif
// SCALED: the node steps use the scaled form of the slab arithmetic (rrt_device.h: trav_set_ray_scaled), which the host
// selects when the scene is walked on one grid and its camera's ray directions are of ordinary magnitude.
template <int MODE, bool COUNT, bool PIPELINED, bool SCALED>
__global__ __launch_bounds__(RENDER_BLOCK, MODE == MODE_TWO_LEVEL ? RRT_INST_WAVES : (PIPELINED ? (MODE == MODE_UNROLLED ? RRT_UNROLLED_WAVES : RRT_FLAT_WAVES) : RRT_ONE_ITEM_WAVES)) void k_render(SceneDev S, RenderGeom G, float* __restrict__ out,
                                                         unsigned long long* __restrict__ counters,
                                                         unsigned int* __restrict__ heads) {
    constexpr bool INST = MODE == MODE_TWO_LEVEL;
    static_assert(!(SCALED && INST), "the scaled slab arithmetic is for scenes walked on one grid");
    // RENDER_STACK entries per lane of the traversal stack live in LDS, the rest in private memory (six workgroups per
    // CU, seven in the pipelined form, whose large scenes gain more from the extra wavefront than they lose to its
    // spills).  Instanced scenes add six floats per lane for the world-space ray of a lane that walks inside a mesh.
    constexpr int K = INST ? RRT_INST_STACK : RENDER_STACK;
    __shared__ int s_stack[K * RENDER_BLOCK];
    __shared__ float s_wray[INST ? 6 * RENDER_BLOCK : 1];
    __shared__ uint32_t s_items[(RENDER_BLOCK / WAVE) * 16];  // per wavefront: the records of its two work items
    // Unrolled scenes: the instance transforms a triangle leaf needs (offset, sin, cos: the first 32 bytes of a record)
    // are copied to LDS once per workgroup when they fit, slot 0 = the identity record in front of the table.
    __shared__ vec4f s_xf[MODE == MODE_UNROLLED ? 2 * XF_LDS_SLOTS : 1];
    const lds_float4* xf_lds = nullptr;
    if (MODE == MODE_UNROLLED && S.inst_count + 1u <= (uint32_t)XF_LDS_SLOTS) {
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

    if (!PIPELINED) {
        // One item at a time: the lanes start an item together and wait for its last path.  The form for
        // scenes whose trips are short (a handful of node steps, mostly first-ray misses), where the
        // bookkeeping of the pipelined form below costs more than the waiting it removes.
        for (;;) {
            const uint32_t work = draw();
            if (work == NO_WORK) break;  // every queue is empty: this wave is done
            const uint32_t tile = work / G.n_chunks, chunk = work % G.n_chunks;
            uint32_t tx, ty;
            tile_coords(G, tile, tx, ty);
            const uint32_t x = tx * TILE_W + (lane & (TILE_W - 1u)), y = ty * TILE_H + (lane >> RRT_TILE_W_LOG2);
            const bool inside = x < width && y < height;
            // samples [first, first + spp) of this chunk, as absolute sample indices
            const uint32_t first = chunk * G.chunk_spp;
            const uint32_t spp = first < G.spp ? min(G.chunk_spp, G.spp - first) : 0u;
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
            if (has_ray) {
                rng = rng_init(S.seed, pixel, spp_offset + s);
                const float u = (float)x + rng_f32(rng);  // main.rs:92-93
                const float v = (float)y + rng_f32(rng);
                rng_skip(rng);                                // camera.rs:67: the ray time draw (unused by static scenes)
                camera_ray(S.cam, u, v, width, height, o, d);
                beta = v3(1.f, 1.f, 1.f);
                depth = S.max_depth;
                trav_begin<MODE, SCALED>(S, o, d, __builtin_inff(), stack, RENDER_BLOCK, T);
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
                // ---- shade phase: every lane whose walk is complete shades its hit and starts its next ray
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
                        } else {
                            has_ray = false;
                        }
                    }
                    if (has_ray) trav_begin<MODE, SCALED>(S, o, d, __builtin_inff(), stack, RENDER_BLOCK, T);
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
        uint32_t* items = s_items + (threadIdx.x >> 6) * 16u;
        uint32_t cur_slot = 0u;    // record of the current item; the next item's is the other one
        bool next_valid = false;
        auto draw_into = [&](uint32_t slot) -> bool {
            const uint32_t work = draw();
            uint32_t x0 = 0u, y0 = 0u, chunk = 0u, n_spp = 0u, first_sample = 0u;
            if (work != NO_WORK) {
                const uint32_t tile = work / G.n_chunks;
                chunk = work % G.n_chunks;
                const uint32_t first = chunk * G.chunk_spp;  // samples [first, first + n_spp) as absolute sample indices
                uint32_t tx, ty;
                tile_coords(G, tile, tx, ty);
                x0 = tx * TILE_W;
                y0 = ty * TILE_H;
                n_spp = first < G.spp ? min(G.chunk_spp, G.spp - first) : 0u;
                first_sample = G.spp_offset + first;
            }
            if (lane == 0u) {
                uint32_t* rec = items + slot * 8u;
                rec[0] = x0; rec[1] = y0; rec[2] = chunk; rec[3] = n_spp; rec[4] = first_sample;
            }
            return work != NO_WORK;
        };
        draw_into(0u);
        next_valid = draw_into(1u);

        // per lane: `stage` = which of the two items it is on (0: current, 1: next, 2: done with both), + 4 once it
        // has taken its pixel of that item; then its place in the item
        uint32_t stage = 0u;
        // (kept small: the kernel runs at the register budget of six waves per SIMD)
        uint32_t xy = 0;           // the lane's pixel, x | y << 16
        uint32_t next_sample = 0;  // absolute index of the next sample to start
        uint32_t left = 0;         // samples of the item not yet finished
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
                if (!next_valid) {
                    out_of_work = true;  // nobody on the current item and no next item: every lane is done
                    break;
                }
                stage--;
                next_valid = draw_into(cur_slot);  // the old current item's record is free
                cur_slot ^= 1u;
            }
            if (out_of_work) break;

#if RRT_WALK_PRIO || RRT_SHADE_PRIO
            __builtin_amdgcn_s_setprio(RRT_WALK_PRIO);
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
                        trav_nodes<MODE, COUNT, K, RRT_PIPE_STEPS, SCALED>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, share, lim, deep, wray);
                        uint32_t steps = cnt.nodes - nodes0;
                        for (int off = 32; off > 0; off >>= 1) steps = max(steps, (uint32_t)__shfl_xor((int)steps, off, WAVE));
                        if (lane == 0u) {
                            slots_done += (unsigned long long)steps * (64u - n_walk);
                            slots_leaf += (unsigned long long)steps * (n_walk - n_desc);
                        }
                    } else {
                        trav_nodes<MODE, COUNT, K, RRT_PIPE_STEPS, SCALED>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, share, lim, deep, wray);
                    }
                }
                if (T.cur < 0 && T.cur != REF_DONE && !(INST && trav_due(T.cur))) trav_leaf<MODE, COUNT, K, true>(S, o, d, S.t_min, RENDER_BLOCK, T, cnt, lim, deep, wray, xf_lds);
            }

#if RRT_WALK_PRIO || RRT_SHADE_PRIO
            __builtin_amdgcn_s_setprio(RRT_SHADE_PRIO);
#endif
            // ---- shade phase: every lane whose walk is complete shades its hit; a lane whose path ended starts its
            // next sample, a lane whose item ended (or that has none yet) takes its pixel of the item it is due for
            const bool shade_now = has_ray && T.cur == REF_DONE;
            bool take = stage == 0u || stage == 1u;  // on an item without having taken its pixel of it
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
                            const uint32_t chunk = items[(on == 0u ? cur_slot : cur_slot ^ 1u) * 8u + 2u];
                            const uint32_t pixel = (xy >> 16) * width + (xy & 0xFFFFu);
                            float* px = out + 3ull * ((size_t)chunk * width * height + pixel);
                            px[0] = sum.x; px[1] = sum.y; px[2] = sum.z;
                            has_ray = false;
                            start = false;
                            stage = on + 1u;
                            take = stage == 1u;
                        }
                    }
                    if (take) {
                        const uint32_t* rec = items + (stage == 0u ? cur_slot : cur_slot ^ 1u) * 8u;
                        const uint32_t px_x = rec[0] + (lane & (TILE_W - 1u)), px_y = rec[1] + (lane >> RRT_TILE_W_LOG2);
                        const uint32_t n_spp = rec[3];
                        if (px_x < width && px_y < height && n_spp != 0u) {
                            xy = px_x | (px_y << 16);
                            left = n_spp;
                            next_sample = rec[4];
                            sum = v3(0.f, 0.f, 0.f);
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
                    }
                }
                if (has_ray) trav_begin<MODE, SCALED>(S, o, d, __builtin_inff(), stack, RENDER_BLOCK, T);
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

template <int MODE, bool COUNT, bool PIPELINED, bool SCALED = false>
static hipError_t launch_render_variant(const SceneDev& S, const RenderGeom& G, float* out, unsigned long long* d_counters,
                                        unsigned int* d_heads, int cu_count, hipStream_t stream) {
    // persistent grid: as many workgroups as the chip holds at once (never more than there are work items)
    static int blocks_per_cu = 0;  // per template instantiation
    if (blocks_per_cu == 0) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_render<MODE, COUNT, PIPELINED, SCALED>, RENDER_BLOCK, 0);
        if (e != hipSuccess) return e;
        blocks_per_cu = n > 0 ? n : 1;
        // experiment knob: fewer resident workgroups per CU (occupancy sensitivity)
        if (const char* env = getenv("RRT_BLOCKS_PER_CU")) {
            const int v = atoi(env);
            if (v > 0 && v < blocks_per_cu) blocks_per_cu = v;
        }
    }
    unsigned long long blocks = (unsigned long long)blocks_per_cu * (unsigned)cu_count;
    const unsigned long long needed = ((unsigned long long)G.n_work + 3ull) / 4ull;
    if (blocks > needed) blocks = needed;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL((k_render<MODE, COUNT, PIPELINED, SCALED>), dim3((unsigned)blocks), dim3(RENDER_BLOCK), 0, stream, S, G, out, d_counters, d_heads);
    return hipGetLastError();
}

hipError_t launch_render(const SceneDev& S, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                         float* d_accum, float* d_partial, uint32_t n_chunks, unsigned long long* d_counters,
                         unsigned int* d_heads, int cu_count, bool count_visits, const RenderTuning& tuning,
                         hipStream_t stream) {
    RenderGeom G;
    G.width = width; G.height = height; G.spp = spp; G.spp_offset = spp_offset;
    G.tiles_x = (width + TILE_W - 1u) / TILE_W;
    G.n_tiles = G.tiles_x * ((height + TILE_H - 1u) / TILE_H);
    G.band_tiles = tuning.band_tiles == 0u ? G.tiles_x : std::min(tuning.band_tiles, G.tiles_x);
    G.band_items = G.band_tiles * ((height + TILE_H - 1u) / TILE_H);
    G.full_bands = G.tiles_x / G.band_tiles;
    G.n_chunks = n_chunks;
    G.chunk_spp = (spp + n_chunks - 1u) / n_chunks;
    G.n_work = G.n_tiles * G.n_chunks;
    G.node_share_64ths = tuning.node_share_64ths;
    G.leave_walk = tuning.leave_walk;
    G.leave_ready = tuning.leave_ready;
    float* out = n_chunks > 1u ? d_partial : d_accum;
    const int mode = S.unroll_shift != 0u ? MODE_UNROLLED : (S.has_instances != 0u ? MODE_TWO_LEVEL : MODE_FLAT);
    hipError_t e;
    // the pipelined form packs a lane's pixel coordinates into 16 bits each
    const bool pipelined = tuning.overlap_items != 0u && width < 65536u && height < 65536u;
    // the scaled slab arithmetic where the host found it valid for this scene (S.sigma != 0) and the walk stays on one grid
    const bool scaled = S.sigma != 0.0f && tuning.scaled_slab != 0u && mode != MODE_TWO_LEVEL;
    const unsigned variant = (scaled ? 16u : 0u) | ((unsigned)mode << 2) | (count_visits ? 2u : 0u) | (pipelined ? 1u : 0u);
    switch (variant) {
        case 27u: e = launch_render_variant<MODE_UNROLLED, true, true, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 26u: e = launch_render_variant<MODE_UNROLLED, true, false, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 25u: e = launch_render_variant<MODE_UNROLLED, false, true, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 24u: e = launch_render_variant<MODE_UNROLLED, false, false, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 19u: e = launch_render_variant<MODE_FLAT, true, true, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 18u: e = launch_render_variant<MODE_FLAT, true, false, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 17u: e = launch_render_variant<MODE_FLAT, false, true, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 16u: e = launch_render_variant<MODE_FLAT, false, false, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 11u: e = launch_render_variant<MODE_UNROLLED, true, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 10u: e = launch_render_variant<MODE_UNROLLED, true, false>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 9u: e = launch_render_variant<MODE_UNROLLED, false, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 8u: e = launch_render_variant<MODE_UNROLLED, false, false>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 7u: e = launch_render_variant<MODE_TWO_LEVEL, true, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 6u: e = launch_render_variant<MODE_TWO_LEVEL, true, false>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 5u: e = launch_render_variant<MODE_TWO_LEVEL, false, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 4u: e = launch_render_variant<MODE_TWO_LEVEL, false, false>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 3u: e = launch_render_variant<MODE_FLAT, true, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 2u: e = launch_render_variant<MODE_FLAT, true, false>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        case 1u: e = launch_render_variant<MODE_FLAT, false, true>(S, G, out, d_counters, d_heads, cu_count, stream); break;
        default: e = launch_render_variant<MODE_FLAT, false, false>(S, G, out, d_counters, d_heads, cu_count, stream); break;
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
    static int blocks_per_cu = 0;  // per template instantiation
    if (blocks_per_cu == 0) {
        int k = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&k, k_trace_stream<MODE, COUNT>, RENDER_BLOCK, 0);
        if (e != hipSuccess) return e;
        blocks_per_cu = k > 0 ? k : 1;
    }
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
    if (getenv("RRT_INTERSECT_STATIC")) {  // experiment knob: the plain one-lane-per-ray kernel
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
