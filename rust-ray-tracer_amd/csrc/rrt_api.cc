// rrt_api.cc -- implementation of the C ABI in include/rrt_ffi.h (context, scene upload, launches).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rrt_ffi.h"
#include "rrt_internal.h"
#include "lbvh_build.h"
#include "rrt_kernels.h"
#include "scene_build.h"

using namespace rrt;

namespace {
thread_local std::string g_thread_error;
}

namespace rrt {
void set_thread_error(const std::string& msg) { g_thread_error = msg; }
}  // namespace rrt

struct rrt_ctx {
    rrt_config cfg{};
    std::string err;
    bool has_scene = false;
    SceneDev sdev{};
    rrt_scene_info info{};
    // device buffers of the scene
    void* d_nodes = nullptr; void* d_tris = nullptr; void* d_spheres = nullptr; void* d_rects = nullptr;
    void* d_insts = nullptr; void* d_mats = nullptr; void* d_lights = nullptr;
    int cu_count = 1;
    unsigned long long* d_counters = nullptr;  // 256 B: rays, node visits, primitive tests; work-queue heads at +64 B
    float* d_partial = nullptr;                // per-chunk partial images of the last render geometry (grown on demand)
    size_t partial_bytes = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipStream_t last_stream = nullptr;
    bool render_pending = false;
    uint64_t pending_samples = 0;
    uint64_t pending_rays = 0;  // ray count of a pending ray batch (its kernel does not count rays itself)
};

namespace {

rrt_status set_err(rrt_ctx* ctx, rrt_status st, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else g_thread_error = msg;
    return st;
}
rrt_status hip_fail(rrt_ctx* ctx, hipError_t e, const char* what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    return set_err(ctx, e == hipErrorOutOfMemory ? RRT_ERR_OOM : RRT_ERR_HIP, m);
}
#define RRT_HIP(ctx, call)                                  \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

void free_scene(rrt_ctx* c) {
    void** bufs[] = {&c->d_nodes, &c->d_tris, &c->d_spheres, &c->d_rects, &c->d_insts, &c->d_mats, &c->d_lights};
    for (void** b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    c->has_scene = false;
}

template <typename T>
rrt_status upload(rrt_ctx* c, const std::vector<T>& v, void** dst, uint64_t& total) {
    // every array gets at least one record so that kernels never see a NULL base
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    RRT_HIP(c, hipMalloc(dst, bytes));
    RRT_HIP(c, hipMemset(*dst, 0, bytes));
    if (!v.empty()) RRT_HIP(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    total += bytes;
    return RRT_OK;
}

unsigned build_threads() {
    if (const char* e = std::getenv("RRT_BUILD_THREADS")) {
        int n = std::atoi(e);
        if (n > 0) return (unsigned)n;
    }
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    return std::min(hw, 16u);
}

}  // namespace

extern "C" {

uint32_t rrt_abi_version(void) { return RRT_ABI_VERSION; }

const char* rrt_last_error(const rrt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_thread_error.c_str(); }

rrt_status rrt_create(const rrt_config* cfg, rrt_ctx** out) {
    if (!cfg || !out) return set_err(nullptr, RRT_ERR_INVALID, "rrt_create: NULL argument");
    *out = nullptr;
    if (cfg->max_depth < 1 || cfg->max_depth > 64) return set_err(nullptr, RRT_ERR_INVALID, "rrt_create: max_depth must be in 1..64");
    if (!(cfg->t_min >= 0.0f)) return set_err(nullptr, RRT_ERR_INVALID, "rrt_create: t_min must be >= 0");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return set_err(nullptr, RRT_ERR_NO_DEVICE, "rrt_create: no HIP device available");
    if (cfg->device < 0 || cfg->device >= n) return set_err(nullptr, RRT_ERR_INVALID, "rrt_create: device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return set_err(nullptr, RRT_ERR_NO_DEVICE, "rrt_create: cannot query the device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(nullptr, RRT_ERR_NO_DEVICE, std::string("rrt_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    rrt_ctx* c = new (std::nothrow) rrt_ctx();
    if (!c) return set_err(nullptr, RRT_ERR_OOM, "rrt_create: out of host memory");
    c->cfg = *cfg;
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    if ((e = hipSetDevice(cfg->device)) != hipSuccess || (e = hipMalloc((void**)&c->d_counters, 256)) != hipSuccess ||
        (e = hipEventCreate(&c->ev_start)) != hipSuccess || (e = hipEventCreate(&c->ev_stop)) != hipSuccess) {
        rrt_status st = hip_fail(nullptr, e, "rrt_create");
        rrt_destroy(c);
        return st;
    }
    *out = c;
    return RRT_OK;
}

void rrt_destroy(rrt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    free_scene(c);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    delete c;
}

rrt_status rrt_scene_upload(rrt_ctx* c, const rrt_scene_desc* scene) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_scene_upload: ctx is NULL");
    if (!scene) return set_err(c, RRT_ERR_INVALID, "rrt_scene_upload: scene is NULL");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    // BVHs over at least this many primitives are built on the GPU when the context asks for it
    uint32_t device_min = 0;
    if (c->cfg.flags & RRT_CFG_DEVICE_BUILD) {
        device_min = 4096;
        if (const char* e = std::getenv("RRT_DEVICE_BUILD_MIN")) device_min = (uint32_t)std::max(2, std::atoi(e));
    }
    HostScene hs;
    std::string err;
    rrt_status st;
    uint64_t total = 0;
    float build_ms = 0.f, upload_ms = 0.f;
    for (;;) {
        const auto t0 = std::chrono::steady_clock::now();
        try {
            st = build_host_scene(scene, hs, err, build_threads(), device_min);
        } catch (const std::bad_alloc&) {
            return set_err(c, RRT_ERR_OOM, "rrt_scene_upload: out of host memory while building the BVH");
        } catch (const std::exception& ex) {
            return set_err(c, RRT_ERR_INVALID, std::string("rrt_scene_upload: ") + ex.what());
        }
        if (st != RRT_OK) return set_err(c, st, "rrt_scene_upload: " + err);
        const auto t1 = std::chrono::steady_clock::now();

        RRT_HIP(c, hipDeviceSynchronize());
        free_scene(c);
        total = 0;
        if ((st = upload(c, hs.qnodes, &c->d_nodes, total)) != RRT_OK || (st = upload(c, hs.tris, &c->d_tris, total)) != RRT_OK ||
            (st = upload(c, hs.spheres, &c->d_spheres, total)) != RRT_OK || (st = upload(c, hs.rects, &c->d_rects, total)) != RRT_OK ||
            (st = upload(c, hs.insts, &c->d_insts, total)) != RRT_OK || (st = upload(c, hs.mats, &c->d_mats, total)) != RRT_OK ||
            (st = upload(c, hs.lights, &c->d_lights, total)) != RRT_OK) {
            free_scene(c);
            return st;
        }
        const auto t2 = std::chrono::steady_clock::now();

        // the BVHs left to the GPU builder (none unless the context was created with RRT_CFG_DEVICE_BUILD)
        bool fits = true;
        for (const DeferredBuild& db : hs.deferred) {
            LbvhResult r;
            // a mesh may use every level the (already built) top level leaves free
            const uint32_t budget = db.mesh >= 0 && hs.top_depth > 0u && hs.top_depth + 2u <= MAX_TREE_DEPTH
                                        ? MAX_TREE_DEPTH - hs.top_depth - 1u : db.budget;
            const hipError_t e = lbvh_build(db.prims.data(), (uint32_t)db.prims.size(), db.grid, (QNode*)c->d_nodes, db.node_base,
                                            (TriRec*)c->d_tris, db.tri_base, db.tri_count, budget, &r);
            if (e != hipSuccess) {
                free_scene(c);
                return hip_fail(c, e, "rrt_scene_upload: GPU BVH build");
            }
            if (r.depth == 0u) { fits = false; break; }
            if (db.mesh < 0) hs.top_depth = r.depth;
            else hs.deepest_mesh = std::max(hs.deepest_mesh, r.depth);
        }
        if (fits && hs.top_depth + (hs.has_instances ? hs.deepest_mesh + 1u : 0u) > MAX_TREE_DEPTH) fits = false;
        const auto t3 = std::chrono::steady_clock::now();
        build_ms += std::chrono::duration<float, std::milli>(t1 - t0).count() + std::chrono::duration<float, std::milli>(t3 - t2).count();
        upload_ms += std::chrono::duration<float, std::milli>(t2 - t1).count();
        if (fits) break;
        // a linear BVH deeper than the traversal stack allows: build everything with the depth-bounded host builder
        if (device_min == 0u) {
            free_scene(c);
            return set_err(c, RRT_ERR_INVALID, "rrt_scene_upload: scene too deep for the traversal stack");
        }
        device_min = 0;
    }
    hs.depth = hs.top_depth + (hs.has_instances ? hs.deepest_mesh : 0u);
    c->info.device_built = (uint32_t)hs.deferred.size();

    SceneDev& S = c->sdev;
    S.nodes = (const QNode*)c->d_nodes; S.tris = (const TriRec*)c->d_tris; S.spheres = (const SphereRec*)c->d_spheres;
    S.rects = (const RectRec*)c->d_rects; S.insts = (const InstRec*)c->d_insts; S.mats = (const MatRec*)c->d_mats;
    S.lights = (const LightRec*)c->d_lights;
    S.light_count = (uint32_t)hs.lights.size();
    S.root = device_ref(hs.root);
    S.has_instances = hs.has_instances ? 1u : 0u;
    S.light_weight = S.light_count ? 1.0f / (float)S.light_count : 0.0f;
    S.light_zone = S.light_count ? (uint32_t)((0x100000000ull) % S.light_count) : 0u;
    S.cam = hs.cam;
    for (int a = 0; a < 3; ++a) S.background[a] = hs.background[a];
    S.t_min = c->cfg.t_min;
    S.max_depth = c->cfg.max_depth;
    S.seed = c->cfg.seed;
    S.grid = hs.grid;

    c->info.prim_count = hs.prim_count;
    c->info.node_count = (uint32_t)hs.nodes.size();
    c->info.tri_count = (uint32_t)hs.tris.size();
    c->info.bvh_depth = hs.depth;
    c->info.device_bytes = total;
    c->info.build_ms = build_ms;
    c->info.upload_ms = upload_ms;
    c->has_scene = true;
    return RRT_OK;
}

rrt_status rrt_scene_check(const rrt_scene_desc* scene, rrt_scene_info* out) {
    if (!scene || !out) return set_err(nullptr, RRT_ERR_INVALID, "rrt_scene_check: NULL argument");
    HostScene hs;
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    rrt_status st;
    try {
        st = build_host_scene(scene, hs, err, build_threads());
    } catch (const std::bad_alloc&) {
        return set_err(nullptr, RRT_ERR_OOM, "rrt_scene_check: out of host memory while building the BVH");
    } catch (const std::exception& ex) {
        return set_err(nullptr, RRT_ERR_INVALID, std::string("rrt_scene_check: ") + ex.what());
    }
    if (st != RRT_OK) return set_err(nullptr, st, "rrt_scene_check: " + err);
    std::memset(out, 0, sizeof *out);
    out->prim_count = hs.prim_count;
    out->node_count = (uint32_t)hs.nodes.size();
    out->tri_count = (uint32_t)hs.tris.size();
    out->bvh_depth = hs.depth;
    auto sz = [](size_t n, size_t rec) { return (uint64_t)(n ? n : 1) * rec; };
    out->device_bytes = sz(hs.qnodes.size(), sizeof(QNode)) + sz(hs.tris.size(), sizeof(TriRec)) + sz(hs.spheres.size(), sizeof(SphereRec)) +
                        sz(hs.rects.size(), sizeof(RectRec)) + sz(hs.insts.size(), sizeof(InstRec)) + sz(hs.mats.size(), sizeof(MatRec)) +
                        sz(hs.lights.size(), sizeof(LightRec));
    out->build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RRT_OK;
}

rrt_status rrt_scene_get_info(const rrt_ctx* c, rrt_scene_info* out) {
    if (!c || !out) return set_err(nullptr, RRT_ERR_INVALID, "rrt_scene_get_info: NULL argument");
    if (!c->has_scene) return set_err(const_cast<rrt_ctx*>(c), RRT_ERR_STATE, "rrt_scene_get_info: no scene uploaded");
    *out = c->info;
    return RRT_OK;
}

rrt_status rrt_intersect_device(rrt_ctx* c, const rrt_ray* d_rays, size_t n, float t_min, float t_max, rrt_hit* d_out,
                                void* stream) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_intersect_device: ctx is NULL");
    if (!c->has_scene) return set_err(c, RRT_ERR_STATE, "rrt_intersect_device: no scene uploaded");
    if (n == 0) return RRT_OK;
    if (!d_rays || !d_out) return set_err(c, RRT_ERR_INVALID, "rrt_intersect_device: NULL buffer");
    if (n > 0x7FFFFFFFull * 64ull) return set_err(c, RRT_ERR_INVALID, "rrt_intersect_device: batch too large");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    if (n > 0xF0000000ull) return set_err(c, RRT_ERR_INVALID, "rrt_intersect_device: batch too large");
    hipStream_t s = (hipStream_t)stream;
    RenderTuning tuning;
    if (const char* e = std::getenv("RRT_NODE_SHARE")) tuning.node_share_64ths = (uint32_t)std::max(0, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("RRT_REFILL_MIN")) tuning.refill_min = (uint32_t)std::max(1, std::min(64, std::atoi(e)));
    RRT_HIP(c, hipMemsetAsync(c->d_counters, 0, 256, s));
    RRT_HIP(c, hipEventRecord(c->ev_start, s));
    RRT_HIP(c, launch_intersect(c->sdev, d_rays, n, t_min, t_max, d_out, c->d_counters,
                                reinterpret_cast<unsigned int*>(c->d_counters + 8), c->cu_count,
                                (c->cfg.flags & RRT_CFG_COUNT_VISITS) != 0, tuning, s));
    RRT_HIP(c, hipEventRecord(c->ev_stop, s));
    c->last_stream = s;
    c->render_pending = true;
    c->pending_samples = 0;
    c->pending_rays = n;
    return RRT_OK;
}

rrt_status rrt_intersect(rrt_ctx* c, const rrt_ray* rays, size_t n, float t_min, float t_max, rrt_hit* out) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_intersect: ctx is NULL");
    if (!c->has_scene) return set_err(c, RRT_ERR_STATE, "rrt_intersect: no scene uploaded");
    if (n == 0) return RRT_OK;
    if (!rays || !out) return set_err(c, RRT_ERR_INVALID, "rrt_intersect: NULL buffer");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    rrt_ray* d_rays = nullptr;
    rrt_hit* d_hits = nullptr;
    hipError_t e = hipMalloc((void**)&d_rays, n * sizeof(rrt_ray));
    if (e == hipSuccess) e = hipMalloc((void**)&d_hits, n * sizeof(rrt_hit));
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, n * sizeof(rrt_ray), hipMemcpyHostToDevice);
    rrt_status st = RRT_OK;
    if (e == hipSuccess) st = rrt_intersect_device(c, d_rays, n, t_min, t_max, d_hits, nullptr);
    if (e == hipSuccess && st == RRT_OK) e = hipDeviceSynchronize();
    if (e == hipSuccess && st == RRT_OK) e = hipMemcpy(out, d_hits, n * sizeof(rrt_hit), hipMemcpyDeviceToHost);
    if (d_rays) (void)hipFree(d_rays);
    if (d_hits) (void)hipFree(d_hits);
    if (e != hipSuccess) return hip_fail(c, e, "rrt_intersect");
    return st;
}

rrt_status rrt_render_device(rrt_ctx* c, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                             float* d_accum, void* stream) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_render_device: ctx is NULL");
    if (!c->has_scene) return set_err(c, RRT_ERR_STATE, "rrt_render_device: no scene uploaded");
    if (!d_accum) return set_err(c, RRT_ERR_INVALID, "rrt_render_device: accum is NULL");
    if (width == 0 || height == 0 || spp == 0) return set_err(c, RRT_ERR_INVALID, "rrt_render_device: width, height and spp must be > 0");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return set_err(c, RRT_ERR_INVALID, "rrt_render_device: image too large");
    if ((uint64_t)spp + spp_offset > 0xFFFFFFFFull) return set_err(c, RRT_ERR_INVALID, "rrt_render_device: sample index overflows");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    // Scenes with a real hierarchy (>= 4096 nodes): paths bounce and walks are long, so items of 8 samples, the
    // straggler rule and lanes that move on to the next item by themselves pay; on tiny scenes each of them costs.
    const bool big_scene = c->info.node_count >= 4096u;
    uint32_t n_chunks = render_chunks(width, height, spp, big_scene ? 8u : 16u);
    if (const char* e = std::getenv("RRT_CHUNKS")) {
        const int v = std::atoi(e);
        if (v >= 1) n_chunks = std::min<uint32_t>((uint32_t)v, spp);
    }
    while (n_chunks > 1u && (uint64_t)((width + 7u) / 8u) * ((height + 7u) / 8u) * n_chunks > 0x7FFFFFFFull) n_chunks /= 2u;
    const size_t need = n_chunks > 1u ? (size_t)n_chunks * width * height * 3u * sizeof(float) : 0u;
    if (need > c->partial_bytes) {
        // the old workspace may still be read by a render in flight on another stream
        RRT_HIP(c, hipDeviceSynchronize());
        if (c->d_partial) (void)hipFree(c->d_partial);
        c->d_partial = nullptr;
        c->partial_bytes = 0;
        RRT_HIP(c, hipMalloc((void**)&c->d_partial, need));
        c->partial_bytes = need;
    }
    // tuning overrides for experiments: RRT_CHUNKS (above), RRT_NODE_SHARE (0..64)
    RenderTuning tuning;
    if (const char* e = std::getenv("RRT_NODE_SHARE")) tuning.node_share_64ths = (uint32_t)std::max(0, std::min(64, std::atoi(e)));
    if (!big_scene) {
        tuning.leave_walk = 0u;
        tuning.overlap_items = 0u;
        tuning.band_tiles = 0u;
    }
    if (const char* e = std::getenv("RRT_LEAVE_WALK")) tuning.leave_walk = (uint32_t)std::max(0, std::min(63, std::atoi(e)));
    if (const char* e = std::getenv("RRT_LEAVE_READY")) tuning.leave_ready = (uint32_t)std::max(1, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("RRT_OVERLAP_ITEMS")) tuning.overlap_items = std::atoi(e) != 0 ? 1u : 0u;
    if (const char* e = std::getenv("RRT_BAND_TILES")) tuning.band_tiles = (uint32_t)std::max(0, std::atoi(e));
    RRT_HIP(c, hipMemsetAsync(c->d_counters, 0, 256, s));
    RRT_HIP(c, hipEventRecord(c->ev_start, s));
    RRT_HIP(c, launch_render(c->sdev, width, height, spp, spp_offset, d_accum, c->d_partial, n_chunks, c->d_counters,
                             reinterpret_cast<unsigned int*>(c->d_counters + 8), c->cu_count,
                             (c->cfg.flags & RRT_CFG_COUNT_VISITS) != 0, tuning, s));
    RRT_HIP(c, hipEventRecord(c->ev_stop, s));
    c->last_stream = s;
    c->render_pending = true;
    c->pending_samples = (uint64_t)width * height * spp;
    c->pending_rays = 0;
    return RRT_OK;
}

rrt_status rrt_render_stats(rrt_ctx* c, rrt_stats* stats) {
    if (!c || !stats) return set_err(c, RRT_ERR_INVALID, "rrt_render_stats: NULL argument");
    if (!c->render_pending) return set_err(c, RRT_ERR_STATE, "rrt_render_stats: no render was issued");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    RRT_HIP(c, hipEventSynchronize(c->ev_stop));
    unsigned long long h[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [8..11] are the work-queue heads
    RRT_HIP(c, hipMemcpy(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
    float ms = 0.f;
    RRT_HIP(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
    std::memset(stats, 0, sizeof *stats);
    stats->rays = c->pending_rays ? c->pending_rays : h[0];
    stats->node_visits = h[1];
    stats->prim_tests = h[2];
    stats->node_wave_steps = h[3];
    stats->prim_wave_steps = h[4];
    stats->shade_wave_passes = h[5];
    stats->refill_wave_passes = h[6];
    stats->node_slots_lane_done = h[12];
    stats->node_slots_lane_at_leaf = h[13];
    stats->samples = c->pending_samples;
    stats->kernel_ms = ms;
    return RRT_OK;
}

rrt_status rrt_render(rrt_ctx* c, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset, float* accum,
                      rrt_stats* stats) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_render: ctx is NULL");
    if (!accum) return set_err(c, RRT_ERR_INVALID, "rrt_render: accum is NULL");
    if (width == 0 || height == 0) return set_err(c, RRT_ERR_INVALID, "rrt_render: width and height must be > 0");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    float* d_accum = nullptr;
    RRT_HIP(c, hipMalloc((void**)&d_accum, bytes));
    rrt_status st = rrt_render_device(c, width, height, spp, spp_offset, d_accum, nullptr);
    rrt_stats local;
    if (st == RRT_OK) st = rrt_render_stats(c, &local);
    if (st == RRT_OK) {
        hipError_t e = hipMemcpy(accum, d_accum, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = hip_fail(c, e, "rrt_render: copy back");
    }
    (void)hipFree(d_accum);
    if (st == RRT_OK && stats) *stats = local;
    return st;
}

rrt_status rrt_tonemap_device(rrt_ctx* c, const float* d_accum, uint32_t width, uint32_t height, uint32_t spp_total,
                              uint8_t* d_rgb8, void* stream) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_tonemap_device: ctx is NULL");
    if (!d_accum || !d_rgb8) return set_err(c, RRT_ERR_INVALID, "rrt_tonemap_device: NULL buffer");
    if (width == 0 || height == 0 || spp_total == 0) return set_err(c, RRT_ERR_INVALID, "rrt_tonemap_device: width, height and spp must be > 0");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return set_err(c, RRT_ERR_INVALID, "rrt_tonemap_device: image too large");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    RRT_HIP(c, launch_tonemap(d_accum, width, height, spp_total, d_rgb8, (hipStream_t)stream));
    return RRT_OK;
}

rrt_status rrt_tonemap(rrt_ctx* c, const float* accum, uint32_t width, uint32_t height, uint32_t spp_total,
                       uint8_t* rgb8) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_tonemap: ctx is NULL");
    if (!accum || !rgb8) return set_err(c, RRT_ERR_INVALID, "rrt_tonemap: NULL buffer");
    if (width == 0 || height == 0) return set_err(c, RRT_ERR_INVALID, "rrt_tonemap: width and height must be > 0");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    const size_t n = (size_t)width * height * 3;
    float* d_accum = nullptr;
    uint8_t* d_rgb = nullptr;
    hipError_t e = hipMalloc((void**)&d_accum, n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&d_rgb, n);
    if (e == hipSuccess) e = hipMemcpy(d_accum, accum, n * sizeof(float), hipMemcpyHostToDevice);
    rrt_status st = RRT_OK;
    if (e == hipSuccess) st = rrt_tonemap_device(c, d_accum, width, height, spp_total, d_rgb, nullptr);
    if (e == hipSuccess && st == RRT_OK) e = hipDeviceSynchronize();
    if (e == hipSuccess && st == RRT_OK) e = hipMemcpy(rgb8, d_rgb, n, hipMemcpyDeviceToHost);
    if (d_accum) (void)hipFree(d_accum);
    if (d_rgb) (void)hipFree(d_rgb);
    if (e != hipSuccess) return hip_fail(c, e, "rrt_tonemap");
    return st;
}

rrt_status rrt_selftest_arith(rrt_ctx* c, uint64_t* recip_mismatches, uint64_t* sqrt_mismatches) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_selftest_arith: ctx is NULL");
    if (!recip_mismatches || !sqrt_mismatches) return set_err(c, RRT_ERR_INVALID, "rrt_selftest_arith: NULL output");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    unsigned long long* d_bad = nullptr;
    unsigned long long bad[2] = {0ull, 0ull};
    hipError_t e = hipMalloc((void**)&d_bad, sizeof bad);
    if (e == hipSuccess) e = hipMemset(d_bad, 0, sizeof bad);
    if (e == hipSuccess) e = launch_check_arith(d_bad, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
    if (d_bad) (void)hipFree(d_bad);
    if (e != hipSuccess) return hip_fail(c, e, "rrt_selftest_arith");
    *recip_mismatches = bad[0];
    *sqrt_mismatches = bad[1];
    return RRT_OK;
}

}  // extern "C"
