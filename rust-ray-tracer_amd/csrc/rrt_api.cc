// rrt_api.cc -- implementation of the C ABI in include/rrt_ffi.h (context, scene upload, launches).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is opened at run time, when a group spans several GPUs

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
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

// Static workspace of rrt_render_device_noalloc: the counter block (256 bytes: ray and visit counters, work-queue
// heads) followed by the per-chunk partial images.  Every GPU the library runs on holds its own copy of the array at
// its own address until the process ends; host code never uses the array's name as an address, only the pointer
// hipGetSymbolAddress returns for the current device (looked up once per device in rrt_create).
static constexpr size_t NOALLOC_COUNTER_BYTES = 256;
static constexpr size_t NOALLOC_PARTIAL_BYTES = (size_t)RRT_NOALLOC_PARTIAL_MAX_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_COUNTER_BYTES + NOALLOC_PARTIAL_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;                      // contexts may be created from several host threads at once
unsigned char* g_scratch_ptr[MAX_DEVICES] = {};  // per device ordinal: address of that device's `scratch`

// Address of the CURRENT device's copy of `scratch` (the caller has made `device` current); asked of the runtime once
// per device.
hipError_t static_workspace(int device, unsigned char** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        unsigned char* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

// The scheduling knobs of a new context: the library's own rules (RRT_TUNE_AUTO) unless the environment names a value.
// Read once per rrt_create; the render and intersect entry points only look at the context's copy.
uint32_t env_u32(const char* name, uint32_t lo, uint32_t hi) {
    const char* e = std::getenv(name);
    if (!e || !*e) return RRT_TUNE_AUTO;
    const long long v = std::atoll(e);
    return (uint32_t)std::max<long long>(lo, std::min<long long>(hi, v));
}
rrt_tuning tuning_from_env() {
    rrt_tuning t;
    t.chunks = env_u32("RRT_CHUNKS", 1u, 0x7FFFFFFFu);
    t.split_items = env_u32("RRT_SPLIT_ITEMS", 0u, 1u);
    t.node_share_64ths = env_u32("RRT_NODE_SHARE", 0u, 64u);
    t.leave_walk = env_u32("RRT_LEAVE_WALK", 0u, 63u);
    t.leave_ready = env_u32("RRT_LEAVE_READY", 1u, 64u);
    t.overlap_items = env_u32("RRT_OVERLAP_ITEMS", 0u, 1u);
    t.band_tiles = env_u32("RRT_BAND_TILES", 0u, 0x7FFFFFFFu);
    t.scaled_slab = env_u32("RRT_SCALED_SLAB", 0u, 1u);
    t.refill_min = env_u32("RRT_REFILL_MIN", 1u, 64u);
    t.blocks_per_cu = env_u32("RRT_BLOCKS_PER_CU", 1u, 64u);
    t.intersect_static = env_u32("RRT_INTERSECT_STATIC", 0u, 1u);
    t.pixel_lists = env_u32("RRT_PIXEL_LISTS", 0u, 1u);
    t.pixel_list_len = env_u32("RRT_PIXEL_LIST_LEN", 1u, PIXEL_LIST_MAX);
    t.list_waves = env_u32("RRT_LIST_WAVES", 6u, 7u);
    for (uint32_t& r : t.reserved) r = RRT_TUNE_AUTO;
    return t;
}
rrt_tuning clamp_tuning(rrt_tuning t) {
    auto cl = [](uint32_t& v, uint32_t lo, uint32_t hi) { if (v != RRT_TUNE_AUTO) v = std::max(lo, std::min(hi, v)); };
    cl(t.chunks, 1u, 0x7FFFFFFFu); cl(t.split_items, 0u, 1u); cl(t.node_share_64ths, 0u, 64u); cl(t.leave_walk, 0u, 63u);
    cl(t.leave_ready, 1u, 64u); cl(t.overlap_items, 0u, 1u); cl(t.band_tiles, 0u, 0x7FFFFFFFu); cl(t.scaled_slab, 0u, 1u);
    cl(t.refill_min, 1u, 64u); cl(t.blocks_per_cu, 1u, 64u); cl(t.intersect_static, 0u, 1u); cl(t.pixel_lists, 0u, 1u); cl(t.pixel_list_len, 1u, PIXEL_LIST_MAX); cl(t.list_waves, 6u, 7u);
    for (uint32_t& r : t.reserved) r = RRT_TUNE_AUTO;
    return t;
}
}  // namespace

struct rrt_ctx {
    rrt_config cfg{};
    rrt_tuning tune{};  // scheduling knobs: the environment as rrt_create found it, or rrt_debug_set_tuning
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
    uint32_t* d_lists = nullptr;               // per-pixel candidate leaf lists of the last render geometry (grown on demand)
    size_t lists_bytes = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipStream_t last_stream = nullptr;
    unsigned long long* last_counters = nullptr;  // the counter block the pending call used (d_counters, or the static one)
    unsigned char* d_static = nullptr;            // this device's copy of the static workspace
    // grow-only staging buffers of the host-pointer variants (rrt_render, rrt_intersect, rrt_tonemap)
    void* d_stage[2] = {nullptr, nullptr};
    size_t stage_bytes[2] = {0, 0};
    bool render_pending = false;
    uint64_t pending_samples = 0;
    uint64_t pending_rays = 0;  // ray count of a pending ray batch (its kernel does not count rays itself)
    uint32_t pending_chunks = 0;  // sample chunks of the pending render
    uint64_t pending_list_pixels = 0;  // pixels of the pending render if it collects per-pixel leaf lists, else 0
    // share of the pixels that got a list in the last such render whose statistics were read (1.0: not known yet);
    // reset by a scene upload
    float list_share = 1.0f;
    // The render kernel with lists exists compiled for 7 and for 6 wavefronts per SIMD, and which is faster depends on
    // the scene: the launches of one frame geometry (width, height, samples) run 7 (the first launch, which also
    // warms the caches: not timed), then 6, then 7 again, and from then on the one whose launch was faster.  A launch
    // counts once its statistics (the kernel time) are read.  [0]: 7, [1]: 6; 0: not measured.  Reset by a scene upload
    // and by another frame geometry.
    float waves_ms[2] = {0.0f, 0.0f};
    uint32_t waves_key[3] = {0u, 0u, 0u};
    uint32_t waves_launches = 0;  // launches of this geometry whose time was read
    int pending_waves = -1;  // which of the two the pending render runs (index into waves_ms), -1: neither
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

// The instance records as the device holds them: an identity record in front (SceneDev::insts points behind it), which
// is what a triangle outside every instance reads in an unrolled scene.
std::vector<InstRec> insts_with_identity(const std::vector<InstRec>& insts) {
    std::vector<InstRec> v(insts.size() + 1u);
    std::memset(&v[0], 0, sizeof(InstRec));
    v[0].cos_t = 1.0f;
    std::copy(insts.begin(), insts.end(), v.begin() + 1);
    return v;
}

// `tail`: records behind v that exist on the device only (zeroed here, written by the GPU builder).
template <typename T>
rrt_status upload(rrt_ctx* c, const std::vector<T>& v, void** dst, uint64_t& total, size_t tail = 0) {
    // every array gets at least one record so that kernels never see a NULL base
    const size_t bytes = std::max<size_t>(v.size() + tail, 1) * sizeof(T);
    RRT_HIP(c, hipMalloc(dst, bytes));
    RRT_HIP(c, hipMemset(*dst, 0, bytes));
    if (!v.empty()) RRT_HIP(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    total += bytes;
    return RRT_OK;
}

// Grow-only device staging buffer `slot` of the host-pointer entry points: allocated on first use, reused afterwards,
// so that repeated rrt_render / rrt_intersect / rrt_tonemap calls do not pay a hipMalloc / hipFree pair each.
rrt_status stage(rrt_ctx* c, int slot, size_t bytes, void** out) {
    if (bytes > c->stage_bytes[slot]) {
        RRT_HIP(c, hipDeviceSynchronize());  // the old buffer may still be in use by a call in flight
        if (c->d_stage[slot]) (void)hipFree(c->d_stage[slot]);
        c->d_stage[slot] = nullptr;
        c->stage_bytes[slot] = 0;
        RRT_HIP(c, hipMalloc(&c->d_stage[slot], bytes));
        c->stage_bytes[slot] = bytes;
    }
    *out = c->d_stage[slot];
    return RRT_OK;
}

// A device call on stream `s` shares the context's counter block, work-queue heads, partial images and events with the
// previous one: if that one was issued on another stream and may still be running, `s` waits for it first.
rrt_status order_after_pending(rrt_ctx* c, hipStream_t s) {
    if (c->render_pending && s != c->last_stream) RRT_HIP(c, hipStreamWaitEvent(s, c->ev_stop, 0));
    return RRT_OK;
}

// Scale of the render kernels' scaled slab arithmetic (rrt_device.h: trav_set_ray_scaled), or sigma = 0 where that form
// is not valid: it needs a grid whose steps are finite, positive and within 2^20 of each other, a scale that is a
// normal float with room to spare, and camera rays whose direction components stay far inside [2^-40, 2^40] (checked
// at nine points of the frame against [2^-24, 2^24]; scattered directions have length 1 or at most 2).
void set_slab_scale(SceneDev& S) {
    S.sigma = 0.0f;
    for (int a = 0; a < 3; ++a) S.step_scaled[a] = 0.0f;
    const float* st = S.grid.step;
    const float step_max = std::max(st[0], std::max(st[1], st[2])), step_min = std::min(st[0], std::min(st[1], st[2]));
    if (!(step_min > 0.0f) || !std::isfinite(step_max) || !(step_min >= step_max * 0x1p-20f)) return;
    int e = 0;
    (void)std::frexp(step_max, &e);  // step_max = m * 2^e, m in [0.5, 1)
    const int se = 49 - e;           // step_max * 2^se lies in [2^48, 2^49)
    if (se - 149 < -120 || se - 149 > -40) return;  // sigma = 2^(se - 149): a scene extent between 2^-44 and 2^36
    const CameraRec& cam = S.cam;
    for (int iy = -1; iy <= 1; ++iy)
        for (int ix = -1; ix <= 1; ++ix) {
            const double px = ix * (double)cam.scale * cam.aspect, py = iy * (double)cam.scale;
            double len2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double d = cam.c0[a] * px + cam.c1[a] * py - cam.c2[a] + cam.tr[a] - cam.origin[a];
                len2 += d * d;
            }
            const double len = std::sqrt(len2);
            if (!(len >= 0x1p-24) || !(len <= 0x1p24)) return;
        }
    S.sigma = std::ldexp(1.0f, se - 149);
    for (int a = 0; a < 3; ++a) S.step_scaled[a] = (float)std::ldexp((double)st[a], se);
}

// Instances are unrolled into the top-level BVH unless the context (RRT_CFG_NO_UNROLL) or the environment
// (RRT_UNROLL_INSTANCES=0, an experiment knob) says otherwise.
bool unroll_wanted(uint32_t cfg_flags) {
    if (cfg_flags & RRT_CFG_NO_UNROLL) return false;
    const char* e = std::getenv("RRT_UNROLL_INSTANCES");
    return !(e && std::atoi(e) == 0);
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
    // address of THIS device's copy of the static workspace (the device is current); looked up once per device
    if (cfg->device < MAX_DEVICES) {
        if ((e = static_workspace(cfg->device, &c->d_static)) != hipSuccess) {
            rrt_status st = hip_fail(nullptr, e, "rrt_create: hipGetSymbolAddress(scratch)");
            rrt_destroy(c);
            return st;
        }
    }
    c->tune = tuning_from_env();
    *out = c;
    return RRT_OK;
}

rrt_status rrt_debug_get_tuning(const rrt_ctx* c, rrt_tuning* out) {
    if (!c || !out) return set_err(nullptr, RRT_ERR_INVALID, "rrt_debug_get_tuning: NULL argument");
    *out = c->tune;
    return RRT_OK;
}

rrt_status rrt_debug_set_tuning(rrt_ctx* c, const rrt_tuning* tuning) {
    if (!c || !tuning) return set_err(c, RRT_ERR_INVALID, "rrt_debug_set_tuning: NULL argument");
    c->tune = clamp_tuning(*tuning);
    return RRT_OK;
}

void rrt_destroy(rrt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    free_scene(c);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_lists) (void)hipFree(c->d_lists);
    for (void* b : c->d_stage) if (b) (void)hipFree(b);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    delete c;
}

}  // extern "C"

// Scene upload proper.  `prebuilt`: a host scene already built from `scene` without deferred BVHs (a group builds it
// once and uploads it to every member), or NULL to build it here.
static rrt_status scene_upload_impl(rrt_ctx* c, const rrt_scene_desc* scene, HostScene* prebuilt) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_scene_upload: ctx is NULL");
    if (!scene) return set_err(c, RRT_ERR_INVALID, "rrt_scene_upload: scene is NULL");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    // BVHs over at least this many primitives are built on the GPU when the context asks for it
    uint32_t device_min = 0;
    if (c->cfg.flags & RRT_CFG_DEVICE_BUILD) {
        device_min = 4096;
        if (const char* e = std::getenv("RRT_DEVICE_BUILD_MIN")) device_min = (uint32_t)std::max(2, std::atoi(e));
    }
    HostScene local;
    HostScene& hs = prebuilt ? *prebuilt : local;
    std::string err;
    rrt_status st;
    uint64_t total = 0;
    float build_ms = 0.f, upload_ms = 0.f;
    for (;;) {
        const auto t0 = std::chrono::steady_clock::now();
        if (!prebuilt) {
            try {
                st = build_host_scene(scene, hs, err, build_threads(), device_min, unroll_wanted(c->cfg.flags));
            } catch (const std::bad_alloc&) {
                return set_err(c, RRT_ERR_OOM, "rrt_scene_upload: out of host memory while building the BVH");
            } catch (const std::exception& ex) {
                return set_err(c, RRT_ERR_INVALID, std::string("rrt_scene_upload: ") + ex.what());
            }
            if (st != RRT_OK) return set_err(c, st, "rrt_scene_upload: " + err);
        }
        const auto t1 = std::chrono::steady_clock::now();

        RRT_HIP(c, hipDeviceSynchronize());
        free_scene(c);
        total = 0;
        if ((st = upload(c, hs.qnodes, &c->d_nodes, total, hs.tail_nodes)) != RRT_OK || (st = upload(c, hs.tris, &c->d_tris, total, hs.tail_tris)) != RRT_OK ||
            (st = upload(c, hs.spheres, &c->d_spheres, total)) != RRT_OK || (st = upload(c, hs.rects, &c->d_rects, total)) != RRT_OK ||
            (st = upload(c, insts_with_identity(hs.insts), &c->d_insts, total)) != RRT_OK || (st = upload(c, hs.mats, &c->d_mats, total)) != RRT_OK ||
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
            // RRT_LBVH_CLUSTER: primitives per subtree of the GPU's radix tree below which the GPU arranges the primitives
            // (up to 64: surface-area heuristic, one wavefront per subtree; above: the radix tree is kept) and above
            // which the host's SAH builder arranges the subtrees; 0: the plain radix tree
            uint32_t cluster = 64u;
            if (const char* e = std::getenv("RRT_LBVH_CLUSTER")) cluster = (uint32_t)std::max(0, std::atoi(e));
            const auto tb = std::chrono::steady_clock::now();
            const hipError_t e = lbvh_build(db.prims.data(), (uint32_t)db.prims.size(), db.grid, (QNode*)c->d_nodes, db.node_base,
                                            (TriRec*)c->d_tris, db.tri_base, db.tri_count, budget, cluster, build_threads(), &r,
                                            db.raw_runs.data(), db.raw_runs.size(), db.pad);
            if (e != hipSuccess) {
                free_scene(c);
                return hip_fail(c, e, "rrt_scene_upload: GPU BVH build");
            }
            if (std::getenv("RRT_BUILD_TRACE"))
                fprintf(stderr, "[rrt build] GPU BVH: %u primitives, %.1f ms with the upload of the triangles, of it device %.1f ms, upper levels on the host %.1f ms (%u subtrees), depth %u\n",
                        (unsigned)(db.prims.size() + db.tri_count), std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tb).count(),
                        r.build_ms, r.top_ms, r.clusters, r.depth);
            if (r.depth == 0u) { fits = false; break; }
            if (db.mesh < 0) hs.top_depth = r.depth;
            else hs.deepest_mesh = std::max(hs.deepest_mesh, r.depth);
        }
        if (fits && hs.top_depth + ((hs.has_instances || hs.unroll_shift) ? hs.deepest_mesh + 1u : 0u) > MAX_TREE_DEPTH) fits = false;
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
    hs.depth = hs.top_depth + ((hs.has_instances || hs.unroll_shift) ? hs.deepest_mesh : 0u);
    c->info.device_built = (uint32_t)hs.deferred.size();

    SceneDev& S = c->sdev;
    S.nodes = (const QNode*)c->d_nodes; S.tris = (const TriRec*)c->d_tris; S.spheres = (const SphereRec*)c->d_spheres;
    S.rects = (const RectRec*)c->d_rects; S.insts = (const InstRec*)c->d_insts + 1; S.mats = (const MatRec*)c->d_mats;
    S.lights = (const LightRec*)c->d_lights;
    S.light_count = (uint32_t)hs.lights.size();
    S.root = device_ref(hs.root);
    S.has_instances = hs.has_instances ? 1u : 0u;
    S.unroll_shift = hs.unroll_shift;
    S.inst_count = (uint32_t)hs.insts.size();
    S.light_weight = S.light_count ? 1.0f / (float)S.light_count : 0.0f;
    S.light_zone = S.light_count ? (uint32_t)((0x100000000ull) % S.light_count) : 0u;
    S.cam = hs.cam;
    for (int a = 0; a < 3; ++a) S.background[a] = hs.background[a];
    S.t_min = c->cfg.t_min;
    S.max_depth = c->cfg.max_depth;
    S.seed = c->cfg.seed;
    S.grid = hs.grid;
    set_slab_scale(S);

    c->info.prim_count = hs.prim_count;
    c->info.node_count = (uint32_t)hs.qnodes.size() + hs.tail_nodes;
    c->info.tri_count = (uint32_t)hs.tris.size() + hs.tail_tris;
    c->info.bvh_depth = hs.depth;
    c->info.unrolled_instances = hs.unrolled_instances;
    c->info.scaled_slab = (S.sigma != 0.0f && !hs.has_instances) ? 1u : 0u;
    c->info.device_bytes = total;
    c->info.build_ms = build_ms;
    c->info.upload_ms = upload_ms;
    c->info.pixel_list_permille = 0u;
    c->list_share = 1.0f;
    c->waves_ms[0] = c->waves_ms[1] = 0.0f;
    c->waves_launches = 0u;
    c->has_scene = true;
    return RRT_OK;
}

extern "C" {

rrt_status rrt_scene_upload(rrt_ctx* c, const rrt_scene_desc* scene) { return scene_upload_impl(c, scene, nullptr); }

rrt_status rrt_scene_check(const rrt_scene_desc* scene, rrt_scene_info* out) {
    if (!scene || !out) return set_err(nullptr, RRT_ERR_INVALID, "rrt_scene_check: NULL argument");
    HostScene hs;
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    rrt_status st;
    try {
        // RRT_CHECK_DEVICE_MIN: the dry run defers BVHs of at least that many primitives, as an upload with the GPU
        // builder would (times the host side of that path without a GPU)
        uint32_t device_min = 0;
        if (const char* e = std::getenv("RRT_CHECK_DEVICE_MIN")) device_min = (uint32_t)std::max(0, std::atoi(e));
        st = build_host_scene(scene, hs, err, build_threads(), device_min, unroll_wanted(0u));
    } catch (const std::bad_alloc&) {
        return set_err(nullptr, RRT_ERR_OOM, "rrt_scene_check: out of host memory while building the BVH");
    } catch (const std::exception& ex) {
        return set_err(nullptr, RRT_ERR_INVALID, std::string("rrt_scene_check: ") + ex.what());
    }
    if (st != RRT_OK) return set_err(nullptr, st, "rrt_scene_check: " + err);
    std::memset(out, 0, sizeof *out);
    out->prim_count = hs.prim_count;
    out->node_count = (uint32_t)hs.qnodes.size() + hs.tail_nodes;
    out->tri_count = (uint32_t)hs.tris.size() + hs.tail_tris;
    out->bvh_depth = hs.depth;
    out->unrolled_instances = hs.unrolled_instances;
    {
        SceneDev tmp{};
        tmp.cam = hs.cam;
        tmp.grid = hs.grid;
        set_slab_scale(tmp);
        out->scaled_slab = (tmp.sigma != 0.0f && !hs.has_instances) ? 1u : 0u;
    }
    auto sz = [](size_t n, size_t rec) { return (uint64_t)(n ? n : 1) * rec; };
    out->device_bytes = sz(hs.qnodes.size() + hs.tail_nodes, sizeof(QNode)) + sz(hs.tris.size() + hs.tail_tris, sizeof(TriRec)) + sz(hs.spheres.size(), sizeof(SphereRec)) +
                        sz(hs.rects.size(), sizeof(RectRec)) + sz(hs.insts.size() + 1u, sizeof(InstRec)) + sz(hs.mats.size(), sizeof(MatRec)) +
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
    const rrt_tuning& tn = c->tune;
    if (tn.node_share_64ths != RRT_TUNE_AUTO) tuning.node_share_64ths = tn.node_share_64ths;
    if (tn.refill_min != RRT_TUNE_AUTO) tuning.refill_min = tn.refill_min;
    if (tn.blocks_per_cu != RRT_TUNE_AUTO) tuning.blocks_per_cu = tn.blocks_per_cu;
    if (tn.intersect_static != RRT_TUNE_AUTO) tuning.intersect_static = tn.intersect_static;
    if (rrt_status st = order_after_pending(c, s)) return st;
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
    c->pending_chunks = 0;
    c->pending_list_pixels = 0;
    c->pending_waves = -1;
    c->last_counters = c->d_counters;
    return RRT_OK;
}

rrt_status rrt_intersect(rrt_ctx* c, const rrt_ray* rays, size_t n, float t_min, float t_max, rrt_hit* out) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_intersect: ctx is NULL");
    if (!c->has_scene) return set_err(c, RRT_ERR_STATE, "rrt_intersect: no scene uploaded");
    if (n == 0) return RRT_OK;
    if (!rays || !out) return set_err(c, RRT_ERR_INVALID, "rrt_intersect: NULL buffer");
    if (n > 0xF0000000ull) return set_err(c, RRT_ERR_INVALID, "rrt_intersect: batch too large");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    void *d_rays = nullptr, *d_hits = nullptr;
    rrt_status st;
    if ((st = stage(c, 0, n * sizeof(rrt_ray), &d_rays)) != RRT_OK || (st = stage(c, 1, n * sizeof(rrt_hit), &d_hits)) != RRT_OK) return st;
    RRT_HIP(c, hipMemcpy(d_rays, rays, n * sizeof(rrt_ray), hipMemcpyHostToDevice));
    if ((st = rrt_intersect_device(c, (const rrt_ray*)d_rays, n, t_min, t_max, (rrt_hit*)d_hits, nullptr)) != RRT_OK) return st;
    RRT_HIP(c, hipStreamSynchronize(nullptr));
    RRT_HIP(c, hipMemcpy(out, d_hits, n * sizeof(rrt_hit), hipMemcpyDeviceToHost));
    return RRT_OK;
}

}  // extern "C"

// Shared body of rrt_render_device and rrt_render_device_noalloc.  `noalloc`: counters, work-queue heads and partial
// images live in the device's static workspace and nothing is allocated (no device memory, and on the success path no
// host memory either: messages are only built when a call fails); a frame whose partial images would exceed that
// workspace is cut into fewer sample chunks until they fit.
static rrt_status render_fail(rrt_ctx* c, rrt_status st, const char* fn, const char* what) {
    return set_err(c, st, std::string(fn) + ": " + what);
}

static rrt_status render_impl(rrt_ctx* c, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset, float* d_accum,
                              void* stream, bool noalloc, const char* fn) {
    if (!c) return render_fail(nullptr, RRT_ERR_INVALID, fn, "ctx is NULL");
    if (!c->has_scene) return render_fail(c, RRT_ERR_STATE, fn, "no scene uploaded");
    if (!d_accum) return render_fail(c, RRT_ERR_INVALID, fn, "accum is NULL");
    if (width == 0 || height == 0 || spp == 0) return render_fail(c, RRT_ERR_INVALID, fn, "width, height and spp must be > 0");
    // the kernels index the image's floats with 32 bits
    if ((uint64_t)width * height > (0xFFFFFFFFull - 255ull) / 3ull) return render_fail(c, RRT_ERR_INVALID, fn, "image too large");
    if ((uint64_t)spp + spp_offset > 0xFFFFFFFFull) return render_fail(c, RRT_ERR_INVALID, fn, "sample index overflows");
    if (noalloc && !c->d_static) return render_fail(c, RRT_ERR_STATE, fn, "no static workspace on this device");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const rrt_tuning& tn = c->tune;
    // Scenes with a real hierarchy (>= 4096 nodes): paths bounce and walks are long, so items of 8 samples, the
    // straggler rule and lanes that move on to the next item by themselves pay; on tiny scenes each of them costs.
    const bool big_scene = c->info.node_count >= 4096u;
    // Multi-million-node scenes (fetch-bound: more than the caches hold): instanced ones, and short frames, keep items
    // of 64 pixels x one chunk and get larger chunks on large frames (measured on the unrolled 10M-triangle scene at 8K:
    // 256 spp in 8 / 16 / 32 chunks 2350 / 2385 / 2478 ms; items of a few pixels x many chunks: 2376 ms, and 165.7 against
    // 160.0 ms at 16 spp); other large scenes get chunks of 8 samples, which their items hold side by side
    // (launch_render).
    const bool huge_scene = c->info.node_count >= (1u << 21);
    const bool one_tree = c->sdev.unroll_shift != 0u || c->sdev.has_instances == 0u;  // no instances, or all of them unrolled
    const uint64_t pixels = (uint64_t)width * height;
    uint32_t per_item = 8u;
    if (huge_scene && spp >= 64u && pixels >= 6000000ull) per_item = 16u;
    if (huge_scene && spp >= 128u && pixels >= 24000000ull) per_item = 32u;
    // A multi-million-node scene walked as one tree (no instances, or unrolled ones), from 128 samples on: items of 8
    // pixels x 8 long chunks whose primary rays take per-pixel candidate lists (below).  Measured on MI355X, 10M unique triangles at 8K, whole tiles against
    // this form: 64 spp 606.5 / 600.6 ms, 128 spp 1193.0 / 1122.4, 256 spp 2364.1 / 2025.0 (16 / 32 chunks: 2154.6 /
    // 2228.3; 8 chunks without lists: 2322.2), 512 spp 4720.6 / 3940.1 (16 / 32 chunks: 4145.6 / 3940.5), 1024 spp
    // 9368.7 / 7778.6 (16 / 32 chunks: 8108.9 / 8081.2).
    const bool huge_lists = huge_scene && one_tree && tn.pixel_lists != 0u && spp >= 128u;
    // Items of a few pixels x many of their chunks (launch_render) on every other scene without instances that has
    // pixels of unequal cost to even out: suzanne 1080p x 256 spp 11.1 -> 10.6 ms, x 1024 spp 32.7 -> 29.9 ms; not on
    // scenes of a few primitives (12-primitive scene 0.94 -> 1.03 ms with it).
    bool split_items = tn.split_items != RRT_TUNE_AUTO ? tn.split_items != 0u : ((!huge_scene || huge_lists) && c->info.node_count >= 64u);
    // Frames whose primary rays can take per-pixel candidate lists (below; whether a launch collects them also depends on
    // what the last one found, which must not change how a frame is cut): a listed sample costs one trip less, so an
    // item holds 16 samples per lane for its 16 trips (measured on MI355X, 1M-triangle scene at 4K with lists: 128 spp in
    // 8 / 16 / 32 chunks 118.7 / 120.3 / 132.5 ms, 256 spp 223.2 / 220.1 / 231.3 ms, 100 spp 95.6 / 99.3 / 112.7 ms).
    // Scenes with unrolled instances alike (a 100k-triangle mesh x 16 instances at 4K: 128 spp without lists 251.0 ms,
    // with lists in 8 / 16 / 32 chunks 227.9 / 231.8 / 249.6; 1024 spp 1855.5 against 1651.5 / 1590.3 / 1569.4).
    const bool lists_possible = tn.pixel_lists != 0u && spp >= 32u && split_items && (huge_scene ? huge_lists : one_tree);
    if (lists_possible && big_scene) per_item = 16u;
    uint32_t n_chunks = render_chunks(width, height, spp, big_scene ? per_item : 16u);
    if (lists_possible && huge_scene) n_chunks = 8u;
    if (tn.chunks != RRT_TUNE_AUTO) n_chunks = std::min(tn.chunks, spp);
    while (n_chunks > 1u && (uint64_t)((width + 7u) / 8u) * ((height + 7u) / 8u) * n_chunks > 0x7FFFFFFFull) n_chunks /= 2u;
    // Split items hold a power of two of chunks side by side (launch_render): the power of two from 8 to 32 nearest (in
    // ratio, within a factor of two) to the count chosen above; whole tiles where there is none.  Chunk lengths differ
    // by at most one sample (RenderGeom), so every sample count splits.  (A chunk count the caller fixed is taken as it
    // is.)
    // Where the chunks come out unequal and short, the lanes of an item differ by one sample in a few: the fewest
    // chunks in range then (measured on MI355X, 1M-triangle scene at 4K: 100 spp in 8 chunks of 12 or 13 samples
    // 122.4 ms, in 16 chunks of 6 or 7 126.0 ms, in 32 chunks 138.8 ms; whole tiles 123.1 ms).
    if (split_items && n_chunks > 1u && tn.chunks == RRT_TUNE_AUTO) {
        uint32_t best = 0u, smallest = 0u;
        for (uint32_t p = 8u; p <= 32u; p *= 2u) {  // (items of 16 pixels and more are no better than whole tiles)
            if (p > spp || p > 2u * n_chunks || n_chunks > 2u * p) continue;
            if (smallest == 0u) smallest = p;
            if (spp % p != 0u && spp / p < 24u) continue;  // unequal and short
            // nearer to the chosen count in ratio; of two equally near ones the larger
            if (best == 0u || (p >= n_chunks ? (uint64_t)p * best <= (uint64_t)n_chunks * n_chunks : true)) best = p;
        }
        if (best == 0u) best = smallest;
        if (best != 0u) n_chunks = best;
        else split_items = false;
    }
    // The static workspace holds so many partial images and no more: fewer, longer chunks until they fit (the sum of a
    // pixel's samples is then formed in another order than rrt_render_device forms it: same samples, last-bit
    // differences).  One chunk needs no partial image at all.
    if (noalloc && n_chunks > 1u) {
        const uint64_t image_bytes = pixels * 3ull * sizeof(float);
        const uint64_t fit = NOALLOC_PARTIAL_BYTES / image_bytes;  // partial images the workspace holds
        if (fit < 2u)
            return set_err(c, RRT_ERR_INVALID, std::string(fn) + ": a frame of " + std::to_string(pixels) + " pixels is cut into sample chunks, and the static workspace (" +
                                                   std::to_string(NOALLOC_PARTIAL_BYTES) + " bytes) does not hold two partial images of " +
                                                   std::to_string(image_bytes) + " bytes; use rrt_render_device");
        while ((uint64_t)n_chunks > fit) n_chunks /= 2u;  // (stays >= 2, and a power of two where it was one)
    }
    if (n_chunks <= 1u) split_items = false;
    const size_t need = n_chunks > 1u ? (size_t)n_chunks * width * height * 3u * sizeof(float) : 0u;
    float* d_partial = nullptr;
    unsigned long long* d_counters = nullptr;
    if (noalloc) {
        d_counters = reinterpret_cast<unsigned long long*>(c->d_static);
        d_partial = reinterpret_cast<float*>(c->d_static + NOALLOC_COUNTER_BYTES);
    } else {
        if (need > c->partial_bytes) {
            // the old workspace may still be read by a render in flight on another stream
            RRT_HIP(c, hipDeviceSynchronize());
            if (c->d_partial) (void)hipFree(c->d_partial);
            c->d_partial = nullptr;
            c->partial_bytes = 0;
            RRT_HIP(c, hipMalloc((void**)&c->d_partial, need));
            c->partial_bytes = need;
        }
        d_counters = c->d_counters;
        d_partial = c->d_partial;
    }
    RenderTuning tuning;
    if (!big_scene) {
        tuning.leave_walk = 0u;
        tuning.overlap_items = 0u;
        tuning.band_tiles = 0u;
    } else {
        // bands of 64 tiles instead of 16 on frames of 8K width (measured on MI355X: the unrolled 10M-triangle scene in 159.9
        // instead of 162.8 ms, the flat one in 165.0 instead of 169.3; at 4K x 1024 spp 16 / 32 / 64 tiles take 1174 / 1182 /
        // 1195 ms, at 2560x1440 64 cost 1 %)
        // (round 3, after the node step got cheaper: 128 tiles render the flat 10M-triangle scene in 160.4 ms against 162.0)
        if (width >= 7000u) tuning.band_tiles = huge_scene ? 128u : 64u;
    }
    tuning.split_items = split_items ? 1u : 0u;
    if (big_scene && huge_scene) {
        // the walks of multi-million-node scenes have longer tails: more stragglers are left behind per trip
        // (measured on MI355X, ten million triangles: 12 -> 20 is 1-2 % faster, flat and unrolled alike)
        tuning.leave_walk = 20u;
        // ... and a node pass is worth keeping up a little longer (48 / 64 of the walkers descending instead of 40 / 64:
        // 160.4 against 162.0 ms on the flat ten-million-triangle scene, no difference on the unrolled one)
        tuning.node_share_64ths = 48u;
    }
    if (tn.node_share_64ths != RRT_TUNE_AUTO) tuning.node_share_64ths = tn.node_share_64ths;
    if (tn.leave_walk != RRT_TUNE_AUTO) tuning.leave_walk = tn.leave_walk;
    if (tn.leave_ready != RRT_TUNE_AUTO) tuning.leave_ready = tn.leave_ready;
    if (tn.overlap_items != RRT_TUNE_AUTO) tuning.overlap_items = tn.overlap_items;
    if (tn.band_tiles != RRT_TUNE_AUTO) tuning.band_tiles = tn.band_tiles;
    if (tn.scaled_slab != RRT_TUNE_AUTO) tuning.scaled_slab = tn.scaled_slab;
    if (tn.blocks_per_cu != RRT_TUNE_AUTO) tuning.blocks_per_cu = tn.blocks_per_cu;
    // Per-pixel candidate leaf lists for the primary rays (launch_render; split items in the pipelined form): collecting
    // them costs about what one sample per pixel costs, and a listed pixel's samples then cost about a fifth less, so
    // they pay from a few dozen samples of listed pixels on.  The share of pixels that get a list (fine geometry: few) is
    // taken from the scene's last render whose statistics were read; until one was, every pixel is assumed to.
    tuning.pixel_lists = tn.pixel_lists != RRT_TUNE_AUTO ? tn.pixel_lists : (lists_possible && (float)spp * c->list_share >= 32.0f ? 1u : 0u);
    if (tn.pixel_list_len != RRT_TUNE_AUTO) tuning.pixel_list_len = tn.pixel_list_len;
    if (big_scene && huge_scene && !noalloc && render_takes_pixel_lists(c->sdev, width, height, n_chunks, tuning)) {
        // a multi-million-node scene whose primary rays take lists: the walks of a trip are then all bounce rays, and the
        // straggler threshold of the other large scenes is the better one again, as are narrower bands (measured on
        // MI355X, 256 spp at 8K: leave_walk 8 / 12 / 20 / 28 -> flat 2050 / 2019 / 2027 / 2091 ms, unrolled 2189 / 2166 /
        // 2195 / 2288; bands of 32 / 64 / 128 tiles -> flat 2034 / 2035 / 2027, unrolled 2162 / 2169 / 2195)
        if (tn.leave_walk == RRT_TUNE_AUTO) tuning.leave_walk = RenderTuning().leave_walk;
        if (tn.band_tiles == RRT_TUNE_AUTO && width >= 7000u) tuning.band_tiles = 64u;
    }
    int waves_slot = -1;
    if (tn.list_waves != RRT_TUNE_AUTO) {
        tuning.list_waves = tn.list_waves;
    } else if (tuning.pixel_lists != 0u && tuning.overlap_items != 0u) {
        if (c->waves_key[0] != width || c->waves_key[1] != height || c->waves_key[2] != spp) {
            c->waves_key[0] = width; c->waves_key[1] = height; c->waves_key[2] = spp;
            c->waves_ms[0] = c->waves_ms[1] = 0.0f;
            c->waves_launches = 0u;
        }
        waves_slot = c->waves_launches == 0u ? 0 : (c->waves_ms[1] == 0.0f ? 1 : (c->waves_ms[0] == 0.0f ? 0 : (c->waves_ms[1] < c->waves_ms[0] ? 1 : 0)));
        tuning.list_waves = waves_slot == 1 ? 6u : 7u;
    }
    // per-pixel candidate leaf lists for the primary rays (launch_render): a frame-sized buffer of the context, grown on
    // demand; the entry point that may not allocate goes without them
    uint32_t* d_lists = nullptr;
    if (!noalloc && render_takes_pixel_lists(c->sdev, width, height, n_chunks, tuning)) {
        const size_t lists_need = (size_t)pixels * PIXEL_LIST_DWORDS * sizeof(uint32_t);
        if (lists_need > c->lists_bytes) {
            RRT_HIP(c, hipDeviceSynchronize());
            if (c->d_lists) (void)hipFree(c->d_lists);
            c->d_lists = nullptr;
            c->lists_bytes = 0;
            RRT_HIP(c, hipMalloc((void**)&c->d_lists, lists_need));
            c->lists_bytes = lists_need;
        }
        d_lists = c->d_lists;
    }
    if (rrt_status st = order_after_pending(c, s)) return st;
    RRT_HIP(c, hipMemsetAsync(d_counters, 0, 256, s));
    RRT_HIP(c, hipEventRecord(c->ev_start, s));
    RRT_HIP(c, launch_render(c->sdev, width, height, spp, spp_offset, d_accum, d_partial, n_chunks, d_counters,
                             reinterpret_cast<unsigned int*>(d_counters + 8), c->cu_count,
                             (c->cfg.flags & RRT_CFG_COUNT_VISITS) != 0, tuning, s, d_lists));
    RRT_HIP(c, hipEventRecord(c->ev_stop, s));
    c->last_stream = s;
    c->render_pending = true;
    c->pending_samples = (uint64_t)width * height * spp;
    c->pending_rays = 0;
    c->pending_chunks = n_chunks;
    c->pending_list_pixels = d_lists ? pixels : 0u;
    c->pending_waves = d_lists ? waves_slot : -1;
    c->last_counters = d_counters;
    return RRT_OK;
}

extern "C" {

rrt_status rrt_render_device(rrt_ctx* c, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                             float* d_accum, void* stream) {
    return render_impl(c, width, height, spp, spp_offset, d_accum, stream, false, "rrt_render_device");
}

rrt_status rrt_render_device_noalloc(rrt_ctx* c, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                                     float* d_accum, void* stream) {
    return render_impl(c, width, height, spp, spp_offset, d_accum, stream, true, "rrt_render_device_noalloc");
}

rrt_status rrt_render_stats(rrt_ctx* c, rrt_stats* stats) {
    if (!c || !stats) return set_err(c, RRT_ERR_INVALID, "rrt_render_stats: NULL argument");
    if (!c->render_pending) return set_err(c, RRT_ERR_STATE, "rrt_render_stats: no render was issued");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    RRT_HIP(c, hipEventSynchronize(c->ev_stop));
    unsigned long long h[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [8..11] are the work-queue heads
    RRT_HIP(c, hipMemcpy(h, c->last_counters ? c->last_counters : c->d_counters, sizeof h, hipMemcpyDeviceToHost));
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
    stats->chunks = c->pending_chunks;
    if (c->pending_waves >= 0) {
        if (c->waves_launches != 0u && c->waves_ms[c->pending_waves] == 0.0f && ms > 0.0f) c->waves_ms[c->pending_waves] = ms;
        c->waves_launches++;
    }
    c->pending_waves = -1;
    if (c->pending_list_pixels != 0u) {
        c->list_share = (float)((double)h[14] / (double)c->pending_list_pixels);
        c->info.pixel_list_permille = (uint32_t)(c->list_share * 1000.0f + 0.5f);
    }
    return RRT_OK;
}

rrt_status rrt_render(rrt_ctx* c, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset, float* accum,
                      rrt_stats* stats) {
    if (!c) return set_err(nullptr, RRT_ERR_INVALID, "rrt_render: ctx is NULL");
    if (!accum) return set_err(c, RRT_ERR_INVALID, "rrt_render: accum is NULL");
    if (width == 0 || height == 0) return set_err(c, RRT_ERR_INVALID, "rrt_render: width and height must be > 0");
    RRT_HIP(c, hipSetDevice(c->cfg.device));
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    void* d_accum = nullptr;
    rrt_status st = stage(c, 0, bytes, &d_accum);
    if (st != RRT_OK) return st;
    if ((st = rrt_render_device(c, width, height, spp, spp_offset, (float*)d_accum, nullptr)) != RRT_OK) return st;
    rrt_stats local;
    if ((st = rrt_render_stats(c, &local)) != RRT_OK) return st;
    RRT_HIP(c, hipMemcpy(accum, d_accum, bytes, hipMemcpyDeviceToHost));
    if (stats) *stats = local;
    return RRT_OK;
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
    void *d_accum = nullptr, *d_rgb = nullptr;
    rrt_status st;
    if ((st = stage(c, 0, n * sizeof(float), &d_accum)) != RRT_OK || (st = stage(c, 1, n, &d_rgb)) != RRT_OK) return st;
    RRT_HIP(c, hipMemcpy(d_accum, accum, n * sizeof(float), hipMemcpyHostToDevice));
    if ((st = rrt_tonemap_device(c, (const float*)d_accum, width, height, spp_total, (uint8_t*)d_rgb, nullptr)) != RRT_OK) return st;
    RRT_HIP(c, hipStreamSynchronize(nullptr));
    RRT_HIP(c, hipMemcpy(rgb8, d_rgb, n, hipMemcpyDeviceToHost));
    return RRT_OK;
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

// ---------------------------------------------------------------------------------------------------------------
// Multi-GPU group: one host process, one context and one stream per GPU.  Replaces the reference's only form of
// parallelism, the rayon row pool (src/main.rs:42-49,86), across GPUs: every sample is independent (src/main.rs:91),
// so member g renders a contiguous slice of the sample indices of every pixel and the f32 accumulation buffers are
// summed on member 0 -- one RCCL ncclReduce over xGMI when the members sit on distinct GPUs (librccl is opened at run
// time, only then), or a copy to member 0's GPU plus an add kernel otherwise (members sharing a GPU, RCCL absent,
// RRT_GROUP_NO_RCCL set).
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

// librccl, opened the first time a group wants it (once per process, whichever thread comes first); nullptr when the
// library or one of its entry points is missing.
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(api.handle, "ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.Reduce = reinterpret_cast<decltype(api.Reduce)>(dlsym(api.handle, "ncclReduce"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.handle, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.handle, "ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        api.ok = api.CommInitAll && api.CommDestroy && api.Reduce && api.GroupStart && api.GroupEnd && api.GetErrorString;
    });
    return api.ok ? &api : nullptr;
}

}  // namespace

struct rrt_group {
    std::vector<rrt_ctx*> members;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;       // per member: its render (and, for RCCL, its part of the reduce) is complete
    std::vector<float*> d_accum;        // per member: accumulation buffer on its GPU (grown on demand)
    float* d_recv = nullptr;            // on member 0's GPU: landing buffer of the copy-and-add reduce
    size_t accum_bytes = 0;
    std::vector<ncclComm_t> comms;      // one per member when the reduce goes through RCCL, else empty
    std::string err;
    std::string reduce_path;            // "none" (one member), "rccl" or "copy+add"
};

namespace {

rrt_status group_err(rrt_group* g, rrt_status st, const std::string& msg) {
    if (g) g->err = msg;
    else g_thread_error = msg;
    return st;
}
rrt_status group_member_err(rrt_group* g, size_t i, rrt_status st) {
    return group_err(g, st, "member " + std::to_string(i) + " (device " + std::to_string(g->members[i]->cfg.device) + "): " + g->members[i]->err);
}
#define RRT_GHIP(g, call)                                                                                        \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return group_err(g, e_ == hipErrorOutOfMemory ? RRT_ERR_OOM : RRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

extern "C" {

rrt_status rrt_group_create(const rrt_config* cfg, const int32_t* devices, uint32_t n_devices, rrt_group** out) {
    if (!cfg || !devices || !out) return group_err(nullptr, RRT_ERR_INVALID, "rrt_group_create: NULL argument");
    *out = nullptr;
    if (n_devices == 0 || n_devices > 64u) return group_err(nullptr, RRT_ERR_INVALID, "rrt_group_create: 1..64 devices");
    std::unique_ptr<rrt_group> g(new (std::nothrow) rrt_group());
    if (!g) return group_err(nullptr, RRT_ERR_OOM, "rrt_group_create: out of host memory");
    auto fail = [&](rrt_status st, const std::string& msg) {
        rrt_group_destroy(g.release());
        return group_err(nullptr, st, msg);
    };
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; ++i) {
        for (uint32_t j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
        rrt_config member_cfg = *cfg;
        member_cfg.device = devices[i];
        rrt_ctx* c = nullptr;
        const rrt_status st = rrt_create(&member_cfg, &c);
        if (st != RRT_OK) return fail(st, "rrt_group_create: device " + std::to_string(devices[i]) + ": " + g_thread_error);
        g->members.push_back(c);
        hipStream_t s = nullptr;
        hipEvent_t ev = nullptr;
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        g->streams.push_back(s);
        g->done.push_back(ev);
        g->d_accum.push_back(nullptr);
        if (e != hipSuccess) return fail(RRT_ERR_HIP, std::string("rrt_group_create: ") + hipGetErrorString(e));
    }
    // The sum goes through RCCL when the listed devices are distinct and there is somebody to sum with -- or nobody,
    // but the caller asked for the RCCL calls all the same (RRT_CFG_GROUP_FORCE_RCCL: a one-GPU machine then runs
    // communicator set-up, the grouped reduce and its stream ordering).  Without librccl, or when the communicators
    // cannot be set up, the members' buffers are copied to member 0 and added there.
    g->reduce_path = n_devices == 1u ? "none" : "copy+add";
    const bool force = (cfg->flags & RRT_CFG_GROUP_FORCE_RCCL) != 0u;
    if ((n_devices > 1u || force) && distinct && !std::getenv("RRT_GROUP_NO_RCCL")) {
        if (RcclApi* api = rccl_api()) {
            g->comms.assign(n_devices, nullptr);
            std::vector<int> devs(devices, devices + n_devices);
            const ncclResult_t r = api->CommInitAll(g->comms.data(), (int)n_devices, devs.data());
            if (r == ncclSuccess) {
                g->reduce_path = "rccl";
            } else {
                g->comms.clear();
                g->err = std::string("ncclCommInitAll: ") + api->GetErrorString(r) + " (summing by copy and add instead)";
                if (force) return fail(RRT_ERR_HIP, "rrt_group_create: " + g->err);
            }
        } else if (force) {
            return fail(RRT_ERR_HIP, "rrt_group_create: RRT_CFG_GROUP_FORCE_RCCL, but librccl could not be opened");
        }
    } else if (force) {
        return fail(RRT_ERR_INVALID, "rrt_group_create: RRT_CFG_GROUP_FORCE_RCCL needs distinct devices (and RRT_GROUP_NO_RCCL unset)");
    }
    *out = g.release();
    return RRT_OK;
}

void rrt_group_destroy(rrt_group* g) {
    if (!g) return;
    if (!g->comms.empty())
        if (RcclApi* api = rccl_api())
            for (ncclComm_t cm : g->comms) if (cm) (void)api->CommDestroy(cm);
    for (size_t i = 0; i < g->members.size(); ++i) {
        (void)hipSetDevice(g->members[i]->cfg.device);
        if (i < g->streams.size() && g->streams[i]) { (void)hipStreamSynchronize(g->streams[i]); (void)hipStreamDestroy(g->streams[i]); }
        if (i < g->done.size() && g->done[i]) (void)hipEventDestroy(g->done[i]);
        if (i < g->d_accum.size() && g->d_accum[i]) (void)hipFree(g->d_accum[i]);
        if (i == 0 && g->d_recv) (void)hipFree(g->d_recv);
        rrt_destroy(g->members[i]);
    }
    delete g;
}

const char* rrt_group_last_error(const rrt_group* g) { return g ? g->err.c_str() : g_thread_error.c_str(); }

uint32_t rrt_group_size(const rrt_group* g) { return g ? (uint32_t)g->members.size() : 0u; }

const char* rrt_group_reduce_path(const rrt_group* g) { return g ? g->reduce_path.c_str() : ""; }

rrt_status rrt_group_scene_upload(rrt_group* g, const rrt_scene_desc* scene) {
    if (!g) return group_err(nullptr, RRT_ERR_INVALID, "rrt_group_scene_upload: group is NULL");
    if (!scene) return group_err(g, RRT_ERR_INVALID, "rrt_group_scene_upload: scene is NULL");
    // the host BVH is built once and uploaded to every member (with the GPU builder every member builds its own)
    HostScene hs;
    const bool share = (g->members[0]->cfg.flags & RRT_CFG_DEVICE_BUILD) == 0u;
    float build_ms = 0.f;
    if (share) {
        std::string err;
        rrt_status st;
        const auto t0 = std::chrono::steady_clock::now();
        try {
            st = build_host_scene(scene, hs, err, build_threads(), 0u, unroll_wanted(g->members[0]->cfg.flags));
        } catch (const std::bad_alloc&) {
            return group_err(g, RRT_ERR_OOM, "rrt_group_scene_upload: out of host memory while building the BVH");
        } catch (const std::exception& ex) {
            return group_err(g, RRT_ERR_INVALID, std::string("rrt_group_scene_upload: ") + ex.what());
        }
        if (st != RRT_OK) return group_err(g, st, "rrt_group_scene_upload: " + err);
        build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    for (size_t i = 0; i < g->members.size(); ++i) {
        const rrt_status st = scene_upload_impl(g->members[i], scene, share ? &hs : nullptr);
        if (st != RRT_OK) return group_member_err(g, i, st);
        if (share) g->members[i]->info.build_ms = build_ms;
    }
    return RRT_OK;
}

rrt_status rrt_group_scene_get_info(const rrt_group* g, rrt_scene_info* out) {
    if (!g || !out) return group_err(nullptr, RRT_ERR_INVALID, "rrt_group_scene_get_info: NULL argument");
    const rrt_status st = rrt_scene_get_info(g->members[0], out);
    if (st != RRT_OK) return group_member_err(const_cast<rrt_group*>(g), 0, st);
    return RRT_OK;
}

rrt_status rrt_group_render(rrt_group* g, uint32_t width, uint32_t height, uint32_t spp_total, uint32_t spp_offset,
                            float* accum, rrt_stats* stats) {
    if (!g) return group_err(nullptr, RRT_ERR_INVALID, "rrt_group_render: group is NULL");
    if (!accum) return group_err(g, RRT_ERR_INVALID, "rrt_group_render: accum is NULL");
    if (width == 0 || height == 0 || spp_total == 0) return group_err(g, RRT_ERR_INVALID, "rrt_group_render: width, height and spp must be > 0");
    if ((uint64_t)width * height > (0xFFFFFFFFull - 255ull) / 3ull) return group_err(g, RRT_ERR_INVALID, "rrt_group_render: image too large");
    // (checked here for all members, so that no member is launched when another one's slice would be refused)
    if ((uint64_t)spp_total + spp_offset > 0xFFFFFFFFull) return group_err(g, RRT_ERR_INVALID, "rrt_group_render: sample index overflows");
    const size_t n = g->members.size();
    for (size_t i = 0; i < n; ++i)
        if (!g->members[i]->has_scene) return group_member_err(g, i, set_err(g->members[i], RRT_ERR_STATE, "rrt_group_render: no scene uploaded"));
    const size_t n_floats = (size_t)width * height * 3u, bytes = n_floats * sizeof(float);
    if (bytes > g->accum_bytes) {
        for (size_t i = 0; i < n; ++i) {
            RRT_GHIP(g, hipSetDevice(g->members[i]->cfg.device));
            RRT_GHIP(g, hipDeviceSynchronize());
            if (g->d_accum[i]) (void)hipFree(g->d_accum[i]);
            g->d_accum[i] = nullptr;
            RRT_GHIP(g, hipMalloc((void**)&g->d_accum[i], bytes));
            if (i == 0 && n > 1 && g->comms.empty()) {
                if (g->d_recv) (void)hipFree(g->d_recv);
                g->d_recv = nullptr;
                RRT_GHIP(g, hipMalloc((void**)&g->d_recv, bytes));
            }
        }
        g->accum_bytes = bytes;
    }
    // sample slices: spp_total split evenly, the remainder to the low members; members left without samples sit out
    std::vector<uint32_t> first(n), count(n);
    {
        const uint32_t base = spp_total / (uint32_t)n, rem = spp_total % (uint32_t)n;
        uint32_t off = spp_offset;
        for (size_t i = 0; i < n; ++i) {
            count[i] = base + (i < rem ? 1u : 0u);
            first[i] = off;
            off += count[i];
        }
    }
    for (size_t i = 0; i < n; ++i) {
        rrt_ctx* c = g->members[i];
        RRT_GHIP(g, hipSetDevice(c->cfg.device));
        if (count[i] == 0u) {
            RRT_GHIP(g, hipMemsetAsync(g->d_accum[i], 0, bytes, g->streams[i]));
        } else {
            const rrt_status st = rrt_render_device(c, width, height, count[i], first[i], g->d_accum[i], g->streams[i]);
            if (st != RRT_OK) {
                // the members launched so far still write their buffers: drained before the caller gets the error
                for (size_t k = 0; k < i; ++k) {
                    (void)hipSetDevice(g->members[k]->cfg.device);
                    (void)hipStreamSynchronize(g->streams[k]);
                }
                return group_member_err(g, i, st);
            }
        }
    }
    // the exchange step: sum of the members' buffers on member 0
    if (!g->comms.empty()) {
        // One in-place ncclReduce per member, root = member 0, issued as one group on the members' own streams: each
        // stream runs its member's render first, then its part of the reduce, so nothing else has to order them.
        RcclApi* api = rccl_api();
        ncclResult_t r = api->GroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; ++i)
            r = api->Reduce(g->d_accum[i], g->d_accum[i], n_floats, ncclFloat, ncclSum, 0, g->comms[i], g->streams[i]);
        const ncclResult_t r_end = api->GroupEnd();
        if (r == ncclSuccess) r = r_end;
        if (r != ncclSuccess) return group_err(g, RRT_ERR_HIP, std::string("rrt_group_render: RCCL reduce: ") + api->GetErrorString(r));
        for (size_t i = 1; i < n; ++i) {
            RRT_GHIP(g, hipSetDevice(g->members[i]->cfg.device));
            RRT_GHIP(g, hipStreamSynchronize(g->streams[i]));
        }
    } else if (n > 1) {
        for (size_t i = 1; i < n; ++i) {
            RRT_GHIP(g, hipSetDevice(g->members[i]->cfg.device));
            RRT_GHIP(g, hipEventRecord(g->done[i], g->streams[i]));
        }
        RRT_GHIP(g, hipSetDevice(g->members[0]->cfg.device));
        for (size_t i = 1; i < n; ++i) {  // in member order, so that the sum is reproducible
            RRT_GHIP(g, hipStreamWaitEvent(g->streams[0], g->done[i], 0));
            const float* src = g->d_accum[i];
            if (g->members[i]->cfg.device != g->members[0]->cfg.device) {
                RRT_GHIP(g, hipMemcpyPeerAsync(g->d_recv, g->members[0]->cfg.device, g->d_accum[i], g->members[i]->cfg.device, bytes, g->streams[0]));
                src = g->d_recv;
            }
            RRT_GHIP(g, launch_add(g->d_accum[0], src, n_floats, g->streams[0]));
        }
    }
    RRT_GHIP(g, hipSetDevice(g->members[0]->cfg.device));
    RRT_GHIP(g, hipMemcpyAsync(accum, g->d_accum[0], bytes, hipMemcpyDeviceToHost, g->streams[0]));
    RRT_GHIP(g, hipStreamSynchronize(g->streams[0]));
    rrt_stats total;
    std::memset(&total, 0, sizeof total);
    for (size_t i = 0; i < n; ++i) {
        if (count[i] == 0u) continue;
        rrt_stats st;
        const rrt_status rs = rrt_render_stats(g->members[i], &st);
        if (rs != RRT_OK) return group_member_err(g, i, rs);
        total.rays += st.rays; total.samples += st.samples; total.node_visits += st.node_visits; total.prim_tests += st.prim_tests;
        total.node_wave_steps += st.node_wave_steps; total.prim_wave_steps += st.prim_wave_steps;
        total.shade_wave_passes += st.shade_wave_passes; total.refill_wave_passes += st.refill_wave_passes;
        total.chunks = std::max(total.chunks, st.chunks);
        total.node_slots_lane_done += st.node_slots_lane_done; total.node_slots_lane_at_leaf += st.node_slots_lane_at_leaf;
        total.kernel_ms = std::max(total.kernel_ms, st.kernel_ms);  // the members render side by side
    }
    if (stats) *stats = total;
    return RRT_OK;
}

rrt_status rrt_group_tonemap(rrt_group* g, const float* accum, uint32_t width, uint32_t height, uint32_t spp_total, uint8_t* rgb8) {
    if (!g) return group_err(nullptr, RRT_ERR_INVALID, "rrt_group_tonemap: group is NULL");
    const rrt_status st = rrt_tonemap(g->members[0], accum, width, height, spp_total, rgb8);
    if (st != RRT_OK) return group_member_err(g, 0, st);
    return RRT_OK;
}

}  // extern "C"
