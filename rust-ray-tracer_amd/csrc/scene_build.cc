// scene_build.cc -- scene description -> primitive records with reference bounding boxes -> flat BVH.
#include "scene_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "bvh_builder.h"

namespace rrt {
namespace {

// Runs body(lo, hi) over [0, n) in contiguous pieces, on up to `threads` host threads when n is large.
template <typename F>
static void parallel_for(uint32_t n, unsigned threads, F body) {
    const unsigned pieces = n >= (1u << 13) ? std::min<unsigned>(std::max(threads, 1u), 64u) : 1u;
    if (pieces <= 1u) {
        body(0u, n);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned k = 0; k < pieces; ++k) {
        const uint32_t lo = (uint32_t)((uint64_t)n * k / pieces), hi = (uint32_t)((uint64_t)n * (k + 1) / pieces);
        pool.emplace_back([=, &body]() { body(lo, hi); });
    }
    for (std::thread& t : pool) t.join();
}

struct Expander {
    const rrt_scene_desc* d;
    unsigned threads = 1;
    HostScene* hs;
    std::vector<TriRec> tri_src;  // triangle records in submission order; reordered into leaf order at the end
    float pad = 0.f;              // conservative extra padding of every box
    std::string* err;

    bool fail(const std::string& m) {
        *err = m;
        return false;
    }

    // reference src/geometry/triangle.rs:82-94
    static void tri_box(const rrt_triangle& t, float* mn, float* mx) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(t.v0[a], std::min(t.v1[a], t.v2[a])) - 0.0001f;
            mx[a] = std::max(t.v0[a], std::max(t.v1[a], t.v2[a])) + 0.0001f;
        }
    }
    void widen(BuildPrim& p) const {
        for (int a = 0; a < 3; ++a) { p.bmin[a] -= pad; p.bmax[a] += pad; }
    }

    bool add_rect(const rrt_rect& r, uint32_t id, std::vector<BuildPrim>& prims) {
        if (r.axis > 2u) return fail("rect axis must be 0, 1 or 2");
        if (r.material >= d->material_count) return fail("rect material index out of range");
        RectRec rec;
        rec.a0 = r.a0; rec.a1 = r.a1; rec.b0 = r.b0; rec.b1 = r.b1; rec.k = r.k;
        rec.axis_flags = r.axis | ((r.flags & RRT_RECT_FLIP_FACE) ? 0x100u : 0u);
        rec.id = id;
        rec.material = r.material;
        BuildPrim p;
        // reference src/geometry/aarect.rs:60-66,115-121,191-197
        const uint32_t ia = r.axis == 0 ? 1u : 0u, ib = r.axis == 2 ? 1u : 2u;
        p.bmin[r.axis] = r.k - 0.0001f; p.bmax[r.axis] = r.k + 0.0001f;
        p.bmin[ia] = std::min(r.a0, r.a1); p.bmax[ia] = std::max(r.a0, r.a1);
        p.bmin[ib] = std::min(r.b0, r.b1); p.bmax[ib] = std::max(r.b0, r.b1);
        p.kind = KIND_RECT;
        p.index = (uint32_t)hs->rects.size();
        widen(p);
        hs->rects.push_back(rec);
        prims.push_back(p);
        return true;
    }

    // Appends the primitives of `objs` to `prims`; ids start at `id` and advance.  mesh_info: per mesh
    // {root node, primitive count, root box} for instances (top level only).
    struct MeshInfo {
        int32_t root;
        uint32_t prim_count;
        uint32_t depth;
        float mn[3], mx[3];
        Grid grid;
    };
    bool expand(const rrt_object* objs, uint32_t n_objs, uint32_t& id, std::vector<BuildPrim>& prims,
                const std::vector<MeshInfo>* meshes) {
        {   // one growth of the big arrays instead of one per run (a scene may list its triangles in many runs)
            size_t n_tris = 0, n_other = 0;
            for (uint32_t oi = 0; oi < n_objs; ++oi) {
                if (objs[oi].type == RRT_OBJ_TRIANGLES) n_tris += objs[oi].count;
                else n_other += objs[oi].type == RRT_OBJ_CUBOIDS ? 6u * (size_t)objs[oi].count : objs[oi].count;
            }
            tri_src.reserve(tri_src.size() + n_tris);
            prims.reserve(prims.size() + n_tris + n_other);
        }
        for (uint32_t oi = 0; oi < n_objs; ++oi) {
            const rrt_object& ob = objs[oi];
            switch (ob.type) {
                case RRT_OBJ_TRIANGLES: {
                    if ((uint64_t)ob.first + ob.count > d->triangle_count) return fail("triangle run out of range");
                    // large runs are filled by several threads: every triangle has its slot and its id up front
                    const size_t src0 = tri_src.size(), prim0 = prims.size();
                    const uint32_t id0 = id;
                    tri_src.resize(src0 + ob.count);
                    prims.resize(prim0 + ob.count);
                    std::atomic<bool> bad_material{false};
                    parallel_for(ob.count, threads, [&](uint32_t lo, uint32_t hi) {
                        for (uint32_t i = lo; i < hi; ++i) {
                            const rrt_triangle& t = d->triangles[ob.first + i];
                            if (t.material >= d->material_count) bad_material.store(true, std::memory_order_relaxed);
                            TriRec rec;
                            for (int a = 0; a < 3; ++a) {
                                rec.v0[a] = t.v0[a];
                                rec.e1[a] = t.v1[a] - t.v0[a];  // triangle.rs:28-29
                                rec.e2[a] = t.v2[a] - t.v0[a];
                            }
                            rec.id = id0 + i;
                            rec.material = t.material;
                            rec.pad = 0;
                            BuildPrim p;
                            tri_box(t, p.bmin, p.bmax);
                            p.kind = KIND_TRI;
                            p.index = (uint32_t)(src0 + i);
                            widen(p);
                            tri_src[src0 + i] = rec;
                            prims[prim0 + i] = p;
                        }
                    });
                    if (bad_material.load()) return fail("triangle material index out of range");
                    id += ob.count;
                    break;
                }
                case RRT_OBJ_SPHERES: {
                    if ((uint64_t)ob.first + ob.count > d->sphere_count) return fail("sphere run out of range");
                    for (uint32_t i = 0; i < ob.count; ++i) {
                        const rrt_sphere& s = d->spheres[ob.first + i];
                        if (s.material >= d->material_count) return fail("sphere material index out of range");
                        SphereRec rec;
                        std::memset(&rec, 0, sizeof rec);
                        for (int a = 0; a < 3; ++a) rec.c[a] = s.center[a];
                        rec.r = s.radius;
                        rec.id = id++;
                        rec.material = s.material;
                        BuildPrim p;
                        const float ar = std::fabs(s.radius);
                        for (int a = 0; a < 3; ++a) {  // sphere.rs:128-133
                            p.bmin[a] = s.center[a] - ar;
                            p.bmax[a] = s.center[a] + ar;
                        }
                        p.kind = KIND_SPHERE;
                        p.index = (uint32_t)hs->spheres.size();
                        widen(p);
                        hs->spheres.push_back(rec);
                        prims.push_back(p);
                    }
                    break;
                }
                case RRT_OBJ_RECTS: {
                    if ((uint64_t)ob.first + ob.count > d->rect_count) return fail("rect run out of range");
                    for (uint32_t i = 0; i < ob.count; ++i)
                        if (!add_rect(d->rects[ob.first + i], id++, prims)) return false;
                    break;
                }
                case RRT_OBJ_CUBOIDS: {
                    if ((uint64_t)ob.first + ob.count > d->cuboid_count) return fail("cuboid run out of range");
                    for (uint32_t i = 0; i < ob.count; ++i) {
                        const rrt_cuboid& c = d->cuboids[ob.first + i];
                        // the six sides in the order of reference src/geometry/cuboid.rs:24-73
                        const rrt_rect sides[6] = {
                            {2u, c.min[0], c.max[0], c.min[1], c.max[1], c.max[2], c.material, 0u},
                            {2u, c.min[0], c.max[0], c.min[1], c.max[1], c.min[2], c.material, 0u},
                            {1u, c.min[0], c.max[0], c.min[2], c.max[2], c.max[1], c.material, 0u},
                            {1u, c.min[0], c.max[0], c.min[2], c.max[2], c.min[1], c.material, 0u},
                            {0u, c.min[1], c.max[1], c.min[2], c.max[2], c.max[0], c.material, 0u},
                            {0u, c.min[1], c.max[1], c.min[2], c.max[2], c.min[0], c.material, 0u},
                        };
                        for (int k = 0; k < 6; ++k)
                            if (!add_rect(sides[k], id++, prims)) return false;
                    }
                    break;
                }
                case RRT_OBJ_INSTANCE: {
                    if (!meshes) return fail("a mesh may not contain instances");
                    if (ob.count != 1u) return fail("instance objects must have count 1");
                    if (ob.first >= d->instance_count) return fail("instance index out of range");
                    const rrt_instance& in = d->instances[ob.first];
                    if (in.mesh >= d->mesh_count) return fail("instance mesh index out of range");
                    const MeshInfo& mi = (*meshes)[in.mesh];
                    InstRec rec;
                    std::memset(&rec, 0, sizeof rec);
                    for (int a = 0; a < 3; ++a) rec.offset[a] = in.offset[a];
                    rec.sin_t = in.sin_theta;
                    rec.cos_t = in.cos_theta;
                    rec.root = device_ref(mi.root);
                    for (int a = 0; a < 3; ++a) { rec.grid_origin[a] = mi.grid.origin[a]; rec.grid_step[a] = mi.grid.step[a]; }
                    rec.id_base = id;
                    id += mi.prim_count;
                    BuildPrim p;
                    for (int a = 0; a < 3; ++a) { p.bmin[a] = FLT_MAX; p.bmax[a] = -FLT_MAX; }
                    for (int c = 0; c < 8; ++c) {  // world box of the rotated + translated mesh box
                        const float x = (c & 1) ? mi.mx[0] : mi.mn[0];
                        const float y = (c & 2) ? mi.mx[1] : mi.mn[1];
                        const float z = (c & 4) ? mi.mx[2] : mi.mn[2];
                        const float q[3] = {in.cos_theta * x + in.sin_theta * z + in.offset[0], y + in.offset[1],
                                            -in.sin_theta * x + in.cos_theta * z + in.offset[2]};
                        for (int a = 0; a < 3; ++a) { p.bmin[a] = std::min(p.bmin[a], q[a]); p.bmax[a] = std::max(p.bmax[a], q[a]); }
                    }
                    p.kind = KIND_INST;
                    p.index = (uint32_t)hs->insts.size();
                    widen(p);
                    widen(p);
                    hs->insts.push_back(rec);
                    prims.push_back(p);
                    break;
                }
                default:
                    return fail("unknown object type");
            }
        }
        return true;
    }
};

// max |v[i]|, or +inf as soon as a component is not finite (std::max alone would skip a NaN)
static float max_abs3(const float* v) {
    float m = 0.f;
    for (int i = 0; i < 3; ++i) {
        if (!std::isfinite(v[i])) return INFINITY;
        m = std::max(m, std::fabs(v[i]));
    }
    return m;
}

// Grid of 65536 planes per axis over a BVH's root box [mn, mx], with two steps of slack at the low end.  The step
// is at least 2^-20 of the largest coordinate, so that rounding the origin to a float moves it by a small
// fraction of a step.
}  // namespace

Grid make_grid(const float* mn, const float* mx) {
    Grid g;
    for (int a = 0; a < 3; ++a) {
        double lo = mn[a], hi = mx[a];
        if (!(hi >= lo)) { lo = 0.0; hi = 0.0; }  // empty BVH
        const double mag = std::max(std::fabs(lo), std::fabs(hi));
        double step = std::max((hi - lo) / 65528.0, mag * (1.0 / 1048576.0));
        step = std::max(step, 1e-30);
        float stepf = (float)step;
        float originf = (float)(lo - 2.0 * (double)stepf);
        while ((double)originf > lo) originf = std::nextafterf(originf, -INFINITY);
        while ((double)originf + 65535.0 * (double)stepf < hi) stepf = std::nextafterf(stepf, INFINITY);
        g.origin[a] = originf;
        g.step[a] = stepf;
    }
    return g;
}

namespace {

// Appends the quantized form of one BVH's nodes (builder output `built`, root box [mn, mx]) to hs.qnodes.
static Grid quantize_nodes(HostScene& hs, const std::vector<BvhNode>& built, const float* mn, const float* mx, unsigned threads) {
    const Grid g = make_grid(mn, mx);
    const size_t first = hs.qnodes.size();
    hs.qnodes.resize(first + built.size());
    parallel_for((uint32_t)built.size(), threads, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; ++i) {
            const BvhNode& n = built[i];
            QNode& q = hs.qnodes[first + i];
            q.left = device_ref(n.left);
            q.right = device_ref(n.right);
            quantize_box(g, n.lmin, n.lmax, q.lbox);
            quantize_box(g, n.rmin, n.rmax, q.rbox);
        }
    });
    return g;
}

}  // namespace

// Reserves nodes and triangle slots for a BVH the GPU will build; returns its root box in mn/mx.
static void defer_build(HostScene& hs, std::vector<BuildPrim>& prims, std::vector<uint32_t>& tri_order, uint32_t budget,
                        int mesh, float* mn, float* mx, int32_t& root, Grid& grid, unsigned threads) {
    DeferredBuild db;
    db.node_base = (uint32_t)hs.qnodes.size();
    db.tri_base = (uint32_t)tri_order.size();
    db.budget = budget;
    db.mesh = mesh;
    // root box: per-piece boxes, then their union
    const uint32_t n = (uint32_t)prims.size();
    constexpr unsigned MAX_PIECES = 64;
    float pmn[MAX_PIECES][3], pmx[MAX_PIECES][3];
    for (unsigned k = 0; k < MAX_PIECES; ++k)
        for (int a = 0; a < 3; ++a) { pmn[k][a] = FLT_MAX; pmx[k][a] = -FLT_MAX; }
    std::atomic<unsigned> piece{0};
    parallel_for(n, threads, [&](uint32_t lo, uint32_t hi) {
        const unsigned k = piece.fetch_add(1) % MAX_PIECES;  // parallel_for never runs more than 64 pieces
        float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) { lmn[a] = std::min(lmn[a], prims[i].bmin[a]); lmx[a] = std::max(lmx[a], prims[i].bmax[a]); }
        for (int a = 0; a < 3; ++a) { pmn[k][a] = lmn[a]; pmx[k][a] = lmx[a]; }
    });
    for (int a = 0; a < 3; ++a) { mn[a] = FLT_MAX; mx[a] = -FLT_MAX; }
    for (unsigned k = 0; k < MAX_PIECES; ++k)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], pmn[k][a]); mx[a] = std::max(mx[a], pmx[k][a]); }
    // triangles keep their submission order (the GPU builder permutes the records itself); a triangle's `index`
    // becomes its ordinal among this BVH's triangles
    uint32_t ordinal = 0;
    const size_t order0 = tri_order.size();
    size_t n_tris = 0;
    for (const BuildPrim& p : prims) n_tris += p.kind == KIND_TRI ? 1u : 0u;
    tri_order.resize(order0 + n_tris);
    for (BuildPrim& p : prims) {
        if (p.kind == KIND_TRI) {
            tri_order[order0 + ordinal] = p.index;
            p.index = ordinal++;
        }
    }
    db.tri_count = ordinal;
    db.grid = grid = make_grid(mn, mx);
    QNode blank;
    std::memset(&blank, 0, sizeof blank);
    hs.qnodes.resize(hs.qnodes.size() + prims.size(), blank);  // the range the GPU builder fills (uploaded blank)
    root = (int32_t)db.node_base;  // the radix tree's root is its first inner node
    db.prims.swap(prims);
    hs.deferred.push_back(std::move(db));
}

// RRT_BUILD_TRACE=1: the phases of a scene build with their host times, on stderr.
struct PhaseTrace {
    bool on = std::getenv("RRT_BUILD_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rrt build] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

rrt_status build_host_scene(const rrt_scene_desc* d, HostScene& hs, std::string& err, unsigned threads,
                            uint32_t device_build_min) {
    PhaseTrace trace;
    hs = HostScene();
    if (!d) { err = "scene is NULL"; return RRT_ERR_INVALID; }
    if (d->material_count == 0 || !d->materials) { err = "scene has no materials"; return RRT_ERR_INVALID; }
    if ((d->triangle_count && !d->triangles) || (d->sphere_count && !d->spheres) || (d->rect_count && !d->rects) ||
        (d->cuboid_count && !d->cuboids) || (d->instance_count && !d->instances) || (d->mesh_count && !d->meshes) ||
        (d->mesh_object_count && !d->mesh_objects) || (d->world_count && !d->world) || (d->light_count && !d->lights)) {
        err = "scene array pointer is NULL with a non-zero count";
        return RRT_ERR_INVALID;
    }

    // scene scale -> conservative padding of every bounding box (beyond the reference's own 1e-4 pads)
    float scale = max_abs3(d->camera.origin);
    {
        std::atomic<uint32_t> bits{0};  // max of non-negative floats == max of their bit patterns (+inf included)
        parallel_for(d->triangle_count, threads, [&](uint32_t lo, uint32_t hi) {
            float m = 0.f;
            for (uint32_t i = lo; i < hi; ++i) {
                const rrt_triangle& t = d->triangles[i];
                m = std::max(m, std::max(max_abs3(t.v0), std::max(max_abs3(t.v1), max_abs3(t.v2))));
            }
            uint32_t b;
            std::memcpy(&b, &m, 4);
            uint32_t seen = bits.load();
            while (b > seen && !bits.compare_exchange_weak(seen, b)) {}
        });
        const uint32_t b = bits.load();
        float m;
        std::memcpy(&m, &b, 4);
        scale = std::max(scale, m);
    }
    for (uint32_t i = 0; i < d->sphere_count; ++i) {
        if (!std::isfinite(d->spheres[i].radius)) scale = INFINITY;
        scale = std::max(scale, max_abs3(d->spheres[i].center) + std::fabs(d->spheres[i].radius));
    }
    for (uint32_t i = 0; i < d->rect_count; ++i) {
        const rrt_rect& r = d->rects[i];
        const float v[5] = {r.a0, r.a1, r.b0, r.b1, r.k};
        for (float f : v) scale = std::isfinite(f) ? std::max(scale, std::fabs(f)) : INFINITY;
    }
    for (uint32_t i = 0; i < d->cuboid_count; ++i) scale = std::max(scale, std::max(max_abs3(d->cuboids[i].min), max_abs3(d->cuboids[i].max)));
    float inst_scale = 0.f;
    for (uint32_t i = 0; i < d->instance_count; ++i) inst_scale = std::max(inst_scale, max_abs3(d->instances[i].offset));
    scale = scale * 1.5f + inst_scale;
    if (!(scale < 1e30f)) { err = "scene coordinates are not finite"; return RRT_ERR_INVALID; }
    trace.mark("scene extent");

    Expander ex;
    ex.d = d;
    ex.threads = threads;
    ex.hs = &hs;
    ex.err = &err;
    ex.pad = scale * (1.0f / 131072.0f);  // 2^-17 relative to the scene's extent

    for (uint32_t i = 0; i < d->material_count; ++i) {
        const rrt_material& m = d->materials[i];
        if (m.kind > 3u) { err = "unknown material kind"; return RRT_ERR_INVALID; }
        MatRec r;
        std::memset(&r, 0, sizeof r);
        r.kind = m.kind; r.r = m.rgb[0]; r.g = m.rgb[1]; r.b = m.rgb[2]; r.param = m.param;
        hs.mats.push_back(r);
    }
    for (uint32_t i = 0; i < d->light_count; ++i) {
        const rrt_light& l = d->lights[i];
        if (l.type > 2u) { err = "unknown light type"; return RRT_ERR_INVALID; }
        LightRec r;
        std::memset(&r, 0, sizeof r);
        r.type = l.type; r.p0 = l.p[0]; r.p1 = l.p[1]; r.p2 = l.p[2]; r.p3 = l.p[3]; r.p4 = l.p[4];
        hs.lights.push_back(r);
    }

    std::vector<uint32_t> tri_order;
    // depth budget: bottom levels get what they need for a balanced fallback (+ slack), the top level the rest
    uint32_t top_budget = MAX_TREE_DEPTH;
    std::vector<Expander::MeshInfo> meshes(d->mesh_count);
    uint32_t deepest_mesh = 0;
    if (d->mesh_count) {
        // first pass: primitive counts, to split the depth budget
        uint32_t world_prims = 0;
        for (uint32_t i = 0; i < d->world_count; ++i) world_prims += d->world[i].type == RRT_OBJ_CUBOIDS ? 6u * d->world[i].count : d->world[i].count;
        uint32_t need_top = 0;
        while ((1ull << need_top) < std::max(world_prims, 1u)) ++need_top;
        need_top += 2;  // slack for the SAH to be unbalanced near the top
        if (need_top + 6 > MAX_TREE_DEPTH) { err = "too many top-level objects for an instanced scene"; return RRT_ERR_INVALID; }
        top_budget = need_top;
    }
    for (uint32_t mi = 0; mi < d->mesh_count; ++mi) {
        const rrt_mesh& m = d->meshes[mi];
        if ((uint64_t)m.first_object + m.object_count > d->mesh_object_count) { err = "mesh object range out of range"; return RRT_ERR_INVALID; }
        std::vector<BuildPrim> prims;
        uint32_t id = 0;
        if (!ex.expand(d->mesh_objects + m.first_object, m.object_count, id, prims, nullptr)) return RRT_ERR_INVALID;
        const uint32_t budget = MAX_TREE_DEPTH - top_budget - 1u;
        if (device_build_min && prims.size() >= device_build_min) {
            defer_build(hs, prims, tri_order, budget, (int)mi, meshes[mi].mn, meshes[mi].mx, meshes[mi].root, meshes[mi].grid, threads);
            meshes[mi].prim_count = id;
            meshes[mi].depth = 0;
            continue;
        }
        std::vector<BvhNode> built;
        BuildResult br = build_bvh(prims, built, tri_order, (uint32_t)tri_order.size(), budget, threads, (int64_t)hs.qnodes.size());
        meshes[mi].grid = quantize_nodes(hs, built, br.root_min, br.root_max, threads);
        if (br.depth > budget) { err = "mesh too large for the traversal stack of an instanced scene"; return RRT_ERR_INVALID; }
        meshes[mi].root = br.root;
        meshes[mi].prim_count = id;
        meshes[mi].depth = br.depth;
        for (int a = 0; a < 3; ++a) { meshes[mi].mn[a] = br.root_min[a]; meshes[mi].mx[a] = br.root_max[a]; }
        if (prims.empty()) for (int a = 0; a < 3; ++a) { meshes[mi].mn[a] = 0.f; meshes[mi].mx[a] = 0.f; }
        deepest_mesh = std::max(deepest_mesh, br.depth);
    }

    trace.mark("meshes");
    std::vector<BuildPrim> prims;
    uint32_t id = 0;
    if (!ex.expand(d->world, d->world_count, id, prims, &meshes)) return RRT_ERR_INVALID;
    trace.mark("world list expanded");
    hs.prim_count = id;
    hs.deepest_mesh = deepest_mesh;
    if (device_build_min && prims.size() >= device_build_min) {
        float mn[3], mx[3];
        defer_build(hs, prims, tri_order, top_budget, -1, mn, mx, hs.root, hs.grid, threads);
        hs.top_depth = 0;
        hs.depth = 0;
    } else {
        std::vector<BvhNode> built;
        BuildResult br = build_bvh(prims, built, tri_order, (uint32_t)tri_order.size(), top_budget, threads, (int64_t)hs.qnodes.size());
        hs.grid = quantize_nodes(hs, built, br.root_min, br.root_max, threads);
        if (br.depth + (hs.insts.empty() ? 0u : deepest_mesh + 1u) > MAX_TREE_DEPTH) {
            err = "scene too deep for the traversal stack";
            return RRT_ERR_INVALID;
        }
        hs.root = br.root;
        hs.top_depth = br.depth;
        hs.depth = br.depth + (hs.insts.empty() ? 0u : deepest_mesh);
    }
    trace.mark("top-level BVH (or deferral)");
    hs.has_instances = !hs.insts.empty();
    if (tri_order.size() > REF_OFFSET_MASK) { err = "too many triangles"; return RRT_ERR_INVALID; }
    if (hs.qnodes.size() > (size_t)REF_OFFSET_MASK) { err = "too many BVH nodes"; return RRT_ERR_INVALID; }  // device_ref: byte offsets below 2^30

    hs.tris.resize(tri_order.size());
    parallel_for((uint32_t)tri_order.size(), threads, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t k = lo; k < hi; ++k) hs.tris[k] = ex.tri_src[tri_order[k]];
    });

    trace.mark("triangle records in leaf order");
    const rrt_camera& c = d->camera;
    for (int a = 0; a < 3; ++a) {
        hs.cam.c0[a] = c.c2w[a]; hs.cam.c1[a] = c.c2w[3 + a]; hs.cam.c2[a] = c.c2w[6 + a]; hs.cam.tr[a] = c.c2w[9 + a];
        hs.cam.origin[a] = c.origin[a];
        hs.background[a] = d->background[a];
    }
    hs.cam.scale = c.scale; hs.cam.aspect = c.aspect; hs.cam.time0 = c.time0; hs.cam.time1 = c.time1;
    return RRT_OK;
}

}  // namespace rrt
