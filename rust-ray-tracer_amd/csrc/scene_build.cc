// scene_build.cc -- scene description -> primitive records with reference bounding boxes -> flat BVH.
#include "scene_build.h"

#include <algorithm>
#include <array>
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
        // for unrolling (instances written out as world-space trees over the mesh's leaves): the leaves of the
        // host-built tree (reference words: kind, count and first record), its node count and depth budget, and the
        // mesh's range of the leaf-ordered triangle records
        std::vector<uint32_t> leaves;
        uint32_t node_count = 0, budget = 0;
        uint32_t tri_base = 0, tri_count = 0;
        bool host_built = false;
    };
    // Unrolled scenes: world box of the primitives of each instance OBJECT of the world list (indexed like hs->insts:
    // one record per object, in world order -- two objects may name the same rrt_instance), computed before the world
    // list is expanded; empty when the scene is not unrolled.
    std::vector<std::array<float, 6>> inst_world_box;
    std::vector<uint32_t> inst_of_record;  // per InstRec pushed: the index of its rrt_instance
    // Top-level triangle runs left to the GPU builder (records and boxes are made there): the runs, their triangle
    // count and the extent of their vertices.
    bool raw_triangles = false;
    std::vector<RawTriRun> raw_runs;
    uint32_t raw_tris = 0;
    float raw_mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, raw_mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    bool expand(const rrt_object* objs, uint32_t n_objs, uint32_t& id, std::vector<BuildPrim>& prims,
                const std::vector<MeshInfo>* meshes) {
        {   // one growth of the big arrays instead of one per run (a scene may list its triangles in many runs)
            size_t n_tris = 0, n_other = 0;
            for (uint32_t oi = 0; oi < n_objs; ++oi) {
                if (objs[oi].type == RRT_OBJ_TRIANGLES) n_tris += (raw_triangles && meshes) ? 0u : objs[oi].count;
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
                    if (raw_triangles && meshes) {
                        // left to the GPU builder: the host only checks the material indices and takes the vertices' extent
                        std::atomic<bool> bad{false};
                        constexpr unsigned MAX_PIECES = 64;
                        float pmn[MAX_PIECES][3], pmx[MAX_PIECES][3];
                        for (unsigned k = 0; k < MAX_PIECES; ++k)
                            for (int a = 0; a < 3; ++a) { pmn[k][a] = FLT_MAX; pmx[k][a] = -FLT_MAX; }
                        std::atomic<unsigned> piece{0};
                        parallel_for(ob.count, threads, [&](uint32_t lo, uint32_t hi) {
                            const unsigned k = piece.fetch_add(1) % MAX_PIECES;
                            float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                            for (uint32_t i = lo; i < hi; ++i) {
                                const rrt_triangle& t = d->triangles[ob.first + i];
                                if (t.material >= d->material_count) bad.store(true, std::memory_order_relaxed);
                                for (int a = 0; a < 3; ++a) {
                                    lmn[a] = std::min(lmn[a], std::min(t.v0[a], std::min(t.v1[a], t.v2[a])));
                                    lmx[a] = std::max(lmx[a], std::max(t.v0[a], std::max(t.v1[a], t.v2[a])));
                                }
                            }
                            for (int a = 0; a < 3; ++a) { pmn[k][a] = std::min(pmn[k][a], lmn[a]); pmx[k][a] = std::max(pmx[k][a], lmx[a]); }
                        });
                        if (bad.load()) return fail("triangle material index out of range");
                        for (unsigned k = 0; k < MAX_PIECES; ++k)
                            for (int a = 0; a < 3; ++a) { raw_mn[a] = std::min(raw_mn[a], pmn[k][a]); raw_mx[a] = std::max(raw_mx[a], pmx[k][a]); }
                        raw_runs.push_back(RawTriRun{d->triangles + ob.first, ob.count, id});
                        raw_tris += ob.count;
                        id += ob.count;
                        break;
                    }
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
                    if (!inst_world_box.empty()) {  // unrolled: the padded world box of the instance's own primitives
                        const std::array<float, 6>& wb = inst_world_box[hs->insts.size()];
                        for (int a = 0; a < 3; ++a) { p.bmin[a] = wb[a]; p.bmax[a] = wb[3 + a]; }
                    } else
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
                    if (inst_world_box.empty()) {
                        widen(p);
                        widen(p);
                    }
                    inst_of_record.push_back(ob.first);
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

// ---- unrolled instances -----------------------------------------------------------------------------------------
// An instance (Translate(RotateY(mesh)), reference src/geometry/transform.rs) can be written out as a tree of its own
// whose boxes are in world space, hung into the top-level tree in place of the instance leaf.  Traversal then never
// changes space: only the triangle test does, on the mesh's object-space records, with the ray moved into the
// instance's space exactly as the reference moves it.  The copy keeps the mesh tree's LEAVES (runs of the mesh's
// leaf-ordered records, which all instances of the mesh share); the levels above them are built anew per instance by the
// SAH builder over the leaves' world boxes: the splits the mesh's own tree chose are good along the mesh's axes and
// poor along the world's once the mesh is rotated (kept as they are, with refitted boxes, the copies cost 10 % more
// node visits and 18 % more triangle tests than the two-level walk).

// Padded world box of the triangle the reference's test sees for record `r` (v0, v0 + e1, v0 + e2: triangle.rs:17-18
// caches the edges) under the instance's point map x' = cos x + sin z, z' = -sin x + cos z, then + offset
// (transform.rs:27,113-124).  The pad is the builder's 2^-17 of the scene's extent; the reference's own 1e-4
// (triangle.rs:82-94: it keeps an axis-aligned triangle's box from being flat) is not carried over: these boxes only have
// to contain what the triangle test accepts.  What the pad has to cover besides the slab test's own error (1/16 of it):
// the ray is moved into object space in f32, so the point the test accepts there, moved back to world space, lies
// off the world ray by the rounding of that move, at most about 8e-7 of the extent (a tenth of the pad).
static void world_tri_box(const TriRec& r, const rrt_instance& in, float pad, float* mn, float* mx) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int k = 0; k < 3; ++k) {
        double p[3];
        for (int a = 0; a < 3; ++a) p[a] = (double)r.v0[a] + (k == 1 ? (double)r.e1[a] : (k == 2 ? (double)r.e2[a] : 0.0));
        const double w[3] = {(double)in.cos_theta * p[0] + (double)in.sin_theta * p[2] + (double)in.offset[0], p[1] + (double)in.offset[1],
                             -(double)in.sin_theta * p[0] + (double)in.cos_theta * p[2] + (double)in.offset[2]};
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], w[a]); hi[a] = std::max(hi[a], w[a]); }
    }
    for (int a = 0; a < 3; ++a) {
        mn[a] = std::nextafterf((float)lo[a], -INFINITY) - pad;
        mx[a] = std::nextafterf((float)hi[a], INFINITY) + pad;
    }
}

// The same for the other primitives a mesh may hold.  A sphere stays a sphere under the instance's rotation: its world
// box is the mapped centre +- |radius| (sphere.rs:128-133).  A rectangle's world box is the box of the eight mapped
// corners of its object-space box, the reference's 1e-4 on the thin axis included (aarect.rs:60-66,115-121,191-197: the
// accepted point's coordinate on that axis is o + t * d, rounded, not k itself).
static void world_box_of_corners(const double* lo_o, const double* hi_o, const rrt_instance& in, float pad, float* mn, float* mx) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int c = 0; c < 8; ++c) {
        const double p[3] = {(c & 1) ? hi_o[0] : lo_o[0], (c & 2) ? hi_o[1] : lo_o[1], (c & 4) ? hi_o[2] : lo_o[2]};
        const double w[3] = {(double)in.cos_theta * p[0] + (double)in.sin_theta * p[2] + (double)in.offset[0], p[1] + (double)in.offset[1],
                             -(double)in.sin_theta * p[0] + (double)in.cos_theta * p[2] + (double)in.offset[2]};
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], w[a]); hi[a] = std::max(hi[a], w[a]); }
    }
    for (int a = 0; a < 3; ++a) {
        mn[a] = std::nextafterf((float)lo[a], -INFINITY) - pad;
        mx[a] = std::nextafterf((float)hi[a], INFINITY) + pad;
    }
}
static void world_sphere_box(const SphereRec& r, const rrt_instance& in, float pad, float* mn, float* mx) {
    const double c[3] = {r.c[0], r.c[1], r.c[2]};
    const double w[3] = {(double)in.cos_theta * c[0] + (double)in.sin_theta * c[2] + (double)in.offset[0], c[1] + (double)in.offset[1],
                         -(double)in.sin_theta * c[0] + (double)in.cos_theta * c[2] + (double)in.offset[2]};
    const double ar = std::fabs((double)r.r) * (1.0 + 1e-6);
    for (int a = 0; a < 3; ++a) {
        mn[a] = std::nextafterf((float)(w[a] - ar), -INFINITY) - pad;
        mx[a] = std::nextafterf((float)(w[a] + ar), INFINITY) + pad;
    }
}
static void world_rect_box(const RectRec& r, const rrt_instance& in, float pad, float* mn, float* mx) {
    const uint32_t axis = r.axis_flags & 3u, ia = axis == 0 ? 1u : 0u, ib = axis == 2 ? 1u : 2u;
    double lo[3], hi[3];
    lo[axis] = (double)r.k - 0.0001; hi[axis] = (double)r.k + 0.0001;
    lo[ia] = std::min(r.a0, r.a1); hi[ia] = std::max(r.a0, r.a1);
    lo[ib] = std::min(r.b0, r.b1); hi[ib] = std::max(r.b0, r.b1);
    world_box_of_corners(lo, hi, in, pad, mn, mx);
}

// The tree of one instance: `leaves` are the mesh's leaves (reference words: kind, count and first record);
// `prims` receives one primitive per leaf with the world box of what the leaf holds (also the instance's world box, in
// mn/mx).  `rec(i)`: triangle record i of the scene's leaf-ordered record array; spheres and rectangles are read
// from the host scene's record arrays.
template <typename RecFn>
static void instance_leaf_prims(const std::vector<uint32_t>& leaves, const rrt_instance& in, float pad, RecFn rec, const HostScene& hs,
                                std::vector<BuildPrim>& prims, float* mn, float* mx) {
    prims.resize(leaves.size());
    for (int a = 0; a < 3; ++a) { mn[a] = FLT_MAX; mx[a] = -FLT_MAX; }
    for (size_t k = 0; k < leaves.size(); ++k) {
        const uint32_t kind = (leaves[k] >> 28) & 7u, first = leaves[k] & REF_OFFSET_MASK, cnt = kind == KIND_TRI ? ((leaves[k] >> 25) & 7u) + 1u : 1u;
        BuildPrim& p = prims[k];
        for (int a = 0; a < 3; ++a) { p.bmin[a] = FLT_MAX; p.bmax[a] = -FLT_MAX; }
        for (uint32_t j = 0; j < cnt; ++j) {
            float tmn[3], tmx[3];
            if (kind == KIND_TRI) world_tri_box(rec(first + j), in, pad, tmn, tmx);
            else if (kind == KIND_SPHERE) world_sphere_box(hs.spheres[first], in, pad, tmn, tmx);
            else world_rect_box(hs.rects[first], in, pad, tmn, tmx);
            for (int a = 0; a < 3; ++a) { p.bmin[a] = std::min(p.bmin[a], tmn[a]); p.bmax[a] = std::max(p.bmax[a], tmx[a]); }
        }
        p.kind = KIND_SUBTREE;
        p.index = (uint32_t)k;
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p.bmin[a]); mx[a] = std::max(mx[a], p.bmax[a]); }
    }
}

// Builds the instance's tree over `prims` (see above) and writes it, on grid g, to out[0 .. nodes): node k becomes
// node copy_base + k, the root first.  Triangle leaves carry (slot << shift) | first record.  Returns the tree's node
// levels, or 0 if it came out with another node count than `nodes` (= leaves - 1, or 1 for a single leaf).
static uint32_t build_instance_tree(std::vector<BuildPrim>& prims, const std::vector<uint32_t>& leaves, uint32_t slot, uint32_t shift,
                                    uint32_t copy_base, uint32_t nodes, uint32_t budget, const Grid& g, QNode* out) {
    std::vector<BvhNode> built;
    std::vector<uint32_t> no_tris;
    const BuildResult br = build_bvh(prims, built, no_tris, 0u, budget, 1u, (int64_t)copy_base);
    if (built.size() != (size_t)nodes || br.root != (int32_t)copy_base) return 0u;
    auto final_ref = [&](int32_t ref) -> int32_t {
        if (ref >= 0) return device_ref(ref);
        const uint32_t u = (uint32_t)ref;
        if (((u >> 28) & 7u) != KIND_SUBTREE) return ref;  // an empty child slot
        const uint32_t leaf = leaves[u & REF_OFFSET_MASK];
        return (int32_t)((leaf & ~REF_OFFSET_MASK) | (slot << shift) | (leaf & REF_OFFSET_MASK));
    };
    for (uint32_t k = 0; k < nodes; ++k) {
        const BvhNode& nd = built[k];
        QNode& q = out[k];
        q.left = final_ref(nd.left);
        q.right = final_ref(nd.right);
        quantize_box(g, nd.lmin, nd.lmax, q.lbox);
        quantize_box(g, nd.rmin, nd.rmax, q.rbox);
    }
    return br.depth;
}

// Reserves nodes and triangle slots for a BVH the GPU will build; returns its root box in mn/mx.
// raw_runs (top level only, may be null): triangle runs the host has not expanded; they join the BVH as its last
// primitives and triangle ordinals, their node range and records exist on the device only (hs.tail_nodes / tail_tris),
// and the extent of their vertices [raw_mn, raw_mx] enters the root box with the pads every triangle box carries.
static void defer_build(HostScene& hs, std::vector<BuildPrim>& prims, std::vector<uint32_t>& tri_order, uint32_t budget,
                        int mesh, float* mn, float* mx, int32_t& root, Grid& grid, unsigned threads,
                        std::vector<RawTriRun>* raw_runs = nullptr, uint32_t raw_tris = 0, const float* raw_mn = nullptr,
                        const float* raw_mx = nullptr, float pad = 0.f) {
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
    if (raw_runs && raw_tris) {
        for (int a = 0; a < 3; ++a) {  // the same two steps as a triangle's own box: tri_box(), then widen()
            mn[a] = std::min(mn[a], (raw_mn[a] - 0.0001f) - pad);
            mx[a] = std::max(mx[a], (raw_mx[a] + 0.0001f) + pad);
        }
        db.raw_runs.swap(*raw_runs);
        db.raw_tris = raw_tris;
        db.pad = pad;
        db.tri_count += raw_tris;
        hs.tail_nodes = (uint32_t)prims.size() + raw_tris;  // this BVH's node range: on the device only
        hs.tail_tris = raw_tris;
    }
    db.grid = grid = make_grid(mn, mx);
    if (!db.raw_tris) {
        QNode blank;
        std::memset(&blank, 0, sizeof blank);
        hs.qnodes.resize(hs.qnodes.size() + prims.size(), blank);  // the range the GPU builder fills (uploaded blank)
    }
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
                            uint32_t device_build_min, bool unroll) {
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
        meshes[mi].tri_base = (uint32_t)tri_order.size();
        BuildResult br = build_bvh(prims, built, tri_order, (uint32_t)tri_order.size(), budget, threads, (int64_t)hs.qnodes.size());
        meshes[mi].tri_count = (uint32_t)tri_order.size() - meshes[mi].tri_base;
        meshes[mi].grid = quantize_nodes(hs, built, br.root_min, br.root_max, threads);
        if (br.depth > budget) { err = "mesh too large for the traversal stack of an instanced scene"; return RRT_ERR_INVALID; }
        meshes[mi].host_built = true;
        meshes[mi].node_count = (uint32_t)built.size();
        meshes[mi].budget = budget;
        for (const BvhNode& nd : built)
            for (int32_t ref : {nd.left, nd.right})
                if (ref < 0 && (((uint32_t)ref >> 28) & 7u) <= KIND_RECT) meshes[mi].leaves.push_back((uint32_t)ref);  // triangles, a sphere, a rectangle
        meshes[mi].root = br.root;
        meshes[mi].prim_count = id;
        meshes[mi].depth = br.depth;
        for (int a = 0; a < 3; ++a) { meshes[mi].mn[a] = br.root_min[a]; meshes[mi].mx[a] = br.root_max[a]; }
        if (prims.empty()) for (int a = 0; a < 3; ++a) { meshes[mi].mn[a] = 0.f; meshes[mi].mx[a] = 0.f; }
        deepest_mesh = std::max(deepest_mesh, br.depth);
    }

    trace.mark("meshes");
    // Unroll the instances into one world-space tree?  Only when every instanced mesh was built here, the copies stay
    // within a node budget (RRT_UNROLL_MAX_NODES, default 2^24 = 512 MB of nodes), and a leaf's 25-bit field has room
    // for (instance slot, first record) -- for triangle, sphere and rectangle records alike.
    uint32_t unroll_shift = 0;
    std::vector<uint32_t> inst_objs;  // the rrt_instance of every instance object of the world list, in world order
    if (unroll && d->instance_count > 0u) {
        bool ok = true;
        uint64_t copy_nodes = 0, world_tris = 0;
        for (uint32_t i = 0; i < d->world_count && ok; ++i) {
            const rrt_object& ob = d->world[i];
            if (ob.type == RRT_OBJ_TRIANGLES) world_tris += ob.count;
            if (ob.type != RRT_OBJ_INSTANCE) continue;
            if (ob.count != 1u || ob.first >= d->instance_count || d->instances[ob.first].mesh >= d->mesh_count) { ok = false; break; }  // reported by expand()
            inst_objs.push_back(ob.first);
            const rrt_instance& in = d->instances[ob.first];
            const Expander::MeshInfo& mi = meshes[in.mesh];
            // (the world boxes take the instance's point map for the inverse of its ray map, which it is for a rotation:
            // an instance whose sin / cos pair is not one keeps the two-level walk, which makes no such assumption)
            const double norm = (double)in.sin_theta * in.sin_theta + (double)in.cos_theta * in.cos_theta;
            ok = mi.host_built && !mi.leaves.empty() && std::fabs(norm - 1.0) < 1e-5;
            copy_nodes += mi.node_count;
        }
        uint64_t max_nodes = 1ull << 24;
        if (const char* e = std::getenv("RRT_UNROLL_MAX_NODES")) max_nodes = (uint64_t)std::max(0ll, std::atoll(e));
        // (spheres and rectangles: an upper bound on the records the world list and the meshes will have made)
        uint64_t other_records = 0;
        for (uint32_t i = 0; i < d->world_count; ++i) if (d->world[i].type != RRT_OBJ_TRIANGLES && d->world[i].type != RRT_OBJ_INSTANCE) other_records += d->world[i].type == RRT_OBJ_CUBOIDS ? 6ull * d->world[i].count : d->world[i].count;
        for (uint32_t i = 0; i < d->mesh_object_count; ++i) if (d->mesh_objects[i].type != RRT_OBJ_TRIANGLES) other_records += d->mesh_objects[i].type == RRT_OBJ_CUBOIDS ? 6ull * d->mesh_objects[i].count : d->mesh_objects[i].count;
        const uint64_t records = std::max((uint64_t)tri_order.size() + world_tris + MAX_LEAF_TRIS, other_records + 1u);
        uint32_t shift = 1;
        while ((1ull << shift) < records) ++shift;
        if (ok && copy_nodes > 0u && copy_nodes <= max_nodes && hs.qnodes.size() + copy_nodes + 2u * d->world_count + 64u < (uint64_t)REF_OFFSET_MASK &&
            shift < 25u && (uint64_t)inst_objs.size() < (1ull << (25u - shift)))  // slots 1 .. objects, slot 0 = not instanced
            unroll_shift = shift;
    }
    // per instance: one primitive per leaf of its mesh, with the world box of the leaf's triangles (kept for the build
    // of the instance's tree, which has to wait for the top level's grid)
    std::vector<std::vector<BuildPrim>> inst_prims;
    auto rec_of = [&](uint32_t i) -> const TriRec& { return ex.tri_src[tri_order[i]]; };
    if (unroll_shift) {
        // (per instance OBJECT, not per rrt_instance: the builds below reorder their primitives in place, and two
        // objects that name the same instance each get a tree and a slot of their own)
        const uint32_t n_objs = (uint32_t)inst_objs.size();
        ex.inst_world_box.resize(n_objs);
        inst_prims.resize(n_objs);
        std::vector<std::thread> pool;
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (uint32_t r; (r = next.fetch_add(1)) < n_objs;) {
                const rrt_instance& in = d->instances[inst_objs[r]];
                std::array<float, 6>& b = ex.inst_world_box[r];
                b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (meshes[in.mesh].leaves.empty()) continue;
                instance_leaf_prims(meshes[in.mesh].leaves, in, ex.pad, rec_of, hs, inst_prims[r], b.data(), b.data() + 3);
            }
        };
        for (unsigned t = 1; t < std::min<unsigned>(threads, n_objs); ++t) pool.emplace_back(work);
        work();
        for (std::thread& t : pool) t.join();
        trace.mark("instance leaf boxes");
    }
    std::vector<BuildPrim> prims;
    uint32_t id = 0;
    {
        // a top level the GPU builder will build leaves its triangle runs unexpanded: the GPU makes records and boxes
        uint64_t world_prims = 0, world_tris = 0;
        for (uint32_t i = 0; i < d->world_count; ++i) {
            world_prims += d->world[i].type == RRT_OBJ_CUBOIDS ? 6ull * d->world[i].count : d->world[i].count;
            if (d->world[i].type == RRT_OBJ_TRIANGLES) world_tris += d->world[i].count;
        }
        ex.raw_triangles = device_build_min != 0u && world_prims >= device_build_min && world_tris > 0u && world_prims <= REF_OFFSET_MASK &&
                           std::getenv("RRT_HOST_EXPAND") == nullptr;
    }
    if (!ex.expand(d->world, d->world_count, id, prims, &meshes)) return RRT_ERR_INVALID;
    trace.mark("world list expanded");
    hs.prim_count = id;
    hs.deepest_mesh = deepest_mesh;
    if (device_build_min && prims.size() + ex.raw_tris >= device_build_min) {
        float mn[3], mx[3];
        defer_build(hs, prims, tri_order, top_budget, -1, mn, mx, hs.root, hs.grid, threads, &ex.raw_runs, ex.raw_tris, ex.raw_mn,
                    ex.raw_mx, ex.pad);
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
        if (unroll_shift && !hs.insts.empty()) {
            // one world-space tree per instance, on the top level's grid, behind the nodes there are; then every
            // instance leaf of the top level becomes a reference to its instance's root
            const size_t top_first = hs.qnodes.size() - built.size(), top_end = hs.qnodes.size();
            std::vector<uint32_t> copy_base(hs.insts.size());
            size_t total = top_end;
            for (size_t r = 0; r < hs.insts.size(); ++r) {
                copy_base[r] = (uint32_t)total;
                total += meshes[d->instances[ex.inst_of_record[r]].mesh].node_count;
            }
            hs.qnodes.resize(total);
            std::vector<std::thread> pool;
            std::atomic<uint32_t> next{0}, deepest{0};
            std::atomic<bool> failed{false};
            auto work = [&]() {
                for (uint32_t r; (r = next.fetch_add(1)) < hs.insts.size();) {
                    const uint32_t i = ex.inst_of_record[r];
                    const Expander::MeshInfo& mi = meshes[d->instances[i].mesh];
                    const uint32_t depth = build_instance_tree(inst_prims[r], mi.leaves, r + 1u, unroll_shift, copy_base[r], mi.node_count,
                                                               mi.budget, hs.grid, hs.qnodes.data() + copy_base[r]);
                    if (depth == 0u || depth > mi.budget) failed.store(true);
                    uint32_t seen = deepest.load();
                    while (depth > seen && !deepest.compare_exchange_weak(seen, depth)) {}
                    std::vector<BuildPrim>().swap(inst_prims[r]);
                }
            };
            for (unsigned t = 1; t < std::min<unsigned>(threads, (unsigned)hs.insts.size()); ++t) pool.emplace_back(work);
            work();
            for (std::thread& t : pool) t.join();
            if (failed.load()) { err = "instance tree does not fit the traversal stack"; return RRT_ERR_INVALID; }
            for (size_t k = top_first; k < top_end; ++k) {
                int32_t* refs[2] = {&hs.qnodes[k].left, &hs.qnodes[k].right};
                for (int32_t* ref : refs) {
                    const uint32_t u = (uint32_t)*ref;
                    if (*ref < 0 && ((u >> 28) & 7u) == KIND_INST) *ref = device_ref((int32_t)copy_base[u & REF_OFFSET_MASK]);
                }
            }
            hs.unroll_shift = unroll_shift;
            hs.unrolled_instances = (uint32_t)hs.insts.size();
            hs.deepest_mesh = std::max(hs.deepest_mesh, deepest.load());
            hs.depth = br.depth + hs.deepest_mesh;
            trace.mark("instances unrolled");
        }
    }
    trace.mark("top-level BVH (or deferral)");
    hs.has_instances = !hs.insts.empty() && hs.unroll_shift == 0u;
    if (tri_order.size() + (size_t)hs.tail_tris > REF_OFFSET_MASK) { err = "too many triangles"; return RRT_ERR_INVALID; }
    if (hs.qnodes.size() + (size_t)hs.tail_nodes > (size_t)REF_OFFSET_MASK) { err = "too many BVH nodes"; return RRT_ERR_INVALID; }  // device_ref: byte offsets below 2^30

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
