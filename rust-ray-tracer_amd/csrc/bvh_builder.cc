// bvh_builder.cc -- binned-SAH BVH2 over mixed primitives, depth-bounded, built on host threads.
#include "bvh_builder.h"

#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <future>
#include <thread>
#include <vector>

namespace rrt {
namespace {

#ifndef RRT_SAH_BINS
#define RRT_SAH_BINS 16
#endif
constexpr int kBins = RRT_SAH_BINS;
// experiment knobs (read once): RRT_LEAF_TRIS, RRT_SAH_TCOST
static uint32_t env_u32(const char* name, uint32_t def, uint32_t lo, uint32_t hi) {
    const char* e = std::getenv(name);
    if (!e) return def;
    const long v = std::atol(e);
    return (uint32_t)std::min<long>(std::max<long>(v, lo), hi);
}
static float env_f32(const char* name, float def) {
    const char* e = std::getenv(name);
    return e ? (float)std::atof(e) : def;
}
const uint32_t kLeafTris = env_u32("RRT_LEAF_TRIS", 4u, 1u, MAX_LEAF_TRIS);  // SAH may stop at this many triangles per leaf
const float kTraversalCost = env_f32("RRT_SAH_TCOST", 1.0f);  // cost of one node visit relative to one primitive test

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { mn[a] = FLT_MAX; mx[a] = -FLT_MAX; }
    }
    void grow(const float* lo, const float* hi) {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], lo[a]); mx[a] = std::max(mx[a], hi[a]); }
    }
    void grow(const Box& o) { grow(o.mn, o.mx); }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0.f || dy < 0.f || dz < 0.f) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Child {
    int32_t ref;
    Box box;
    uint32_t depth;  // node levels below (leaf = 0)
};

static uint32_t ceil_log2(uint32_t n) {
    uint32_t l = 0;
    while ((1ull << l) < n) ++l;
    return l;
}

// Node levels a balanced fallback tree needs for n primitives: triangle-only ranges may end in leaves of
// kLeafTris triangles, everything else in single-primitive leaves.
static uint32_t levels_needed(uint32_t n, bool all_tri) {
    return all_tri ? ceil_log2((n + kLeafTris - 1u) / kLeafTris) : ceil_log2(n);
}

struct Builder {
    BuildPrim* prims;
    std::vector<BvhNode> local;          // nodes of this build, indices relative to 0
    std::atomic<uint32_t> next_node{0};
    std::atomic<int> spare_threads{0};

    uint32_t alloc_node() { return next_node.fetch_add(1, std::memory_order_relaxed); }

    static inline float centroid(const BuildPrim& p, int a) { return 0.5f * (p.bmin[a] + p.bmax[a]); }

    Child make_leaf(uint32_t b, uint32_t e, const Box& box) {
        Child c;
        // temporary encoding: offset field = position in prims[]; patched after the build
        c.ref = (int32_t)make_leaf_ref(prims[b].kind, b, e - b);
        c.box = box;
        c.depth = 0;
        return c;
    }

    unsigned total_threads = 1;

    // body(piece, lo, hi) over [b, e) in `pieces` contiguous pieces, one thread each
    template <typename F>
    static void in_pieces(uint32_t b, uint32_t e, unsigned pieces, F body) {
        if (pieces <= 1u) {
            body(0u, b, e);
            return;
        }
        std::vector<std::thread> pool;
        const uint64_t n = e - b;
        for (unsigned k = 0; k < pieces; ++k) {
            const uint32_t lo = b + (uint32_t)(n * k / pieces), hi = b + (uint32_t)(n * (k + 1) / pieces);
            pool.emplace_back([=, &body]() { body(k, lo, hi); });
        }
        for (std::thread& t : pool) t.join();
    }

    Child build_range(uint32_t b, uint32_t e, uint32_t budget, int task_depth) {
        const uint32_t n = e - b;
        // The passes over a large range (bounds, then the bins of all three axes) run on several threads: at the top
        // of a ten-million-primitive build they are most of the time, and the ranges below are not yet many enough to
        // keep the threads busy.  Minima, maxima and counts do not depend on how the range is cut, so the tree is the
        // same for any number of threads.
        constexpr unsigned MAX_PIECES = 32;
        unsigned pieces = 1;
        if (n >= (1u << 17) && total_threads > 1u) pieces = std::min(std::min(total_threads >> std::min(task_depth, 4), MAX_PIECES), n >> 15);
        if (pieces < 1u) pieces = 1u;
        Box box;
        box.reset();
        Box cbox;
        cbox.reset();
        bool all_tri = true;
        {
            Box pbox[MAX_PIECES], pcbox[MAX_PIECES];
            bool ptri[MAX_PIECES];
            in_pieces(b, e, pieces, [&](unsigned k, uint32_t lo, uint32_t hi) {
                Box bx, cb;
                bx.reset();
                cb.reset();
                bool tri = true;
                for (uint32_t i = lo; i < hi; ++i) {
                    bx.grow(prims[i].bmin, prims[i].bmax);
                    float c[3] = {centroid(prims[i], 0), centroid(prims[i], 1), centroid(prims[i], 2)};
                    cb.grow(c, c);
                    tri = tri && prims[i].kind == KIND_TRI;
                }
                pbox[k] = bx; pcbox[k] = cb; ptri[k] = tri;
            });
            for (unsigned k = 0; k < pieces; ++k) {
                box.grow(pbox[k]);
                cbox.grow(pcbox[k]);
                all_tri = all_tri && ptri[k];
            }
        }
        if (n == 1) return make_leaf(b, e, box);

        uint32_t mid = 0;
        bool have_split = false;
        float best_cost = FLT_MAX;
        int best_axis = -1, best_bin = -1;
        const float ext[3] = {cbox.mx[0] - cbox.mn[0], cbox.mx[1] - cbox.mn[1], cbox.mx[2] - cbox.mn[2]};

        // --- binned SAH over the three axes: the bins of all axes are filled in one pass over the range
        struct Bins {
            Box bb[3][kBins];
            uint32_t cnt[3][kBins];
        };
        Bins one;  // (the common case, a range on one thread, allocates nothing)
        std::vector<Bins> many;
        if (pieces > 1u) many.resize(pieces);
        Bins* pbins = pieces > 1u ? many.data() : &one;
        float scale3[3];
        for (int a = 0; a < 3; ++a) scale3[a] = ext[a] > 0.f ? (float)kBins / ext[a] : 0.f;
        in_pieces(b, e, pieces, [&](unsigned k, uint32_t lo, uint32_t hi) {
            Bins& B = pbins[k];
            for (int a = 0; a < 3; ++a)
                for (int j = 0; j < kBins; ++j) { B.bb[a][j].reset(); B.cnt[a][j] = 0; }
            for (uint32_t i = lo; i < hi; ++i) {
                for (int a = 0; a < 3; ++a) {
                    if (!(ext[a] > 0.f)) continue;
                    int j = (int)((centroid(prims[i], a) - cbox.mn[a]) * scale3[a]);
                    j = std::min(std::max(j, 0), kBins - 1);
                    B.bb[a][j].grow(prims[i].bmin, prims[i].bmax);
                    B.cnt[a][j]++;
                }
            }
        });
        for (int a = 0; a < 3; ++a) {
            if (!(ext[a] > 0.f)) continue;
            Box bb[kBins];
            uint32_t cnt[kBins];
            for (int k = 0; k < kBins; ++k) { bb[k].reset(); cnt[k] = 0; }
            for (unsigned q = 0; q < pieces; ++q)
                for (int k = 0; k < kBins; ++k) {
                    if (pbins[q].cnt[a][k]) bb[k].grow(pbins[q].bb[a][k]);
                    cnt[k] += pbins[q].cnt[a][k];
                }
            float right_area[kBins];
            uint32_t right_cnt[kBins];
            Box acc;
            acc.reset();
            uint32_t c = 0;
            for (int k = kBins - 1; k > 0; --k) {
                acc.grow(bb[k]);
                c += cnt[k];
                right_area[k] = acc.half_area();
                right_cnt[k] = c;
            }
            acc.reset();
            c = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                acc.grow(bb[k]);
                c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                float cost = acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
            }
        }

        const float parent_area = box.half_area();
        if (all_tri && n <= kLeafTris) {
            if (budget == 0) return make_leaf(b, e, box);
            float leaf_cost = (float)n;
            float split_cost = (best_axis >= 0 && parent_area > 0.f) ? kTraversalCost + best_cost / parent_area : FLT_MAX;
            if (leaf_cost <= split_cost) return make_leaf(b, e, box);
        }

        if (best_axis >= 0) {
            const int a = best_axis;
            const float scale = (float)kBins / ext[a];
            const float cmin = cbox.mn[a];
            BuildPrim* m = std::partition(prims + b, prims + e, [&](const BuildPrim& p) {
                int k = (int)((centroid(p, a) - cmin) * scale);
                k = std::min(std::max(k, 0), kBins - 1);
                return k <= best_bin;
            });
            mid = (uint32_t)(m - prims);
            have_split = mid > b && mid < e;
            // keep the depth bound: both sides must still fit a balanced tree in the remaining budget
            if (have_split) {
                bool lt = true, rt = true;
                for (uint32_t i = b; i < mid; ++i) lt = lt && prims[i].kind == KIND_TRI;
                for (uint32_t i = mid; i < e; ++i) rt = rt && prims[i].kind == KIND_TRI;
                if (levels_needed(mid - b, lt) + 1 > budget || levels_needed(e - mid, rt) + 1 > budget) have_split = false;
            }
        }
        if (!have_split) {
            // median split on the widest centroid axis (or by position when all centroids coincide)
            int a = 0;
            if (ext[1] > ext[a]) a = 1;
            if (ext[2] > ext[a]) a = 2;
            mid = b + n / 2;
            if (ext[a] > 0.f) {
                std::nth_element(prims + b, prims + mid, prims + e,
                                 [a](const BuildPrim& x, const BuildPrim& y) { return centroid(x, a) < centroid(y, a); });
            }
        }

        const uint32_t node = alloc_node();
        Child l, r;
        const uint32_t child_budget = budget > 0 ? budget - 1 : 0;
        bool spawned = false;
        if (n > (1u << 14) && task_depth < 8) {
            int avail = spare_threads.fetch_sub(1, std::memory_order_relaxed);
            if (avail > 0) {
                spawned = true;
                auto fut = std::async(std::launch::async,
                                      [&, this]() { return build_range(b, mid, child_budget, task_depth + 1); });
                r = build_range(mid, e, child_budget, task_depth + 1);
                l = fut.get();
                spare_threads.fetch_add(1, std::memory_order_relaxed);
            } else {
                spare_threads.fetch_add(1, std::memory_order_relaxed);
            }
        }
        if (!spawned) {
            l = build_range(b, mid, child_budget, task_depth + 1);
            r = build_range(mid, e, child_budget, task_depth + 1);
        }
        BvhNode& nd = local[node];
        for (int a = 0; a < 3; ++a) {
            nd.lmin[a] = l.box.mn[a]; nd.lmax[a] = l.box.mx[a];
            nd.rmin[a] = r.box.mn[a]; nd.rmax[a] = r.box.mx[a];
        }
        nd.left = l.ref;
        nd.right = r.ref;
        Child out;
        out.ref = (int32_t)node;
        out.box = box;
        out.depth = 1 + std::max(l.depth, r.depth);
        return out;
    }
};

}  // namespace

BuildResult build_bvh(std::vector<BuildPrim>& prims, std::vector<BvhNode>& nodes, std::vector<uint32_t>& tri_order,
                      uint32_t tri_base, uint32_t max_depth, unsigned threads, int64_t index_base) {
    BuildResult res;
    const uint32_t n = (uint32_t)prims.size();
    const uint32_t node_base = index_base >= 0 ? (uint32_t)index_base : (uint32_t)nodes.size();
    Builder bld;
    bld.prims = prims.data();
    bld.local.resize(std::max<uint32_t>(n, 1u) + 1u);
    std::memset(bld.local.data(), 0, bld.local.size() * sizeof(BvhNode));
    bld.spare_threads.store(threads > 1 ? (int)threads - 1 : 0);
    bld.total_threads = threads > 0 ? threads : 1u;

    const int32_t ref_empty = (int32_t)(REF_LEAF | (5u << 28));  // kind 5: empty child, skipped by traversal
    auto set_empty_box = [](float* mn, float* mx) {
        for (int a = 0; a < 3; ++a) { mn[a] = 3.0e38f; mx[a] = 3.0e38f; }
    };

    Child top;
    if (n == 0) {
        uint32_t node = bld.alloc_node();
        BvhNode& nd = bld.local[node];
        set_empty_box(nd.lmin, nd.lmax);
        set_empty_box(nd.rmin, nd.rmax);
        nd.left = nd.right = ref_empty;
        top.ref = (int32_t)node;
        top.box.reset();
        top.depth = 1;
    } else {
        bool all_tri = true;
        for (uint32_t i = 0; i < n; ++i) all_tri = all_tri && prims[i].kind == KIND_TRI;
        // the caller's bound is honoured whenever a balanced tree fits it; otherwise the tree gets the
        // levels it needs and the caller sees the excess in BuildResult::depth
        const uint32_t budget = std::max<uint32_t>(max_depth, levels_needed(n, all_tri));
        top = bld.build_range(0, n, budget, 0);
        if (top.ref < 0) {  // single primitive: wrap the leaf so that the root is an inner node
            uint32_t node = bld.alloc_node();
            BvhNode& nd = bld.local[node];
            for (int a = 0; a < 3; ++a) { nd.lmin[a] = top.box.mn[a]; nd.lmax[a] = top.box.mx[a]; }
            set_empty_box(nd.rmin, nd.rmax);
            nd.left = top.ref;
            nd.right = ref_empty;
            top.ref = (int32_t)node;
            top.depth = 1;
        }
    }

    // triangle records are emitted in prims[] order (leaf ranges are contiguous there)
    std::vector<uint32_t> tri_pos(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        tri_pos[i + 1] = tri_pos[i] + (prims[i].kind == KIND_TRI ? 1u : 0u);
        if (prims[i].kind == KIND_TRI) tri_order.push_back(prims[i].index);
    }
    const uint32_t used = bld.next_node.load();
    // Nodes are emitted root first, the two children of a node next to each other, then the left subtree before the
    // right one: the far child of a visit (popped later) shares a cache line with the near child more often than
    // not, and a subtree is a contiguous run of nodes.
    std::vector<uint32_t> order, remap(used, 0u), todo;
    order.reserve(used);
    order.push_back((uint32_t)top.ref);
    todo.push_back((uint32_t)top.ref);
    while (!todo.empty()) {
        const uint32_t n = todo.back();
        todo.pop_back();
        const int32_t l = bld.local[n].left, r = bld.local[n].right;
        if (l >= 0) { remap[(uint32_t)l] = (uint32_t)order.size(); order.push_back((uint32_t)l); }
        if (r >= 0) { remap[(uint32_t)r] = (uint32_t)order.size(); order.push_back((uint32_t)r); }
        if (r >= 0) todo.push_back((uint32_t)r);
        if (l >= 0) todo.push_back((uint32_t)l);
    }
    auto patch = [&](int32_t ref) -> int32_t {
        if (ref >= 0) return (int32_t)(remap[(uint32_t)ref] + node_base);
        uint32_t u = (uint32_t)ref;
        uint32_t kind = (u >> 28) & 7u;
        if (kind == 5u) return ref;
        uint32_t pos = u & REF_OFFSET_MASK;
        uint32_t cnt = ((u >> 25) & 7u) + 1u;
        if (kind == KIND_TRI) return (int32_t)make_leaf_ref(KIND_TRI, tri_base + tri_pos[pos], cnt);
        return (int32_t)make_leaf_ref(kind, prims[pos].index, 1u);
    };
    nodes.reserve(nodes.size() + order.size());
    for (uint32_t k = 0; k < (uint32_t)order.size(); ++k) {
        BvhNode nd = bld.local[order[k]];
        nd.left = patch(nd.left);
        nd.right = patch(nd.right);
        nodes.push_back(nd);
    }
    res.root = (int32_t)node_base;  // the root is emitted first
    res.depth = top.depth;
    for (int a = 0; a < 3; ++a) { res.root_min[a] = top.box.mn[a]; res.root_max[a] = top.box.mx[a]; }
    return res;
}

}  // namespace rrt
