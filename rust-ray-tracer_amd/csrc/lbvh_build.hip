// lbvh_build.hip -- linear BVH on the GPU: 48-bit Morton codes -> radix sort -> radix-tree topology (every inner node
// from the sorted codes alone, Karras 2012) -> boxes bottom-up, one kernel launch per tree level -> (optionally) the
// levels above a cut rebuilt with the surface-area heuristic over the subtrees below it.
//
// A radix tree splits where Morton codes change, which is a poor place near the root (children that overlap or span
// empty space) and a fair one near the leaves (a few neighbouring primitives).  The hybrid keeps the radix tree's
// small subtrees ("clusters", at most cluster_size primitives: contiguous runs of the sorted order, so their triangle
// leaves stay contiguous runs of records) and replaces everything above them: the clusters' boxes go to the host's
// binned SAH builder (n / cluster_size-ish items instead of n), and its nodes are written into the slots of the radix
// nodes they replace -- a binary tree over c clusters has c - 1 inner nodes either way.
//
// The bottom-up pass hands boxes from children to parents only ACROSS kernel launches: a node is finished in
// launch k if both children were finished in launches < k.  Inside one launch nothing written by another
// workgroup is consumed, so no fences or cache write-backs are needed (the L2s of the eight XCDs are not coherent
// with each other within a launch).  A tree of h levels takes h launches of a few microseconds each.
#include "lbvh_build.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace rrt {
namespace {

// A radix subtree of at most this many primitives, all triangles, becomes one leaf (experiment knob RRT_LBVH_LEAF).
// Measured on MI355X with the upper levels rebuilt: 2 -- the two triangles of a quad, what the host's SAH leaves
// mostly are -- renders faster than 4, whose leaves cost nearly twice the primitive tests per ray.
constexpr uint32_t LBVH_LEAF_TRIS_DEFAULT = 2;
constexpr int BLOCK = 256;

struct LeafInfo {  // one sorted primitive
    float mn[3];
    uint32_t kind;
    float mx[3];
    uint32_t index;
};
static_assert(sizeof(LeafInfo) == 32, "LeafInfo must be 32 bytes");

struct Topo {  // one inner node of the radix tree over sorted positions [first, last]
    int left, right;  // >= 0: inner node; < 0: ~(sorted position of a leaf)
    int first, last;
};

struct NodeState {
    float mn[3];
    uint32_t depth;  // node levels of the subtree as the traversal sees it (collapsed subtrees count 0)
    float mx[3];
    uint32_t done;   // number of the launch that finished the node, 0: not finished
};
static_assert(sizeof(NodeState) == 32, "NodeState must be 32 bytes");

// Spreads the low 16 bits of v to every third bit of a 64-bit word.
__device__ __forceinline__ unsigned long long spread3(uint32_t v) {
    unsigned long long x = v & 0xFFFFu;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// 48-bit Morton code of the box centre: the three 16-bit plane indices of the BVH's grid, interleaved.
__global__ void k_morton(const BuildPrim* __restrict__ prims, uint32_t n, Grid g, unsigned long long* __restrict__ keys,
                         uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const BuildPrim p = prims[i];
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float c = 0.5f * (p.bmin[a] + p.bmax[a]);
        const float f = fminf(fmaxf((c - g.origin[a]) / g.step[a], 0.0f), 65535.0f);
        q[a] = (uint32_t)f;
    }
    keys[i] = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
    vals[i] = i;
}

// The caller's triangles -> records and boxes, with the host expander's arithmetic (scene_build.cc): record = v0 and the
// edges v1 - v0, v2 - v0 (reference src/geometry/triangle.rs:28-29), box = the vertices' extent -+ 1e-4
// (triangle.rs:82-94) -+ the builder's pad.  Triangle i of the run gets ordinal ordinal0 + i.
__global__ void k_expand_tris(const rrt_triangle* __restrict__ src, uint32_t count, uint32_t id0, uint32_t ordinal0, float pad,
                              BuildPrim* __restrict__ prims, TriRec* __restrict__ tris) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const rrt_triangle t = src[i];
    TriRec rec;
    BuildPrim p;
    for (int a = 0; a < 3; ++a) {
        rec.v0[a] = t.v0[a];
        rec.e1[a] = t.v1[a] - t.v0[a];
        rec.e2[a] = t.v2[a] - t.v0[a];
        p.bmin[a] = (fminf(t.v0[a], fminf(t.v1[a], t.v2[a])) - 0.0001f) - pad;
        p.bmax[a] = (fmaxf(t.v0[a], fmaxf(t.v1[a], t.v2[a])) + 0.0001f) + pad;
    }
    rec.id = id0 + i;
    rec.material = t.material;
    rec.pad = 0u;
    p.kind = KIND_TRI;
    p.index = ordinal0 + i;
    prims[i] = p;
    tris[ordinal0 + i] = rec;
}

__global__ void k_gather(const BuildPrim* __restrict__ prims, const uint32_t* __restrict__ vals, uint32_t n,
                         LeafInfo* __restrict__ leaf, uint32_t* __restrict__ is_tri) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j > n) return;
    if (j == n) {
        is_tri[n] = 0u;  // the scan runs over n + 1 entries so that prefix[n] is the triangle total
        return;
    }
    const BuildPrim p = prims[vals[j]];
    LeafInfo l;
    for (int a = 0; a < 3; ++a) { l.mn[a] = p.bmin[a]; l.mx[a] = p.bmax[a]; }
    l.kind = p.kind;
    l.index = p.index;
    leaf[j] = l;
    is_tri[j] = p.kind == KIND_TRI ? 1u : 0u;
}

// Length of the common prefix of the keys at sorted positions i and j; the position breaks ties between equal
// Morton codes (as 32 further bits), so all keys are distinct.  -1 outside the array.
__device__ __forceinline__ int common_prefix(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((unsigned)i ^ (unsigned)j));
}

__global__ void k_topology(const unsigned long long* __restrict__ keys, int n, Topo* __restrict__ topo, int* __restrict__ parent) {
    const int i = (int)(blockIdx.x * BLOCK + threadIdx.x);
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int delta_min = common_prefix(keys, n, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > delta_min) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (common_prefix(keys, n, i, i + (l + t) * d) > delta_min) l += t;
    const int j = i + l * d;
    const int delta_node = common_prefix(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (common_prefix(keys, n, i, i + (s + t) * d) > delta_node) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    Topo o;
    o.first = i < j ? i : j;
    o.last = i < j ? j : i;
    o.left = o.first == gamma ? ~gamma : gamma;
    o.right = o.last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    topo[i] = o;
    if (o.left >= 0) parent[o.left] = i;
    if (o.right >= 0) parent[o.right] = i;
    if (i == 0) parent[0] = -1;
}

__device__ __forceinline__ uint32_t count_of(const Topo& t) { return (uint32_t)(t.last - t.first + 1); }

// A radix subtree that the traversal sees as one triangle leaf.
__device__ __forceinline__ bool collapsed_leaf(const Topo& t, const uint32_t* __restrict__ tri_prefix, uint32_t leaf_tris) {
    // (both loads always: with the second one behind the first comparison, hipcc 7.2 reuses the register of a value
    // loaded earlier -- the child's depth in view_child -- and sets it to zero on both sides of the branch)
    const uint32_t cnt = count_of(t), tris = tri_prefix[t.last + 1] - tri_prefix[t.first];
    return (cnt <= leaf_tris) & (tris == cnt);
}

struct ChildView {
    bool ready;
    float mn[3], mx[3];
    int32_t ref;
    uint32_t depth;
};

__device__ __forceinline__ ChildView view_child(int c, const Topo* __restrict__ topo, const LeafInfo* __restrict__ leaf,
                                                const uint32_t* __restrict__ tri_prefix, const NodeState* __restrict__ st,
                                                uint32_t launch, uint32_t node_base, uint32_t tri_base, uint32_t leaf_tris) {
    ChildView v;
    if (c < 0) {
        const int pos = ~c;
        const LeafInfo l = leaf[pos];
        for (int a = 0; a < 3; ++a) { v.mn[a] = l.mn[a]; v.mx[a] = l.mx[a]; }
        v.ref = l.kind == KIND_TRI ? (int32_t)make_leaf_ref(KIND_TRI, tri_base + tri_prefix[pos], 1u)
                                   : (int32_t)make_leaf_ref(l.kind, l.index, 1u);
        v.depth = 0u;
        v.ready = true;
        return v;
    }
    const NodeState s = st[c];
    v.ready = s.done != 0u && s.done < launch;  // finished by an EARLIER launch
    for (int a = 0; a < 3; ++a) { v.mn[a] = s.mn[a]; v.mx[a] = s.mx[a]; }
    const Topo t = topo[c];
    const uint32_t first_tri = tri_prefix[t.first];
    const bool collapsed = collapsed_leaf(t, tri_prefix, leaf_tris);
    v.ref = collapsed ? (int32_t)make_leaf_ref(KIND_TRI, tri_base + first_tri, count_of(t)) : (int32_t)(node_base + (uint32_t)c);
    v.depth = collapsed ? 0u : s.depth;
    return v;
}

// The radix nodes whose slots belong to k_cluster_sah (sah_size > 0): everything inside a subtree of at most sah_size
// primitives that hangs below a larger node, and that subtree's root unless it is a collapsed leaf.
__device__ __forceinline__ bool owned_by_cluster(int i, const Topo& t, const Topo* __restrict__ topo, const int* __restrict__ parent,
                                                 const uint32_t* __restrict__ tri_prefix, uint32_t sah_size, uint32_t leaf_tris) {
    if (sah_size == 0u || count_of(t) > sah_size) return false;
    const int p = parent[i];
    if (p >= 0 && count_of(topo[p]) <= sah_size) return true;
    return !collapsed_leaf(t, tri_prefix, leaf_tris);
}

__global__ void k_refit(const Topo* __restrict__ topo, const int* __restrict__ parent, const LeafInfo* __restrict__ leaf,
                        const uint32_t* __restrict__ tri_prefix, NodeState* __restrict__ st, QNode* __restrict__ nodes, uint32_t n,
                        uint32_t launch, Grid g, uint32_t node_base, uint32_t tri_base, uint32_t leaf_tris, uint32_t sah_size,
                        uint32_t* __restrict__ weight) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n - 1u) return;
    if (st[i].done != 0u) return;
    const Topo t = topo[i];
    if (owned_by_cluster((int)i, t, topo, parent, tri_prefix, sah_size, leaf_tris)) return;
    const ChildView l = view_child(t.left, topo, leaf, tri_prefix, st, launch, node_base, tri_base, leaf_tris);
    if (!l.ready) return;
    const ChildView r = view_child(t.right, topo, leaf, tri_prefix, st, launch, node_base, tri_base, leaf_tris);
    if (!r.ready) return;
    QNode q;
    q.left = device_ref(l.ref);
    q.right = device_ref(r.ref);
    quantize_box(g, l.mn, l.mx, q.lbox);
    quantize_box(g, r.mn, r.mx, q.rbox);
    nodes[node_base + i] = q;
    NodeState s;
    for (int a = 0; a < 3; ++a) { s.mn[a] = fminf(l.mn[a], r.mn[a]); s.mx[a] = fmaxf(l.mx[a], r.mx[a]); }
    s.depth = 1u + (l.depth > r.depth ? l.depth : r.depth);
    s.done = launch;
    st[i] = s;
    if (weight) {
        // the number of finished subtrees (and stray primitives) below a node above the cut: what a second round of
        // k_subtree_sah groups by
        const uint32_t wl = t.left >= 0 && count_of(topo[t.left]) > sah_size ? weight[t.left] : 1u;
        const uint32_t wr = t.right >= 0 && count_of(topo[t.right]) > sah_size ? weight[t.right] : 1u;
        weight[i] = wl + wr;
    }
}

// One subtree below the cut, as the host's builder sees it.
struct Cluster {
    float mn[3];
    int32_t ref;     // what a parent stores for it: a leaf reference or the subtree's node index (not yet a device reference)
    float mx[3];
    uint32_t depth;  // node levels of the subtree
    uint32_t first;  // first sorted position it covers (gives the clusters a fixed order)
    uint32_t pad;
};

// The cut: radix nodes over more than `cluster_size` primitives are "upper" nodes (to be replaced), and every child of
// an upper node that is not itself upper is a cluster.  counters[0]: upper nodes, counters[1]: clusters,
// counters[2]: set if a cluster was not finished by the refit launches.
__global__ void k_cut(const Topo* __restrict__ topo, const LeafInfo* __restrict__ leaf, const uint32_t* __restrict__ tri_prefix,
                      const NodeState* __restrict__ st, uint32_t n, uint32_t cluster_size, uint32_t node_base, uint32_t tri_base,
                      uint32_t leaf_tris, uint32_t* __restrict__ upper, Cluster* __restrict__ clusters, uint32_t capacity,
                      uint32_t* __restrict__ counters, const uint32_t* __restrict__ weight, uint32_t group_size) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n - 1u) return;
    const Topo t = topo[i];
    // weight != nullptr: the cut lies above the groups the second round of k_subtree_sah built
    auto is_upper = [&](uint32_t node, const Topo& tn) { return count_of(tn) > cluster_size && (weight == nullptr || weight[node] > group_size); };
    if (!is_upper(i, t)) return;
    const uint32_t u = atomicAdd(&counters[0], 1u);
    if (u < capacity) upper[u] = i;
    const int child[2] = {t.left, t.right};
    for (int k = 0; k < 2; ++k) {
        const int c = child[k];
        uint32_t first;
        if (c >= 0) {
            const Topo tc = topo[c];
            if (is_upper((uint32_t)c, tc)) continue;  // upper as well
            first = (uint32_t)tc.first;
        } else {
            first = (uint32_t)~c;
        }
        const ChildView v = view_child(c, topo, leaf, tri_prefix, st, 0xFFFFFFFFu, node_base, tri_base, leaf_tris);
        if (!v.ready) counters[2] = 1u;
        Cluster o;
        for (int a = 0; a < 3; ++a) { o.mn[a] = v.mn[a]; o.mx[a] = v.mx[a]; }
        o.ref = v.ref;
        o.depth = v.depth;
        o.first = first;
        o.pad = 0u;
        const uint32_t at = atomicAdd(&counters[1], 1u);
        if (at < capacity) clusters[at] = o;  // the host sees the overflow in the counter and keeps the radix tree
    }
}

__global__ void k_scatter_nodes(const QNode* __restrict__ src, const uint32_t* __restrict__ slot, uint32_t m, QNode* __restrict__ nodes) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < m) nodes[slot[k]] = src[k];
}

// Triangle records into leaf order: sorted position j holds the triangle with ordinal leaf[j].index.
__global__ void k_permute_tris(const LeafInfo* __restrict__ leaf, const uint32_t* __restrict__ tri_prefix, uint32_t n,
                               const TriRec* __restrict__ src, uint32_t tri_count, TriRec* __restrict__ dst) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n) return;
    const LeafInfo l = leaf[j];
    if (l.kind != KIND_TRI || l.index >= tri_count) return;
    const uint32_t to = tri_prefix[j];
    if (to >= tri_count) return;
    const float4* s = reinterpret_cast<const float4*>(src + l.index);
    float4* d = reinterpret_cast<float4*>(dst + to);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

// ---- subtrees below the cut, rebuilt on the GPU with the surface-area heuristic -----------------------------------
//
// One wavefront per subtree of at most 64 primitives, one primitive per lane.  The tree is built top-down, all segments
// of one level at once: a segment is a run of lanes [lo, hi); every level sorts the lanes by (segment, centroid) along
// each axis with one bitonic network over the wavefront, forms the boxes left and right of every split position with
// segmented scans, and takes the cheapest split of the three axes (full sweep, the host builder's cost: area x count).
// Triangle segments end as leaves by the host builder's rule (count <= node cost + split cost / area).  The subtree's
// nodes go into the slots of the radix nodes they replace: a radix subtree over sorted positions [f, l] owns the
// slots f..l-1 (root at f) or f+1..l (root at l), and the root keeps its slot, so whatever refers to the subtree stays
// valid.  While it is built, a node is named by the gap between neighbouring positions at which it splits its segment.
constexpr int CL_MAX = 64;          // primitives per rebuilt subtree
constexpr int CL_WAVES = 4;         // wavefronts per workgroup
constexpr uint32_t CL_LEVELS = 10;  // node levels a rebuilt subtree may have (a balanced tree over 64 primitives needs 6)

__global__ void k_cluster_roots(const Topo* __restrict__ topo, const int* __restrict__ parent, const uint32_t* __restrict__ tri_prefix,
                                uint32_t n, uint32_t sah_size, uint32_t leaf_tris, uint32_t* __restrict__ roots, uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n - 1u) return;
    const Topo t = topo[i];
    if (count_of(t) > sah_size) return;
    const int p = parent[i];
    if (p < 0 || count_of(topo[p]) <= sah_size) return;
    if (collapsed_leaf(t, tri_prefix, leaf_tris)) return;
    roots[atomicAdd(count, 1u)] = i;  // the order does not matter: subtrees are independent of each other
}

__device__ __forceinline__ uint32_t orderable(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__device__ __forceinline__ float half_area(const float* mn, const float* mx) {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ uint32_t ceil_log2_dev(uint32_t x) { return x <= 1u ? 0u : 32u - (uint32_t)__clz((int)(x - 1u)); }

// Ascending sort of 64 distinct keys, one per lane (bitonic network).
__device__ __forceinline__ unsigned long long wave_sort(unsigned long long key, int lane) {
    for (int k2 = 2; k2 <= 64; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned long long other = __shfl_xor(key, j, 64);
            const bool take_min = ((lane & j) == 0) == ((lane & k2) == 0);
            const bool swap = take_min ? other < key : other > key;
            key = swap ? other : key;
        }
    return key;
}

// Lanes sorted by (segment, c) -- src: the lane whose primitive comes to this position -- and, for every position r of
// a segment [lo, hi), the box of positions lo..r (lmn, lmx) and of positions r+1..hi-1 (rmn, rmx; garbage at r = hi-1).
__device__ __forceinline__ void sweep(int lane, int lo, int hi, float c, const float* mn, const float* mx, int& src, float* lmn,
                                      float* lmx, float* rmn, float* rmx) {
    unsigned long long key = ((unsigned long long)(uint32_t)lo << 38) | ((unsigned long long)orderable(c) << 6) | (unsigned long long)lane;
    key = wave_sort(key, lane);
    src = (int)(key & 63ull);
    float smn[3], smx[3];
    for (int a = 0; a < 3; ++a) {
        lmn[a] = smn[a] = __shfl(mn[a], src, 64);
        lmx[a] = smx[a] = __shfl(mx[a], src, 64);
    }
    for (int d = 1; d < 64; d <<= 1) {
        const bool in = lane - d >= lo;
        for (int a = 0; a < 3; ++a) {
            const float o0 = __shfl_up(lmn[a], d, 64), o1 = __shfl_up(lmx[a], d, 64);
            if (in) { lmn[a] = fminf(lmn[a], o0); lmx[a] = fmaxf(lmx[a], o1); }
        }
    }
    for (int d = 1; d < 64; d <<= 1) {
        const bool in = lane + d < hi;
        for (int a = 0; a < 3; ++a) {
            const float o0 = __shfl_down(smn[a], d, 64), o1 = __shfl_down(smx[a], d, 64);
            if (in) { smn[a] = fminf(smn[a], o0); smx[a] = fmaxf(smx[a], o1); }
        }
    }
    for (int a = 0; a < 3; ++a) {
        rmn[a] = __shfl_down(smn[a], 1, 64);
        rmx[a] = __shfl_down(smx[a], 1, 64);
    }
}

// A reference inside a node that is still being assembled: the gap of the child node (bit 30 marks it: a finished
// reference to a node is a byte offset below 2^30, a leaf reference has bit 31).
constexpr int32_t GAP_REF = 0x40000000;
__device__ __forceinline__ bool is_gap_ref(int32_t r) { return ((uint32_t)r & 0xC0000000u) == (uint32_t)GAP_REF; }

// LEVEL 1: the items are the primitives of a radix subtree of at most 64 (roots: from k_cluster_roots); leaves of
// several triangles; the primitives' new order is written back.  LEVEL 2: the items are the finished subtrees (and
// stray primitives) below a radix node that has at most 64 of them (roots: from k_group_roots); every item is a
// leaf; the nodes go into the slots of the radix nodes between that node and the items.
template <int LEVEL>
__global__ __launch_bounds__(CL_MAX * CL_WAVES) void k_subtree_sah(
    const uint32_t* __restrict__ roots, const uint32_t* __restrict__ root_count, const Topo* __restrict__ topo, LeafInfo* __restrict__ leaf,
    uint32_t* __restrict__ tri_prefix, NodeState* __restrict__ st, QNode* __restrict__ nodes, Grid g, uint32_t node_base, uint32_t tri_base,
    uint32_t leaf_tris, float node_cost, uint32_t cluster_size, uint32_t radix_leaf_tris, uint32_t* __restrict__ failed) {
    __shared__ QNode s_nodes[CL_WAVES][CL_MAX];
    __shared__ uint8_t s_rank[CL_WAVES][CL_MAX];
    __shared__ float s_box[LEVEL == 2 ? CL_WAVES : 1][LEVEL == 2 ? CL_MAX : 1][6];
    __shared__ int32_t s_ref[LEVEL == 2 ? CL_WAVES : 1][LEVEL == 2 ? CL_MAX : 1];
    __shared__ uint32_t s_depth[LEVEL == 2 ? CL_WAVES : 1][LEVEL == 2 ? CL_MAX : 1];
    __shared__ int s_slot[LEVEL == 2 ? CL_WAVES : 1][LEVEL == 2 ? CL_MAX : 1];   // radix nodes the subtree may use, its root first
    __shared__ int s_todo[LEVEL == 2 ? CL_WAVES : 1][LEVEL == 2 ? CL_MAX : 1];
    __shared__ int s_count[LEVEL == 2 ? CL_WAVES : 1][2];
    const int lane = (int)(threadIdx.x & 63u), w = (int)(threadIdx.x >> 6);
    const uint32_t ci = blockIdx.x * CL_WAVES + (uint32_t)w;
    if (ci >= *root_count) return;  // the whole wavefront leaves; the kernel has no workgroup barrier
    const uint32_t c = roots[ci];
    const Topo t = topo[c];
    int k;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t kind = 0xFFFFFFFFu, index = 0u;  // LEVEL 2: the item's reference and its depth
    uint32_t tri0 = 0u;
    if constexpr (LEVEL == 1) {
        k = t.last - t.first + 1;
        if (k < 2 || k > CL_MAX) {
            if (lane == 0) *failed = 1u;
            return;
        }
        if (lane < k) {
            const LeafInfo l = leaf[t.first + lane];
            for (int a = 0; a < 3; ++a) { mn[a] = l.mn[a]; mx[a] = l.mx[a]; }
            kind = l.kind;
            index = l.index;
        }
        tri0 = tri_prefix[t.first];
    } else {
        // one lane walks the radix nodes below the root down to the items (at most 63 nodes, 64 items)
        if (lane == 0) {
            int items = 0, inner = 0, sp = 0;
            bool bad = false;
            s_todo[w][sp++] = (int)c;
            while (sp > 0 && !bad) {
                const int node = s_todo[w][--sp];
                if (inner >= CL_MAX - 1) { bad = true; break; }
                s_slot[w][inner++] = node;
                const Topo tn = topo[node];
                const int child[2] = {tn.left, tn.right};
                for (int e = 0; e < 2; ++e) {
                    const int cc = child[e];
                    if (cc >= 0 && count_of(topo[cc]) > cluster_size) {
                        if (sp >= CL_MAX) { bad = true; break; }
                        s_todo[w][sp++] = cc;
                        continue;
                    }
                    if (items >= CL_MAX) { bad = true; break; }
                    const ChildView v = view_child(cc, topo, leaf, tri_prefix, st, 0xFFFFFFFFu, node_base, tri_base, radix_leaf_tris);
                    if (!v.ready) { bad = true; break; }
                    for (int a = 0; a < 3; ++a) { s_box[w][items][a] = v.mn[a]; s_box[w][items][3 + a] = v.mx[a]; }
                    s_ref[w][items] = device_ref(v.ref);
                    s_depth[w][items] = v.depth;
                    ++items;
                }
            }
            s_count[w][0] = bad ? 0 : items;
            s_count[w][1] = inner;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        k = s_count[w][0];
        if (k < 2 || k > CL_MAX || s_count[w][1] != k - 1) {
            if (lane == 0) *failed = 1u;
            return;
        }
        if (lane < k) {
            for (int a = 0; a < 3; ++a) { mn[a] = s_box[w][lane][a]; mx[a] = s_box[w][lane][3 + a]; }
            kind = (uint32_t)s_ref[w][lane];
            index = s_depth[w][lane];
        }
    }
    const bool valid = lane < k;
    int lo = valid ? 0 : lane, hi = valid ? k : lane + 1;
    bool done = !valid;
    int parent_gap = -1, side = 0;
    uint32_t gap_level = 0xFFFFu;  // lane g: the level at which the node that splits between positions g and g + 1 was made
    uint32_t depth_here = 0u;
    unsigned long long used = 0ull;
    float cmn[3] = {0.f, 0.f, 0.f}, cmx[3] = {0.f, 0.f, 0.f};  // the subtree's box

    for (uint32_t level = 0; level <= CL_LEVELS; ++level) {
        const unsigned long long tri_mask = LEVEL == 1 ? __ballot(kind == KIND_TRI) : 0ull;
        const unsigned long long seg_mask = (hi >= 64 ? ~0ull : (1ull << hi) - 1ull) & ~((1ull << lo) - 1ull);
        const int cnt = hi - lo;
        const bool all_tri = LEVEL == 1 && (tri_mask & seg_mask) == seg_mask;
        const bool open = !done;
        bool eval = open && cnt > 1;
        if (__ballot(open) == 0ull) break;
        const uint32_t rem = CL_LEVELS - level;  // node levels this segment's subtree may still use
        bool make_leaf = open && cnt == 1;
        if (eval && rem == 0u) {
            // cannot happen (a split is only taken if both sides still fit a balanced tree); kept so that a mistake
            // ends in an error and not in a tree the traversal cannot walk
            if (lane == lo) *failed = 1u;
            make_leaf = true;
            eval = false;
        }

        // --- cheapest split of every open segment: full sweep along the three axes
        unsigned long long best = ~0ull;
        float ext[3] = {0.f, 0.f, 0.f}, parent_area = 0.f;
        for (int a = 0; a < 3; ++a) {
            const float cen = eval ? 0.5f * (mn[a] + mx[a]) : 0.f;
            int src;
            float lmn[3], lmx[3], rmn[3], rmx[3];
            sweep(lane, lo, hi, cen, mn, mx, src, lmn, lmx, rmn, rmx);
            const float sorted_c = __shfl(cen, src, 64);
            ext[a] = __shfl(sorted_c, hi - 1, 64) - __shfl(sorted_c, lo, 64);
            if (a == 0) {
                parent_area = __shfl(half_area(lmn, lmx), hi - 1, 64);
                if (level == 0u)
                    for (int b = 0; b < 3; ++b) { cmn[b] = __shfl(lmn[b], k - 1, 64); cmx[b] = __shfl(lmx[b], k - 1, 64); }
            }
            float cost = half_area(lmn, lmx) * (float)(lane - lo + 1) + half_area(rmn, rmx) * (float)(hi - 1 - lane);
            if (!eval || lane >= hi - 1 || !(cost >= 0.f) || !(cost < 3.0e38f)) cost = 3.4e38f;
            const unsigned long long ck = ((unsigned long long)__float_as_uint(cost) << 8) | ((unsigned long long)a << 6) | (unsigned long long)lane;
            best = ck < best ? ck : best;
        }
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(best, d, 64);
            if (lane - d >= lo && o < best) best = o;
        }
        best = __shfl(best, hi - 1, 64);
        const float best_cost = __uint_as_float((uint32_t)(best >> 8));
        int axis = (int)((best >> 6) & 3ull), split = (int)(best & 63ull);
        bool have = eval && best_cost < 3.0e38f && axis < 3 && split >= lo && split < hi - 1;
        if (eval && level > 0u && all_tri && (uint32_t)cnt <= leaf_tris) {
            const float split_cost = (have && parent_area > 0.f) ? node_cost + best_cost / parent_area : 3.4e38f;
            if ((float)cnt <= split_cost) { make_leaf = true; eval = false; }
        }
        if (eval) {
            if (have && (ceil_log2_dev((uint32_t)(split - lo + 1)) + 1u > rem || ceil_log2_dev((uint32_t)(hi - 1 - split)) + 1u > rem)) have = false;
            if (!have) {  // median split along the widest axis (by position when all centroids coincide)
                axis = 0;
                if (ext[1] > ext[axis]) axis = 1;
                if (ext[2] > ext[axis]) axis = 2;
                split = lo + cnt / 2 - 1;
            }
        }

        // --- the lanes of a splitting segment take the order of its axis; everything else stays where it is
        {
            const float cen = eval ? 0.5f * (mn[axis] + mx[axis]) : 0.f;
            int src;
            float lmn[3], lmx[3], rmn[3], rmx[3];
            sweep(lane, lo, hi, cen, mn, mx, src, lmn, lmx, rmn, rmx);
            for (int a = 0; a < 3; ++a) { mn[a] = __shfl(mn[a], src, 64); mx[a] = __shfl(mx[a], src, 64); }
            kind = (uint32_t)__shfl((int)kind, src, 64);
            index = (uint32_t)__shfl((int)index, src, 64);
            if (eval && lane == split) {
                QNode q;
                q.left = q.right = REF_DONE;
                quantize_box(g, lmn, lmx, q.lbox);
                quantize_box(g, rmn, rmx, q.rbox);
                s_nodes[w][split] = q;
            }
        }
        used |= __ballot(eval && lane == split);
        if (eval && lane == split) gap_level = level;

        // --- what this segment is goes into its parent's node
        if ((eval || make_leaf) && lane == lo && parent_gap >= 0) {
            int32_t ref = GAP_REF | split;  // the gap of the node; turned into a reference when the nodes are written out
            if (make_leaf) {
                if constexpr (LEVEL == 2) ref = (int32_t)kind;
                else  // (the primitives below lo belong to other segments: their number of triangles is final)
                    ref = kind == KIND_TRI ? (int32_t)make_leaf_ref(KIND_TRI, tri_base + tri0 + (uint32_t)__popcll(tri_mask & ((1ull << lo) - 1ull)), (uint32_t)cnt)
                                           : (int32_t)make_leaf_ref(kind, index, 1u);
            }
            if (side == 0) s_nodes[w][parent_gap].left = ref;
            else s_nodes[w][parent_gap].right = ref;
        }
        if (make_leaf) {
            done = true;
            depth_here = LEVEL == 2 ? level + index : level;
        }
        if (eval) {
            parent_gap = split;
            if (lane <= split) { hi = split + 1; side = 0; }
            else { lo = split + 1; side = 1; }
        }
    }
    if (__ballot(!done) != 0ull) {
        if (lane == 0) *failed = 1u;
        return;
    }
    uint32_t depth = valid ? depth_here : 0u;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)depth, d, 64);
        depth = o > depth ? o : depth;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // --- results: the primitives in their new order, the nodes in their slots, the subtree's box for the levels above
    if constexpr (LEVEL == 1) {
        if (valid) {
            LeafInfo l;
            for (int a = 0; a < 3; ++a) { l.mn[a] = mn[a]; l.mx[a] = mx[a]; }
            l.kind = kind;
            l.index = index;
            leaf[t.first + lane] = l;
        }
        const unsigned long long tri_final = __ballot(kind == KIND_TRI);
        if (valid) tri_prefix[t.first + lane] = tri0 + (uint32_t)__popcll(tri_final & ((1ull << lane) - 1ull));
    }
    // The nodes take the subtree's slots level by level, left to right: the two children of a node are then neighbours
    // (nothing else of their level lies between them), and -- LEVEL 1 -- the slots that stay unused (leaves of several
    // triangles need fewer nodes than the radix tree had) are together at the end instead of between the nodes.  The
    // root is first in this order and must keep its own slot.  LEVEL 1: that is the first or the last of the run of
    // slots, and the order runs from there.
    const unsigned long long order = wave_sort(((unsigned long long)gap_level << 6) | (unsigned long long)lane, lane);
    s_rank[w][(int)(order & 63ull)] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool root_first = c == (uint32_t)t.first;
    const uint32_t run0 = node_base + (root_first ? (uint32_t)t.first : (uint32_t)t.first + 1u);  // LEVEL 1: first of the k - 1 slots
    auto slot_of = [&](int gap) -> uint32_t {
        const uint32_t r = s_rank[w][gap];
        if constexpr (LEVEL == 2) return node_base + (uint32_t)s_slot[w][r];
        else return run0 + (root_first ? r : (uint32_t)(k - 2) - r);
    };
    if (lane < k - 1 && ((used >> lane) & 1ull)) {
        QNode q = s_nodes[w][lane];
        if (is_gap_ref(q.left)) q.left = device_ref((int32_t)slot_of(q.left & 63));
        if (is_gap_ref(q.right)) q.right = device_ref((int32_t)slot_of(q.right & 63));
        nodes[slot_of(lane)] = q;
    }
    if (lane == 0) {
        if constexpr (LEVEL == 2) {
            st[c].depth = depth;  // (the box the refit gave the node stands: it is the union of the same items)
        } else {
            NodeState s;
            for (int a = 0; a < 3; ++a) { s.mn[a] = cmn[a]; s.mx[a] = cmx[a]; }
            s.depth = depth;
            s.done = 1u;
            st[c] = s;
        }
    }
}

// Radix nodes above the cut with at most `group_size` finished subtrees below them, whose parent has more.
__global__ void k_group_roots(const Topo* __restrict__ topo, const int* __restrict__ parent, const uint32_t* __restrict__ weight, uint32_t n,
                              uint32_t cluster_size, uint32_t group_size, uint32_t* __restrict__ roots, uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n - 1u) return;
    if (count_of(topo[i]) <= cluster_size || weight[i] > group_size) return;
    const int p = parent[i];
    if (p >= 0 && weight[p] <= group_size) return;
    roots[atomicAdd(count, 1u)] = i;
}

struct Buffers {
    std::vector<void*> p;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Buffers() {
        for (void* q : p)
            if (q) (void)hipFree(q);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    template <typename T>
    hipError_t alloc(T** out, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) return e;
        p.push_back(q);
        *out = static_cast<T*>(q);
        return hipSuccess;
    }
};

#define LBVH_TRY(x)                         \
    do {                                    \
        hipError_t e_ = (x);                \
        if (e_ != hipSuccess) return e_;    \
    } while (0)

}  // namespace

// The levels above the cut, rebuilt: clusters and upper nodes are fetched, the host's SAH builder runs over the
// clusters' boxes, and its nodes go into the upper nodes' slots (the root keeps slot 0).  Returns the depth of the
// finished tree in *depth (0: not done -- too many clusters for the buffers, an unfinished cluster, a tree deeper than
// max_depth; the plain radix tree is then what d_nodes holds).
static hipError_t rebuild_top(const uint32_t* d_upper, const Cluster* d_clusters, const uint32_t* d_counters, uint32_t capacity,
                              const Grid& grid, QNode* d_nodes, uint32_t node_base, uint32_t max_depth, unsigned threads,
                              Buffers& B, uint32_t* depth, uint32_t* n_clusters) {
    *depth = 0;
    *n_clusters = 0;
    const bool trace = getenv("RRT_BUILD_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rrt build]   upper levels: %-28s %7.1f ms\n", what, std::chrono::duration<float, std::milli>(now - t_last).count());
        t_last = now;
    };
    uint32_t counters[3] = {0u, 0u, 0u};
    LBVH_TRY(hipMemcpy(counters, d_counters, sizeof counters, hipMemcpyDeviceToHost));
    const uint32_t n_upper = counters[0], n_cl = counters[1];
    if (counters[2] != 0u || n_cl < 2u || n_cl > capacity || n_upper > capacity || n_upper + 1u != n_cl) return hipSuccess;
    std::vector<uint32_t> upper(n_upper);
    std::vector<Cluster> cl(n_cl);
    LBVH_TRY(hipMemcpy(upper.data(), d_upper, (size_t)n_upper * sizeof(uint32_t), hipMemcpyDeviceToHost));
    LBVH_TRY(hipMemcpy(cl.data(), d_clusters, (size_t)n_cl * sizeof(Cluster), hipMemcpyDeviceToHost));
    // the kernel appended in whatever order its threads ran: a fixed order makes the build reproducible
    std::sort(upper.begin(), upper.end());
    std::sort(cl.begin(), cl.end(), [](const Cluster& a, const Cluster& b) { return a.first < b.first; });
    mark("fetched and ordered");
    if (upper[0] != 0u) return hipSuccess;  // the radix root is always an upper node
    uint32_t deepest = 0;
    std::vector<BuildPrim> prims(n_cl);
    for (uint32_t k = 0; k < n_cl; ++k) {
        for (int a = 0; a < 3; ++a) { prims[k].bmin[a] = cl[k].mn[a]; prims[k].bmax[a] = cl[k].mx[a]; }
        prims[k].kind = KIND_SUBTREE;
        prims[k].index = k;
        deepest = std::max(deepest, cl[k].depth);
    }
    if (deepest + 1u >= max_depth) return hipSuccess;
    std::vector<BvhNode> top;
    std::vector<uint32_t> no_tris;
    const BuildResult br = build_bvh(prims, top, no_tris, 0u, max_depth - deepest, threads);
    mark("SAH build");
    if (br.depth + deepest > max_depth || top.size() != (size_t)n_upper) return hipSuccess;
    // builder node k -> slot upper[k] (node 0, the root, -> slot 0); cluster ordinals -> the clusters' references
    auto final_ref = [&](int32_t ref) -> int32_t {
        if (ref >= 0) return device_ref((int32_t)(node_base + upper[(uint32_t)ref]));
        const uint32_t u = (uint32_t)ref, kind = (u >> 28) & 7u;
        if (kind != KIND_SUBTREE) return ref;  // an empty child slot
        return device_ref(cl[u & REF_OFFSET_MASK].ref);
    };
    std::vector<QNode> q(top.size());
    std::vector<uint32_t> slot(top.size());
    for (size_t k = 0; k < top.size(); ++k) {
        q[k].left = final_ref(top[k].left);
        q[k].right = final_ref(top[k].right);
        quantize_box(grid, top[k].lmin, top[k].lmax, q[k].lbox);
        quantize_box(grid, top[k].rmin, top[k].rmax, q[k].rbox);
        slot[k] = node_base + upper[k];
    }
    mark("nodes quantised");
    QNode* d_q;
    uint32_t* d_slot;
    LBVH_TRY(B.alloc(&d_q, q.size()));
    LBVH_TRY(B.alloc(&d_slot, slot.size()));
    LBVH_TRY(hipMemcpy(d_q, q.data(), q.size() * sizeof(QNode), hipMemcpyHostToDevice));
    LBVH_TRY(hipMemcpy(d_slot, slot.data(), slot.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const uint32_t m = (uint32_t)q.size();
    hipLaunchKernelGGL(k_scatter_nodes, dim3((m + BLOCK - 1u) / BLOCK), dim3(BLOCK), 0, (hipStream_t)0, d_q, d_slot, m, d_nodes);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipDeviceSynchronize());
    mark("nodes written");
    *depth = br.depth + deepest;
    *n_clusters = n_cl;
    return hipSuccess;
}

// RRT_LBVH_VERIFY=1: the finished tree is fetched and walked on the host before anything traverses it -- every node
// slot inside the BVH's range and reached once, every triangle record in exactly one leaf, as many other leaves as
// there are other primitives, no walk deeper than the depth reported.
static hipError_t verify_tree(const QNode* d_nodes, uint32_t node_base, uint32_t n, uint32_t tri_base, uint32_t tri_count, uint32_t depth) {
    std::vector<QNode> nodes(n - 1u);
    LBVH_TRY(hipMemcpy(nodes.data(), d_nodes + node_base, nodes.size() * sizeof(QNode), hipMemcpyDeviceToHost));
    std::vector<uint8_t> seen_node(n - 1u, 0), seen_tri(tri_count, 0);
    uint64_t others = 0;
    struct Item { uint32_t node, level; };
    std::vector<Item> todo{{0u, 1u}};
    auto fail = [&](const char* what, uint32_t at) {
        fprintf(stderr, "rrt: GPU-built BVH failed verification: %s (node %u of %u, depth reported %u)\n", what, at, n - 1u, depth);
        if (n <= 64u)
            for (uint32_t k = 0; k + 1u < n; ++k) fprintf(stderr, "  node %u: left %08x right %08x\n", k, (unsigned)nodes[k].left, (unsigned)nodes[k].right);
        return hipErrorUnknown;
    };
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.level > depth) return fail("deeper than reported", it.node);
        if (seen_node[it.node]) return fail("node reached twice", it.node);
        seen_node[it.node] = 1;
        const int32_t child[2] = {nodes[it.node].left, nodes[it.node].right};
        for (int k = 0; k < 2; ++k) {
            if (child[k] >= 0) {
                const uint32_t idx = (uint32_t)child[k] >> 5;
                if (((uint32_t)child[k] & 31u) != 0u || idx < node_base || idx - node_base >= n - 1u) return fail("child outside the BVH's nodes", it.node);
                todo.push_back({idx - node_base, it.level + 1u});
                continue;
            }
            const uint32_t u = (uint32_t)child[k], kind = (u >> 28) & 7u, first = u & REF_OFFSET_MASK, cnt = ((u >> 25) & 7u) + 1u;
            if (kind == KIND_TRI) {
                if (first < tri_base || first - tri_base + cnt > tri_count) return fail("triangle leaf outside the BVH's records", it.node);
                for (uint32_t j = 0; j < cnt; ++j) {
                    if (seen_tri[first - tri_base + j]) return fail("triangle in two leaves", it.node);
                    seen_tri[first - tri_base + j] = 1;
                }
            } else if (kind == KIND_SPHERE || kind == KIND_RECT || kind == KIND_INST) {
                ++others;
            } else {
                return fail("unexpected kind of child", it.node);
            }
        }
    }
    for (uint32_t j = 0; j < tri_count; ++j)
        if (!seen_tri[j]) return fail("triangle in no leaf", j);
    if (others != (uint64_t)n - tri_count) return fail("other primitives miscounted", (uint32_t)others);
    return hipSuccess;
}

hipError_t lbvh_build(const BuildPrim* h_prims, uint32_t n_host, const Grid& grid, QNode* d_nodes, uint32_t node_base,
                      TriRec* d_tris, uint32_t tri_base, uint32_t tri_count, uint32_t max_depth, uint32_t cluster_size,
                      unsigned threads, LbvhResult* out, const RawTriRun* runs, size_t n_runs, float pad) {
    out->depth = 0;
    out->build_ms = 0.f;
    out->clusters = 0;
    out->top_ms = 0.f;
    uint64_t raw_tris = 0, largest_run = 0;
    for (size_t r = 0; r < n_runs; ++r) {
        raw_tris += runs[r].count;
        largest_run = std::max<uint64_t>(largest_run, runs[r].count);
    }
    if ((uint64_t)n_host + raw_tris < 2u || (uint64_t)n_host + raw_tris > REF_OFFSET_MASK || raw_tris > tri_count) return hipErrorInvalidValue;
    const uint32_t n = n_host + (uint32_t)raw_tris;
    Buffers B;
    BuildPrim* d_prims;
    unsigned long long *keys_a, *keys_b;
    uint32_t *vals_a, *vals_b, *is_tri, *tri_prefix;
    LeafInfo* leaf;
    Topo* topo;
    NodeState* st;
    TriRec* tri_tmp;
    LBVH_TRY(B.alloc(&d_prims, n));
    LBVH_TRY(B.alloc(&keys_a, n));
    LBVH_TRY(B.alloc(&keys_b, n));
    LBVH_TRY(B.alloc(&vals_a, n));
    LBVH_TRY(B.alloc(&vals_b, n));
    LBVH_TRY(B.alloc(&is_tri, (size_t)n + 1));
    LBVH_TRY(B.alloc(&tri_prefix, (size_t)n + 1));
    LBVH_TRY(B.alloc(&leaf, n));
    LBVH_TRY(B.alloc(&topo, n));
    LBVH_TRY(B.alloc(&st, n));
    LBVH_TRY(B.alloc(&tri_tmp, tri_count));
    int* parent;
    LBVH_TRY(B.alloc(&parent, n));
    // buffers of the cut: about 2 n / cluster_size clusters are to be expected; a tree that yields more than
    // 8 n / cluster_size + 1024 keeps its radix topology
    const bool cut = cluster_size >= 2u && n > 2u * cluster_size;
    const uint32_t capacity = cut ? (uint32_t)std::min<uint64_t>(n, 8ull * n / cluster_size + 1024ull) : 0u;
    uint32_t *d_upper = nullptr, *d_counters = nullptr;
    Cluster* d_clusters = nullptr;
    if (cut) {
        LBVH_TRY(B.alloc(&d_upper, capacity));
        LBVH_TRY(B.alloc(&d_clusters, capacity));
        LBVH_TRY(B.alloc(&d_counters, 4));
    }
    size_t sort_bytes = 0, scan_bytes = 0;
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 48, (hipStream_t)0));
    LBVH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, is_tri, tri_prefix, (int)(n + 1u), (hipStream_t)0));
    unsigned char* temp;
    LBVH_TRY(B.alloc(&temp, sort_bytes > scan_bytes ? sort_bytes : scan_bytes));
    // RRT_LBVH_SAH=0: keep the radix topology below the cut
    bool sah = cut && cluster_size <= (uint32_t)CL_MAX;
    if (const char* e = getenv("RRT_LBVH_SAH")) sah = sah && atoi(e) != 0;
    // RRT_LBVH_GROUP (default 0: none): a second round of the subtree kernel groups up to that many finished subtrees
    // (at most 64), so that the host arranges a few thousand items instead of a few hundred thousand.  Measured on
    // MI355X: the GPU part of a ten-million-triangle build falls from 145-185 ms to 55-65 ms, and the tree renders 3 %
    // slower (one node visit per ray more, whatever the group size: the groups are radix subtrees, and what the radix
    // tree puts together above the subtrees is what the host's builder otherwise arranges freely).
    uint32_t group_size = 0u;
    if (const char* e = getenv("RRT_LBVH_GROUP")) group_size = (uint32_t)std::min(std::max(atoi(e), 0), CL_MAX);
    const bool groups = sah && group_size >= 2u && (uint64_t)n > 4ull * cluster_size * group_size;
    uint32_t *cl_roots = nullptr, *cl_counters = nullptr, *weight = nullptr;
    // counters: [0] subtrees to rebuild, [1] set if one could not be rebuilt, [2] groups
    if (sah) {
        LBVH_TRY(B.alloc(&cl_roots, n));
        LBVH_TRY(B.alloc(&cl_counters, 4));
        if (groups) LBVH_TRY(B.alloc(&weight, n));
    }
    LBVH_TRY(hipEventCreate(&B.e0));
    LBVH_TRY(hipEventCreate(&B.e1));

    if (n_host) LBVH_TRY(hipMemcpy(d_prims, h_prims, (size_t)n_host * sizeof(BuildPrim), hipMemcpyHostToDevice));
    const hipStream_t s = (hipStream_t)0;
    if (raw_tris) {
        // the caller's vertices go up run by run through one staging buffer; records and boxes are made here
        rrt_triangle* staging;
        LBVH_TRY(B.alloc(&staging, (size_t)largest_run));
        uint32_t ordinal = tri_count - (uint32_t)raw_tris;  // the host-expanded triangles come first
        uint32_t at = n_host;
        for (size_t r = 0; r < n_runs; ++r) {
            const RawTriRun& run = runs[r];
            if (run.count == 0u) continue;
            LBVH_TRY(hipMemcpy(staging, run.src, (size_t)run.count * sizeof(rrt_triangle), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_expand_tris, dim3((run.count + BLOCK - 1u) / BLOCK), dim3(BLOCK), 0, s, staging, run.count, run.id0, ordinal,
                               pad, d_prims + at, d_tris + tri_base);
            LBVH_TRY(hipGetLastError());
            LBVH_TRY(hipStreamSynchronize(s));  // the staging buffer is reused by the next run
            ordinal += run.count;
            at += run.count;
        }
    }
    const unsigned blocks = (n + BLOCK - 1u) / BLOCK, blocks1 = (n + 1u + BLOCK - 1u) / BLOCK;
    LBVH_TRY(hipEventRecord(B.e0, s));
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(BLOCK), 0, s, d_prims, n, grid, keys_a, vals_a);
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(temp, sort_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 48, s));
    hipLaunchKernelGGL(k_gather, dim3(blocks1), dim3(BLOCK), 0, s, d_prims, vals_b, n, leaf, is_tri);
    LBVH_TRY(hipcub::DeviceScan::ExclusiveSum(temp, scan_bytes, is_tri, tri_prefix, (int)(n + 1u), s));
    hipLaunchKernelGGL(k_topology, dim3(blocks), dim3(BLOCK), 0, s, keys_b, (int)n, topo, parent);
    LBVH_TRY(hipMemsetAsync(st, 0, (size_t)n * sizeof(NodeState), s));
    // a tree whose collapsed depth fits max_depth has at most max_depth + 7 radix-tree levels (a collapsed subtree
    // of at most MAX_LEAF_TRIS = 8 leaves is at most 7 levels high)
    const uint32_t launches = max_depth + 7u;
    uint32_t leaf_tris = LBVH_LEAF_TRIS_DEFAULT;
    if (const char* e = getenv("RRT_LBVH_LEAF")) leaf_tris = (uint32_t)std::min(std::max(atoi(e), 1), (int)MAX_LEAF_TRIS);
    uint32_t sah_failed = 0u;
    if (sah) {
        uint32_t sah_leaf = 4u;  // the host builder's bound (RRT_LEAF_TRIS there)
        if (const char* e = getenv("RRT_LBVH_SAH_LEAF")) sah_leaf = (uint32_t)std::min(std::max(atoi(e), 1), (int)MAX_LEAF_TRIS);
        float node_cost = 1.0f;
        if (const char* e = getenv("RRT_SAH_TCOST")) node_cost = (float)atof(e);
        LBVH_TRY(hipMemsetAsync(cl_counters, 0, 4 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_cluster_roots, dim3(blocks), dim3(BLOCK), 0, s, topo, parent, tri_prefix, n, cluster_size, leaf_tris, cl_roots, cl_counters);
        LBVH_TRY(hipGetLastError());
        uint32_t n_roots = 0;
        LBVH_TRY(hipMemcpy(&n_roots, cl_counters, sizeof n_roots, hipMemcpyDeviceToHost));
        if (n_roots > n - 1u) return hipErrorUnknown;
        if (n_roots) {
            hipLaunchKernelGGL(k_subtree_sah<1>, dim3((n_roots + CL_WAVES - 1u) / CL_WAVES), dim3(CL_MAX * CL_WAVES), 0, s, cl_roots, cl_counters, topo,
                               leaf, tri_prefix, st, d_nodes, grid, node_base, tri_base, sah_leaf, node_cost, cluster_size, leaf_tris, cl_counters + 1);
            LBVH_TRY(hipGetLastError());
        }
    }
    // (the rebuilt subtrees carry launch number 1: the levels above them see them as finished from launch 2 on)
    for (uint32_t k = 2; k <= launches + 1u; ++k)
        hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(BLOCK), 0, s, topo, parent, leaf, tri_prefix, st, d_nodes, n, k, grid, node_base, tri_base,
                           leaf_tris, sah ? cluster_size : 0u, weight);
    if (groups) {
        hipLaunchKernelGGL(k_group_roots, dim3(blocks), dim3(BLOCK), 0, s, topo, parent, weight, n, cluster_size, group_size, cl_roots, cl_counters + 2);
        LBVH_TRY(hipGetLastError());
        uint32_t n_groups = 0;
        LBVH_TRY(hipMemcpy(&n_groups, cl_counters + 2, sizeof n_groups, hipMemcpyDeviceToHost));
        if (n_groups > n - 1u) return hipErrorUnknown;
        if (n_groups) {
            hipLaunchKernelGGL(k_subtree_sah<2>, dim3((n_groups + CL_WAVES - 1u) / CL_WAVES), dim3(CL_MAX * CL_WAVES), 0, s, cl_roots, cl_counters + 2,
                               topo, leaf, tri_prefix, st, d_nodes, grid, node_base, tri_base, 1u, 1.0f, cluster_size, leaf_tris, cl_counters + 1);
            LBVH_TRY(hipGetLastError());
        }
    }
    if (tri_count) {
        LBVH_TRY(hipMemcpyAsync(tri_tmp, d_tris + tri_base, (size_t)tri_count * sizeof(TriRec), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_permute_tris, dim3(blocks), dim3(BLOCK), 0, s, leaf, tri_prefix, n, tri_tmp, tri_count, d_tris + tri_base);
    }
    LBVH_TRY(hipEventRecord(B.e1, s));
    LBVH_TRY(hipGetLastError());
    NodeState root;
    LBVH_TRY(hipMemcpy(&root, st, sizeof root, hipMemcpyDeviceToHost));  // also waits for the stream
    LBVH_TRY(hipEventElapsedTime(&out->build_ms, B.e0, B.e1));
    if (sah) LBVH_TRY(hipMemcpy(&sah_failed, cl_counters + 1, sizeof sah_failed, hipMemcpyDeviceToHost));
    if (sah_failed != 0u) return hipSuccess;  // (depth 0: the caller builds this scene on the host)
    if (root.done != 0u && root.depth <= max_depth) out->depth = root.depth;
    if (cut) {
        // the radix tree above the cut need not fit the depth bound (nor be finished): only the clusters must
        const auto t0 = std::chrono::steady_clock::now();
        LBVH_TRY(hipMemsetAsync(d_counters, 0, 4 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_cut, dim3(blocks), dim3(BLOCK), 0, s, topo, leaf, tri_prefix, st, n, cluster_size, node_base, tri_base,
                           leaf_tris, d_upper, d_clusters, capacity, d_counters, (const uint32_t*)weight, group_size);
        LBVH_TRY(hipGetLastError());
        uint32_t depth = 0, n_clusters = 0;
        LBVH_TRY(rebuild_top(d_upper, d_clusters, d_counters, capacity, grid, d_nodes, node_base, max_depth, threads, B, &depth, &n_clusters));
        if (depth != 0u) {
            out->depth = depth;
            out->clusters = n_clusters;
        } else if (groups) {
            out->depth = 0u;  // the radix tree's depth was taken before the groups were rebuilt: leave the scene to the host
        }
        out->top_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (out->depth != 0u)
        if (const char* e = getenv("RRT_LBVH_VERIFY"))
            if (atoi(e) != 0) LBVH_TRY(verify_tree(d_nodes, node_base, n, tri_base, tri_count, out->depth));
    return hipSuccess;
}

}  // namespace rrt
