// lbvh_build.hip -- linear BVH on the GPU: 48-bit Morton codes -> radix sort -> radix-tree topology (every inner node
// from the sorted codes alone, Karras 2012) -> boxes bottom-up, one kernel launch per tree level.
//
// The bottom-up pass hands boxes from children to parents only ACROSS kernel launches: a node is finished in
// launch k if both children were finished in launches < k.  Inside one launch nothing written by another
// workgroup is consumed, so no fences or cache write-backs are needed (the L2s of the eight XCDs are not coherent
// with each other within a launch).  A tree of h levels takes h launches of a few microseconds each.
#include "lbvh_build.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace rrt {
namespace {

constexpr uint32_t LBVH_LEAF_TRIS = 4;  // a subtree of at most this many primitives, all triangles, becomes one leaf
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

__global__ void k_topology(const unsigned long long* __restrict__ keys, int n, Topo* __restrict__ topo) {
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
}

struct ChildView {
    bool ready;
    float mn[3], mx[3];
    int32_t ref;
    uint32_t depth;
};

__device__ __forceinline__ ChildView view_child(int c, const Topo* __restrict__ topo, const LeafInfo* __restrict__ leaf,
                                                const uint32_t* __restrict__ tri_prefix, const NodeState* __restrict__ st,
                                                uint32_t launch, uint32_t node_base, uint32_t tri_base) {
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
    const uint32_t cnt = (uint32_t)(t.last - t.first + 1);
    if (cnt <= LBVH_LEAF_TRIS && tri_prefix[t.last + 1] - tri_prefix[t.first] == cnt) {
        v.ref = (int32_t)make_leaf_ref(KIND_TRI, tri_base + tri_prefix[t.first], cnt);
        v.depth = 0u;
    } else {
        v.ref = (int32_t)(node_base + (uint32_t)c);
        v.depth = s.depth;
    }
    return v;
}

__global__ void k_refit(const Topo* __restrict__ topo, const LeafInfo* __restrict__ leaf, const uint32_t* __restrict__ tri_prefix,
                        NodeState* __restrict__ st, QNode* __restrict__ nodes, uint32_t n, uint32_t launch, Grid g,
                        uint32_t node_base, uint32_t tri_base) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n - 1u) return;
    if (st[i].done != 0u) return;
    const Topo t = topo[i];
    const ChildView l = view_child(t.left, topo, leaf, tri_prefix, st, launch, node_base, tri_base);
    if (!l.ready) return;
    const ChildView r = view_child(t.right, topo, leaf, tri_prefix, st, launch, node_base, tri_base);
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

struct Buffers {
    void* p[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int used = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Buffers() {
        for (int i = 0; i < used; ++i)
            if (p[i]) (void)hipFree(p[i]);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    template <typename T>
    hipError_t alloc(T** out, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) return e;
        p[used++] = q;
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

hipError_t lbvh_build(const BuildPrim* h_prims, uint32_t n, const Grid& grid, QNode* d_nodes, uint32_t node_base,
                      TriRec* d_tris, uint32_t tri_base, uint32_t tri_count, uint32_t max_depth, LbvhResult* out) {
    out->depth = 0;
    out->build_ms = 0.f;
    if (n < 2u || n > REF_OFFSET_MASK) return hipErrorInvalidValue;
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
    size_t sort_bytes = 0, scan_bytes = 0;
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 48, (hipStream_t)0));
    LBVH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, is_tri, tri_prefix, (int)(n + 1u), (hipStream_t)0));
    unsigned char* temp;
    LBVH_TRY(B.alloc(&temp, sort_bytes > scan_bytes ? sort_bytes : scan_bytes));
    LBVH_TRY(hipEventCreate(&B.e0));
    LBVH_TRY(hipEventCreate(&B.e1));

    LBVH_TRY(hipMemcpy(d_prims, h_prims, (size_t)n * sizeof(BuildPrim), hipMemcpyHostToDevice));
    const hipStream_t s = (hipStream_t)0;
    const unsigned blocks = (n + BLOCK - 1u) / BLOCK, blocks1 = (n + 1u + BLOCK - 1u) / BLOCK;
    LBVH_TRY(hipEventRecord(B.e0, s));
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(BLOCK), 0, s, d_prims, n, grid, keys_a, vals_a);
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(temp, sort_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 48, s));
    hipLaunchKernelGGL(k_gather, dim3(blocks1), dim3(BLOCK), 0, s, d_prims, vals_b, n, leaf, is_tri);
    LBVH_TRY(hipcub::DeviceScan::ExclusiveSum(temp, scan_bytes, is_tri, tri_prefix, (int)(n + 1u), s));
    hipLaunchKernelGGL(k_topology, dim3(blocks), dim3(BLOCK), 0, s, keys_b, (int)n, topo);
    LBVH_TRY(hipMemsetAsync(st, 0, (size_t)n * sizeof(NodeState), s));
    // a tree whose collapsed depth fits max_depth has at most max_depth + 3 radix-tree levels (a collapsed subtree
    // of at most 4 leaves is at most 3 levels high)
    const uint32_t launches = max_depth + 3u;
    for (uint32_t k = 1; k <= launches; ++k)
        hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(BLOCK), 0, s, topo, leaf, tri_prefix, st, d_nodes, n, k, grid, node_base, tri_base);
    if (tri_count) {
        LBVH_TRY(hipMemcpyAsync(tri_tmp, d_tris + tri_base, (size_t)tri_count * sizeof(TriRec), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_permute_tris, dim3(blocks), dim3(BLOCK), 0, s, leaf, tri_prefix, n, tri_tmp, tri_count, d_tris + tri_base);
    }
    LBVH_TRY(hipEventRecord(B.e1, s));
    LBVH_TRY(hipGetLastError());
    NodeState root;
    LBVH_TRY(hipMemcpy(&root, st, sizeof root, hipMemcpyDeviceToHost));  // also waits for the stream
    LBVH_TRY(hipEventElapsedTime(&out->build_ms, B.e0, B.e1));
    if (root.done != 0u && root.depth <= max_depth) out->depth = root.depth;
    return hipSuccess;
}

}  // namespace rrt
