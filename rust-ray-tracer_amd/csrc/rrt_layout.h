// rrt_layout.h -- layout of the scene as it lives in HBM (shared by the host builder and the kernels).
//
// Everything a ray touches during traversal is a 16-byte-aligned record read with
// global_load_dwordx4:
//   QNode     32 B  both child references + both child boxes on a 16-bit grid (the idea of the reference's
//                   unfinished flat BVH, src/bvh/mod.rs:19-25: a node stores its children's boxes)
//   TriRec    48 B  v0 | id, v0v1 | material, v0v2 | -      (reference src/geometry/triangle.rs:13-20)
//   SphereRec 32 B  centre, radius | id, material           (reference src/geometry/sphere.rs:17-22)
//   RectRec   32 B  a0,a1,b0,b1 | k, axis+flags, id, material (reference src/geometry/aarect.rs:14-21)
//   InstRec   64 B  Translate(RotateY(mesh)) + the mesh BVH's grid (reference src/geometry/transform.rs:9-12,44-49)
//   MatRec    32 B  kind, rgb, param                        (reference src/material/mod.rs:47-194)
#ifndef RRT_LAYOUT_H
#define RRT_LAYOUT_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RRT_HD __host__ __device__
#else
#define RRT_HD
#endif

namespace rrt {

// Child reference encoding (int32):
//   >= 0                 : index of an inner node
//   bit31 set            : leaf;  bits 30..28 = kind, bits 27..25 = count-1 (triangles), bits 24..0 = first record
static constexpr uint32_t REF_LEAF = 0x80000000u;
static constexpr uint32_t KIND_TRI = 0u, KIND_SPHERE = 1u, KIND_RECT = 2u, KIND_INST = 3u;
static constexpr uint32_t KIND_SUBTREE = 4u;    // builder-internal: a primitive that stands for a finished subtree (never stored in a node)
static constexpr uint32_t KIND_EMPTY = 5u;      // child slot without content (scenes with fewer than two primitives)
static constexpr uint32_t KIND_INST_EXIT = 6u;  // traversal stack marker: leave the instance's object space
static constexpr uint32_t KIND_DONE = 7u;       // traversal stack marker: bottom of stack / empty child
static constexpr uint32_t REF_OFFSET_MASK = 0x01FFFFFFu;
static constexpr uint32_t MAX_LEAF_TRIS = 8u;
static constexpr int32_t REF_DONE = (int32_t)(REF_LEAF | (KIND_DONE << 28));
static constexpr int32_t REF_INST_EXIT = (int32_t)(REF_LEAF | (KIND_INST_EXIT << 28));

// Reference to an inner node as the device sees it: the node's byte offset from the start of the node array
// (index * 32 < 2^30, so it stays non-negative); a fetch needs no shift.  Leaf references pass through unchanged.
RRT_HD static inline int32_t device_ref(int32_t index_or_leaf) {
    return index_or_leaf >= 0 ? (int32_t)((uint32_t)index_or_leaf << 5) : index_or_leaf;
}

RRT_HD static inline uint32_t make_leaf_ref(uint32_t kind, uint32_t first, uint32_t count) {
    return REF_LEAF | (kind << 28) | (((count - 1u) & 7u) << 25) | (first & REF_OFFSET_MASK);
}

// Host-side node as the builder produces it (full-precision child boxes); never uploaded.
struct BvhNode {
    float lmin[3];
    float lmax[3];
    float rmin[3];
    float rmax[3];
    int32_t left, right;
};

// Every BVH (the top level and each mesh) carries a grid of 65536 planes per axis, plane(q) = origin + q * step,
// that covers its root box.  A child box is stored as plane indices rounded outwards, so the stored box contains
// the builder's box, and a node shrinks to two 16-byte loads.  The traversal never forms plane(q) itself: with
// A = step / dir and B = (origin - ray origin) / dir per axis, the plane's ray distance is fma(q, A, B).
struct Grid {
    float origin[3];
    float step[3];
};

struct alignas(16) QNode {
    int32_t left, right;      // child references: device_ref() of a node index, or a leaf reference
    uint32_t lbox[3];         // left child, per axis: bits 15..0 = min plane index, bits 31..16 = max plane index
    uint32_t rbox[3];         // right child
};
static_assert(sizeof(QNode) == 32, "QNode must be 32 bytes");

// Plane indices of one child box [mn, mx] on grid g, rounded outwards to the nearest plane: the stored box contains
// the given one (the quotient is formed in double precision; its rounding error, 1e-16 of a plane index, is far
// inside the margin every box already carries).  Used by the host builder and by the on-device builder alike.
RRT_HD static inline void quantize_box(const Grid& g, const float* mn, const float* mx, uint32_t* out) {
    for (int a = 0; a < 3; ++a) {
        const double o = g.origin[a], s = g.step[a];
        double qlo = floor(((double)mn[a] - o) / s);
        double qhi = ceil(((double)mx[a] - o) / s);
        if (!(qlo >= 0.0)) qlo = 0.0;          // also catches the NaN of an empty slot's sentinel box
        if (!(qhi <= 65535.0)) qhi = 65535.0;
        if (qlo > 65535.0) qlo = 65535.0;
        if (qhi < 0.0) qhi = 0.0;
        out[a] = (uint32_t)qlo | ((uint32_t)qhi << 16);
    }
}

struct alignas(16) TriRec {
    float v0[3];   uint32_t id;
    float e1[3];   uint32_t material;  // v0v1 = v1 - v0
    float e2[3];   uint32_t pad;       // v0v2 = v2 - v0
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

struct alignas(16) SphereRec {
    float c[3];    float r;
    uint32_t id;   uint32_t material;  uint32_t pad0, pad1;
};
static_assert(sizeof(SphereRec) == 32, "SphereRec must be 32 bytes");

struct alignas(16) RectRec {
    float a0, a1, b0, b1;
    float k;       uint32_t axis_flags;  // bits 1..0 axis, bit 8 flip-face
    uint32_t id;   uint32_t material;
};
static_assert(sizeof(RectRec) == 32, "RectRec must be 32 bytes");

struct alignas(16) InstRec {
    float offset[3];   uint32_t id_base;   // ids of the instance's primitives = id_base + local id
    float sin_t, cos_t; int32_t root;      // root node of the mesh's BVH
    uint32_t pad0;
    float grid_origin[3]; uint32_t pad1;   // grid of the mesh's BVH (object space)
    float grid_step[3];   uint32_t pad2;
};
static_assert(sizeof(InstRec) == 64, "InstRec must be 64 bytes");

struct alignas(16) MatRec {
    uint32_t kind;  float r, g, b;
    float param;    uint32_t pad0, pad1, pad2;
};
static_assert(sizeof(MatRec) == 32, "MatRec must be 32 bytes");

struct alignas(16) LightRec {
    uint32_t type;  float p0, p1, p2;
    float p3, p4;   uint32_t pad0, pad1;
};
static_assert(sizeof(LightRec) == 32, "LightRec must be 32 bytes");

// Local primitive ids inside a mesh are stored in TriRec/SphereRec/RectRec.id; the instance adds id_base.

// Camera + globals passed to kernels by value.
struct CameraRec {
    float c0[3], c1[3], c2[3], tr[3];  // camera_to_world columns + translation
    float origin[3];
    float scale, aspect, time0, time1;
};

struct SceneDev {
    const QNode* nodes;
    const TriRec* tris;
    const SphereRec* spheres;
    const RectRec* rects;
    const InstRec* insts;     // insts[-1] is an identity record (what a triangle outside every instance reads in an unrolled scene)
    const MatRec* mats;
    const LightRec* lights;
    uint32_t light_count;
    int32_t root;       // root node of the top-level BVH
    uint32_t has_instances;   // instance leaves in the top-level BVH (two-level walk)
    uint32_t inst_count;      // instance records
    uint32_t unroll_shift;    // != 0: instances unrolled into one world-space tree; a triangle leaf's 25-bit field is
                              // (instance slot << unroll_shift) | first record, slot 0 = not instanced
    uint32_t light_zone;  // 2^32 mod light_count (rejection zone of the uniform light pick)
    CameraRec cam;
    float background[3];
    float t_min;
    uint32_t max_depth;
    uint32_t seed;
    float light_weight;  // 1.0f / light_count
    Grid grid;          // grid of the top-level BVH (world space)
    // scaled slab arithmetic of the render kernels (rrt_device.h, trav_set_ray_scaled): sigma = the power of two that
    // puts the largest of grid.step * sigma * 2^149 into [2^48, 2^49); step_scaled = grid.step * sigma * 2^149
    float step_scaled[3];
    float sigma;
};

// Launch geometry of the render kernel.  The frame is cut into 8x8 pixel tiles and the sample range into
// `n_chunks` (<= spp) chunks whose lengths differ by at most one: with q = spp / n_chunks and r = spp % n_chunks, chunk c
// holds the q + (c < r) samples that start at sample c * q + min(c, r), so any sample count cuts into any number of
// chunks without an empty or a short last one.  One work item = (tile, chunk), item index = tile * n_chunks + chunk,
// so that the chunks of an expensive tile are handed out back to back and run on many CUs at the same time.
struct RenderGeom {
    uint32_t width, height, spp, spp_offset;
    uint32_t tiles_x, n_tiles, n_chunks, chunk_q;
    // Tiles are numbered band by band: vertical bands of `band_tiles` tiles' width, row by row inside a band, so that
    // the items in flight at one time cover a compact block of the image instead of a strip of one tile's height.
    uint32_t band_tiles;        // width of a full band in tiles
    uint32_t band_items;        // tiles in a full band (band_tiles * rows of tiles)
    uint32_t full_bands;        // bands of full width; one narrower band may follow
    uint32_t n_work;
    uint32_t node_share_64ths;  // a node pass runs while at least this share (in 64ths) of the walking lanes descends
    uint32_t leave_walk;        // a trip's walk phase ends when at most this many lanes still walk ...
    uint32_t leave_ready;       // ... and at least this many lanes wait to shade (leave_walk 0: walk to the end)
    uint32_t n_groups;          // work items per tile (pipelined form: an item holds several consecutive chunks of its pixels)
    uint32_t pix_log2, tw_log2; // pipelined form: an item is 2^pix_log2 pixels (a tile 2^tw_log2 pixels wide) x 64 / pixels chunks
    uint32_t chunk_r;           // the first chunk_r chunks hold chunk_q + 1 samples, the others chunk_q
    const uint32_t* lists;      // per-pixel candidate lists (PIXEL_LIST_DWORDS dwords per pixel, row-major), or null
};
// A pixel's list of candidates for its primary rays, as k_pixel_lists writes it: dword 0 = number of entries (or
// 0xFFFFFFFF: no list, the pixel's primary rays walk the tree), dword 1 unused, then per entry a leaf reference (a
// triangle as a triangle leaf of one; a sphere or rectangle leaf as it is) and the bit pattern of a lower bound of the
// ray parameter at which a ray of the pixel can reach the box of the leaf the entry came from, in ascending order of
// that bound.
static constexpr uint32_t PIXEL_LIST_MAX = 11u;
static constexpr uint32_t PIXEL_LIST_DWORDS = 2u * (PIXEL_LIST_MAX + 1u);
// First sample (relative to the launch's first) and length of chunk c of a launch cut as RenderGeom describes.
RRT_HD static inline uint32_t chunk_first(uint32_t c, uint32_t q, uint32_t r) { return c * q + (c < r ? c : r); }
RRT_HD static inline uint32_t chunk_len(uint32_t c, uint32_t q, uint32_t r) { return q + (c < r ? 1u : 0u); }

// Per-lane traversal stack entries (LDS).  One inner-node step pushes at most one entry, so the stack never
// holds more than (tree depth) entries plus the bottom marker and, for instanced scenes, one exit marker;
// the builder bounds the combined depth of top level + bottom level to MAX_TREE_DEPTH.  39 entries x 256 lanes x
// 4 bytes = 39 KB per workgroup, which leaves the render kernel 1 KB for its work-item records: four workgroups
// fill the 160 KB of LDS of a CU, and four are as fast as five (measured); the depth leaves room for the linear
// BVHs of the GPU builder (33 levels for ten million triangles).
// One entry beyond the deepest walk stays free: the branch-free node step writes a value nobody reads above the top.
static constexpr int STACK_DEPTH = 39;
static constexpr uint32_t MAX_TREE_DEPTH = 36;
static_assert(MAX_TREE_DEPTH + 3 <= (uint32_t)STACK_DEPTH, "stack holds the deepest walk, its two markers and one free entry");

}  // namespace rrt
#endif
