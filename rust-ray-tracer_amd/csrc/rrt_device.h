// rrt_device.h -- device functions of the path tracer (gfx950).  Compiled with -ffp-contract=off:
// every expression below rounds exactly like the scalar f32 expression it restates, so that
// primitive tests give the same accept/reject decisions and the same t as the reference arithmetic
// (glam 0.20 Vec3A on SSE2: dot = (x*x' + y*y') + z*z', cross = mul, mul, sub per component,
// normalize = v / sqrt(dot(v, v)); no fused multiply-add anywhere).
#ifndef RRT_DEVICE_H
#define RRT_DEVICE_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "rrt_layout.h"

namespace rrt {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 mulv(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }  // vec3.rs:5-7
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length_squared(V3 a) { return dot(a, a); }
__device__ __forceinline__ float norm1(V3 a) { return (fabsf(a.x) + fabsf(a.y)) + fabsf(a.z); }
__device__ __forceinline__ V3 normalize(V3 a);  // defined after sqrt_rn
__device__ __forceinline__ float comp(V3 a, uint32_t axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }
#define RRT_PI 3.14159274101257324f  // std::f32::consts::PI
#define RRT_INV_PI 0.318309886183790672f
// Rust f32::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN
__device__ __forceinline__ float signum(float x) { return x != x ? x : copysignf(1.0f, x); }

// Shading arithmetic.  Normalisations and the final path weight use the hardware reciprocal and reciprocal square
// root (one instruction, 1 ulp) in place of the correctly rounded IEEE sequences (about ten instructions each): an
// ulp there moves a sample by far less than its Monte-Carlo noise.  Everything that decides WHICH surface a ray
// meets (the primitive tests, the light's own hit test) and the light pdf, whose 1 - cos(theta_max) cancels
// leading digits, keep IEEE division and square root, as the reference's f32 arithmetic has them.
__device__ __forceinline__ float rcp_1ulp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq_1ulp(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ V3 unit(V3 a) { return a * rsq_1ulp(dot(a, a)); }  // glam normalize, to 1-2 ulp

// Correctly rounded 1/x and sqrt(x) in short form.  The compiler's IEEE sequences take 11 and 16 instructions (range
// scaling, two correction steps, fix-ups).  For operands away from the ends of the exponent range one Newton step in
// fused arithmetic on the hardware estimate (1 ulp) already gives the correctly rounded result: k_check_arith in
// rrt_kernels.hip compares both functions with the IEEE operations for EVERY one of the 2^32 float bit patterns on
// the device (tests/test_gpu_arith.py).  Operands outside the range the short form is exact on (zero, denormals,
// negative numbers, infinities, NaN, the last binades) take the IEEE path, so results are bit-identical everywhere.
__device__ __forceinline__ float recip_rn(float x) {
    // positive normal x below 2^126 (above it the quotient is denormal)
    if (__float_as_uint(x) - 0x00800000u < 0x7E800000u - 0x00800000u) {
        const float r = __builtin_amdgcn_rcpf(x);
        return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    }
    return 1.0f / x;
}
__device__ __forceinline__ float sqrt_rn(float x) {
    // finite x of at least 2^-100 (below it the residual of the Newton step is denormal)
    if (__float_as_uint(x) - 0x0D800000u < 0x7F800000u - 0x0D800000u) {
        const float y = __builtin_amdgcn_rsqf(x);
        const float s = x * y;
        return __builtin_fmaf(__builtin_fmaf(-s, s, x), 0.5f * y, s);
    }
    return sqrtf(x);
}

__device__ __forceinline__ V3 normalize(V3 a) { return a / sqrt_rn(dot(a, a)); }

// (sin, cos) of 2*pi*turn for turn in [0, 1): the nearest quarter turn is removed exactly, then the Cephes
// single-precision polynomials on [-pi/4, pi/4] (about 1 ulp), then the quadrant is put back.
__device__ __forceinline__ void sincos_turn(float turn, float& sn, float& cs) {
    const float q = rintf(turn * 4.0f);
    const float x = __builtin_fmaf(q, -0.25f, turn) * (2.0f * RRT_PI);  // the subtraction is exact
    const float z = x * x;
    const float s = __builtin_fmaf(x * z, __builtin_fmaf(z, __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), x);
    const float c = __builtin_fmaf(z * z, __builtin_fmaf(z, __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                                   __builtin_fmaf(z, -0.5f, 1.0f));
    const uint32_t k = (uint32_t)(int)q;
    const float a = (k & 1u) ? c : s, b = (k & 1u) ? s : c;
    sn = __uint_as_float(__float_as_uint(a) ^ ((k & 2u) << 30));
    cs = __uint_as_float(__float_as_uint(b) ^ (((k + 1u) & 2u) << 30));
}

// ---------------------------------------------------------------------------------------------
// Counter-based RNG: the i-th draw of sample `s` of pixel `p` under seed `k` is a pure function
// of (k, p, s, i).  Replaces the reference's thread-local ChaCha12 `thread_rng()` whose stream
// is OS-seeded and therefore not reproducible (reference src/main.rs:88, src/camera.rs:57 ...).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t rng_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// The state keeps k0 + ctr * 0x9E3779B9 as one running sum (`acc`): a draw costs an add instead of a multiply and two
// adds, and the state is two registers.
struct Rng {
    uint32_t acc, k1;
};
__host__ __device__ __forceinline__ void rng_skip(Rng& r) { r.acc += 0x9E3779B9U; }  // a draw whose value is not used
__host__ __device__ __forceinline__ Rng rng_init(uint32_t seed, uint32_t pixel, uint32_t sample) {
    Rng r;
    uint32_t h0 = rng_mix(seed ^ 0xA511E9B3U);
    uint32_t h1 = rng_mix(h0 + pixel);
    r.acc = rng_mix(h1 ^ (sample * 0x9E3779B9U));  // k0; ctr = 0
    r.k1 = rng_mix(r.acc ^ 0x68E31DA4U ^ pixel);
    return r;
}
__host__ __device__ __forceinline__ uint32_t rng_u32(Rng& r) {
    uint32_t x = r.acc;
    r.acc += 0x9E3779B9U;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x ^= r.k1; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// rand 0.8 `Standard` for f32: 24 random bits scaled into [0, 1)
__device__ __forceinline__ float rng_f32(Rng& r) { return (float)(rng_u32(r) >> 8) * (1.0f / 16777216.0f); }
// rand 0.8 `gen_range(lo..hi)` for f32 (UniformFloat::sample_single): 23 bits into [1,2), minus 1, times scale, plus lo;
// a result rounded up to hi is drawn again with scale one ulp smaller.  rand's loop has no bound; this one gives up
// after 8 draws (with the decreasing scale a second failure needs the largest of the 2^23 values twice in a row).
__device__ __forceinline__ float rng_range(Rng& r, float lo, float hi) {
    float scale = hi - lo;
    for (int tries = 0; tries < 8; ++tries) {
        float v12 = __uint_as_float((rng_u32(r) >> 9) | 0x3f800000U);
        float res = (v12 - 1.0f) * scale + lo;
        if (res < hi) return res;
        scale = __uint_as_float(__float_as_uint(scale) - 1u);
    }
    return lo;
}
// rand 0.8 `gen_bool(0.5)`: a u64 draw compared with 2^63, i.e. one fair bit
__device__ __forceinline__ bool rng_bool_half(Rng& r) { return (rng_u32(r) >> 31) == 0u; }
// rand 0.8 `gen_range(0..n)`: unbiased integer in [0, n) (widening multiply with rejection); zone = 2^32 mod n is
// computed once on the host (an integer division per call would cost more than the draw)
__device__ __forceinline__ uint32_t rng_index(Rng& r, uint32_t n, uint32_t zone) {
    for (;;) {
        uint64_t m = (uint64_t)rng_u32(r) * (uint64_t)n;
        if ((uint32_t)m >= zone) return (uint32_t)(m >> 32);
    }
}

// ---------------------------------------------------------------------------------------------
// Orthonormal basis -- reference src/vec3.rs:55-73 with glam's any_orthonormal_pair (Duff et al.)
// ---------------------------------------------------------------------------------------------
struct Onb {
    V3 u, v, w;
};
__device__ __forceinline__ Onb onb_from_w(V3 n) {
    Onb o;
    o.w = unit(n);
    const float sign = copysignf(1.0f, o.w.z);
    const float a = -rcp_1ulp(sign + o.w.z);
    const float b = o.w.x * o.w.y * a;
    o.u = v3(1.0f + sign * o.w.x * o.w.x * a, sign * b, -sign * o.w.x);
    o.v = v3(b, sign + o.w.y * o.w.y * a, -o.w.y);
    return o;
}
__device__ __forceinline__ V3 onb_local(const Onb& o, V3 a) { return (a.x * o.u + a.y * o.v) + a.z * o.w; }

// ---------------------------------------------------------------------------------------------
// Primitive tests.  Each returns true and updates (tbest, best_id, ...) iff the reference test
// accepts the hit in [tmin, tbest] and (t, id) is lexicographically below the current best.
// ---------------------------------------------------------------------------------------------
struct Best {
    float t;        // closest t so far (starts at the query's t_max)
    float u, v;
    uint32_t what;  // kind << 28 | record index of the closest hit; WHAT_NONE: none yet
    int32_t inst;   // instance record the hit lies in, -1 for top-level primitives
};
static constexpr uint32_t WHAT_NONE = 0xFFFFFFFFu;

// Ids of an instance's primitives are id_base + local id (top-level primitives: the local id).  The base is looked up in
// the instance record when it is needed, which is rare against the number of tests; carried in a register through
// the walk it would cost the instanced kernels one of the registers their occupancy depends on.
__device__ __forceinline__ uint32_t instance_id_base(const InstRec* __restrict__ insts, int32_t inst) {
    return inst >= 0 ? insts[inst].id_base : 0u;
}

// Primitive id of a recorded hit.  The closest hit is the minimum of (t, id), but the id only decides between two hits
// at exactly the same t, so the walk does not carry it: it is read from the records when two candidates tie, and when a
// caller wants it reported.
__device__ __forceinline__ uint32_t prim_id_of(const SceneDev& S, uint32_t what, int32_t inst) {
    if (what == WHAT_NONE) return 0xFFFFFFFFu;
    const uint32_t kind = what >> 28, idx = what & REF_OFFSET_MASK;
    const uint32_t local = kind == KIND_TRI ? S.tris[idx].id : (kind == KIND_SPHERE ? S.spheres[idx].id : S.rects[idx].id);
    return instance_id_base(S.insts, inst) + local;
}

// A candidate at t (already known to lie in [tmin, best.t]) with local id `lid` in instance `inst` replaces the closest
// hit so far if it is nearer, or as near with the lower id.
__device__ __forceinline__ bool better(const SceneDev& S, float t, uint32_t lid, int32_t inst, const Best& b) {
    if (t < b.t) return true;
    return instance_id_base(S.insts, inst) + lid < prim_id_of(S, b.what, b.inst);  // t == b.t: rare
}

// reference src/geometry/triangle.rs:38-69
__device__ __forceinline__ void test_triangle(const SceneDev& S, uint32_t idx, V3 o, V3 d, float tmin, int32_t inst, Best& best) {
    // 48-byte records at a 32-bit byte offset from the uniform base
    // (kept as the compiler's one 32-bit multiply: a 24-bit multiply or a shift and a shift-add measured no faster)
    const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.tris) + ((idx << 5) + (idx << 4)));
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
    const V3 v0 = v3(q0.x, q0.y, q0.z), e1 = v3(q1.x, q1.y, q1.z), e2 = v3(q2.x, q2.y, q2.z);
    const V3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    if (det > 1e-5f) {
        const float inv_det = recip_rn(det);  // == 1.0f / det
        const V3 tvec = o - v0;
        const float u = dot(tvec, pvec) * inv_det;
        if (u >= 0.0f && u <= 1.0f) {
            const V3 qvec = cross(tvec, e1);
            const float v = dot(d, qvec) * inv_det;
            if (v >= 0.0f && v < 1.0f - u) {
                const float t = dot(e2, qvec) * inv_det;
                if (t >= tmin && t <= best.t) {
                    if (better(S, t, __float_as_uint(q0.w), inst, best)) {
                        best.t = t; best.u = u; best.v = v;
                        best.what = (KIND_TRI << 28) | idx; best.inst = inst;
                    }
                }
            }
        }
    }
}

// reference src/geometry/sphere.rs:52-95; returns the accepted root or NaN
__device__ __forceinline__ float sphere_root(V3 c, float radius, V3 o, V3 d, float tmin, float tmax) {
    const V3 oc = o - c;
    const float a = length_squared(d);
    const float b = dot(oc, d);
    const float cc = length_squared(oc) - radius * radius;
    const float disc = b * b - a * cc;
    if (disc > 0.0f) {
        const float sq = sqrt_rn(disc);
        float root = (-b - sq) / a;
        if (tmin <= root && root <= tmax) return root;
        root = (-b + sq) / a;
        if (tmin <= root && root <= tmax) return root;
    }
    return __uint_as_float(0x7fc00000U);
}

__device__ __forceinline__ void test_sphere(const SceneDev& S, uint32_t idx, V3 o, V3 d, float tmin, int32_t inst, Best& best) {
    const float4 q0 = *reinterpret_cast<const float4*>(&S.spheres[idx].c[0]);
    const float t = sphere_root(v3(q0.x, q0.y, q0.z), q0.w, o, d, tmin, best.t);
    if (t == t) {
        if (better(S, t, S.spheres[idx].id, inst, best)) {
            best.t = t; best.u = 0.0f; best.v = 0.0f;
            best.what = (KIND_SPHERE << 28) | idx; best.inst = inst;
        }
    }
}

// reference src/geometry/aarect.rs:39-58 (XY), 94-113 (XZ), 170-189 (YZ)
__device__ __forceinline__ void rect_axes(uint32_t axis, uint32_t& ia, uint32_t& ib) {
    // axis 0: YZRect (a = y, b = z); axis 1: XZRect (a = x, b = z); axis 2: XYRect (a = x, b = y)
    ia = axis == 0 ? 1u : 0u;
    ib = axis == 2 ? 1u : 2u;
}
__device__ __forceinline__ bool rect_hit(float a0, float a1, float b0, float b1, float k, uint32_t axis, V3 o, V3 d,
                                         float tmin, float tmax, float& t_out, float& u_out, float& v_out) {
    uint32_t ia, ib;
    rect_axes(axis, ia, ib);
    const float t = (k - comp(o, axis)) / comp(d, axis);
    if (t >= tmin && t <= tmax) {
        const float a = comp(o, ia) + t * comp(d, ia);
        const float b = comp(o, ib) + t * comp(d, ib);
        if (a >= a0 && a <= a1 && b >= b0 && b <= b1) {
            t_out = t;
            u_out = (a - a0) / (a1 - a0);
            v_out = (b - b0) / (b1 - b0);
            return true;
        }
    }
    return false;
}
__device__ __forceinline__ void test_rect(const SceneDev& S, uint32_t idx, V3 o, V3 d, float tmin, int32_t inst, Best& best) {
    const float4 q0 = *reinterpret_cast<const float4*>(&S.rects[idx].a0);
    const float4 q1 = *reinterpret_cast<const float4*>(&S.rects[idx].k);
    float t, u, v;
    if (rect_hit(q0.x, q0.y, q0.z, q0.w, q1.x, __float_as_uint(q1.y) & 3u, o, d, tmin, best.t, t, u, v)) {
        if (better(S, t, __float_as_uint(q1.z), inst, best)) {
            best.t = t; best.u = u; best.v = v;
            best.what = (KIND_RECT << 28) | idx; best.inst = inst;
        }
    }
}

// RotateY then Translate, applied to a ray -- reference src/geometry/transform.rs:24,101-108
__device__ __forceinline__ void to_object_space(const InstRec& in, V3 o, V3 d, V3& oo, V3& od) {
    const V3 m = o - v3(in.offset[0], in.offset[1], in.offset[2]);
    oo = v3(in.cos_t * m.x - in.sin_t * m.z, m.y, in.sin_t * m.x + in.cos_t * m.z);
    od = v3(in.cos_t * d.x - in.sin_t * d.z, d.y, in.sin_t * d.x + in.cos_t * d.z);
}

// ---------------------------------------------------------------------------------------------
// BVH traversal: closest hit = min (t, id) over the reference's per-primitive tests.
// The slab test only has to be conservative (exactness lives in the primitive tests).  Child boxes are stored
// as plane indices q on the grid of their BVH, plane(q) = origin + q * step, and a plane's ray distance is one
// fused multiply-add, t = q * A + B, with A = step * inv, B = (origin - ray origin) * inv and inv the hardware
// reciprocal of the direction (1 ulp) clamped to +-1e30, so that axis-parallel rays give huge finite values
// instead of inf - inf.  The error of t is a few ulp of the distances involved, i.e. about 2^-22 * (scene
// extent) in space -- far inside the 2^-17 * scene-extent margin the builder adds to every box before it
// rounds the box outwards to the grid; a plane's error relative to its t is a displacement of the plane by
// that fraction of its distance from the ray origin, which the same margin covers for every ray that starts
// within 32 scene extents.  The reference's own box test, src/bvh/aabb.rs:14-36, is looser still: it accepts
// every box the ray's three slab intervals each reach.
// A node stores, per axis, the min plane index in the low and the max plane index in the high half of a dword;
// a ray that runs against the axis swaps the halves (one rotate by `rot`), so that the low half is always the
// plane the ray enters through and no per-axis min/max is needed.
// `stack` points at this lane's slot of the LDS stack; entries are `stride` ints apart.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float slab_rcp(float d) {
    return fmaxf(fminf(__builtin_amdgcn_rcpf(d), 1e30f), -1e30f);
}
__device__ __forceinline__ V3 slab_reciprocal(V3 d) { return v3(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z)); }

struct TraceCounters {
    uint32_t nodes, prims;                      // per lane: node steps taken, primitive tests run
    uint32_t wave_nodes, wave_prims, wave_shade; // per wavefront (kept by whichever lane is first active): passes executed
};

// True in exactly one of the currently active lanes.
__device__ __forceinline__ bool first_active_lane() {
    const unsigned long long m = __ballot(1);
    return (unsigned)__lane_id() == (unsigned)__ffsll((long long)m) - 1u;
}

// The traversal stack is addressed as LDS (address space 3) throughout, so that every access is a ds_read / ds_write
// and never a flat access through a generic pointer.
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// How a kernel meets instanced meshes (template parameter MODE of the traversal functions):
//   MODE_FLAT       the scene has no instances: one BVH.
//   MODE_TWO_LEVEL  instance leaves in the top-level BVH; a lane that reaches one changes its ray's space and walks the
//                   mesh's own BVH (reference src/geometry/transform.rs:22-31,99-130 as it stands).
//   MODE_UNROLLED   every instance's copy of its mesh's BVH is part of ONE tree whose boxes are all in world space; the
//                   change of space happens at the primitive test only: a leaf's reference carries the instance
//                   (slot << S.unroll_shift | first record, slot 0 = not instanced), the lane transforms its ray with that
//                   instance's record (S.insts[slot - 1]; S.insts[-1] is an identity record) and runs the reference's
//                   test on the mesh's object-space triangle, sphere or rectangle record.  Same primitive tests on the same operands as
//                   the two-level walk, so the same hits; no markers, no second grid, no parked rays.
static constexpr int MODE_FLAT = 0, MODE_TWO_LEVEL = 1, MODE_UNROLLED = 2;

// Resumable traversal state of one lane.  The render kernel interleaves traversal with shading of the lanes
// that have finished, so a ray's walk can be suspended between any two steps.
struct Trav {
    int cur;          // node / leaf reference being visited; REF_DONE when the walk is complete
    lds_int* top;     // LDS address of this lane's next free stack entry (entries are `stride` ints apart)
    Best best;
    V3 ga, gb;        // slab test constants of the ray on the current BVH's grid: plane q lies at t = q * ga + gb
    int32_t cur_inst; // instance record being traversed, -1 at top level
};

// Instanced scenes: a lane's ray registers (o, d) always hold the ray in the space of the BVH it is walking.  Entering
// an instance parks the world-space ray in LDS (six floats per lane, `stride` floats apart like the stack entries, so a
// wavefront's access touches consecutive banks) and overwrites the registers with the object-space ray; the exit marker
// brings the world-space ray back.  Nothing of the world ray is needed while a lane is inside a mesh, and six
// registers per lane are what separates the instanced kernels from the occupancy of the flat ones.  A walk that is
// complete has popped every exit marker: its registers hold the world ray again.
typedef __attribute__((address_space(3))) float lds_float;
typedef float vec4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) vec4f lds_float4;
__device__ __forceinline__ void wray_park(lds_float* w, int stride, V3 o, V3 d) {
    w[0] = o.x; w[stride] = o.y; w[2 * stride] = o.z; w[3 * stride] = d.x; w[4 * stride] = d.y; w[5 * stride] = d.z;
}
__device__ __forceinline__ void wray_restore(const lds_float* w, int stride, V3& o, V3& d) {
    o = v3(w[0], w[stride], w[2 * stride]);
    d = v3(w[3 * stride], w[4 * stride], w[5 * stride]);
}

// Stack access.  The stack pointer `top` is the LDS address of the lane's next free entry, so that a push or pop
// needs no address arithmetic beyond moving it by one entry.  K == 0: all STACK_DEPTH entries of a lane live in LDS.
// K > 0: LDS holds entries 0 .. K-1, which is what bounds the workgroups per CU, and the entries above them live in
// the lane's `deep` array (private memory); `lim` is the address where entry K of the workgroup's first lane would
// lie (uniform): a lane's entry e lies in LDS exactly if its address is below lim, because a lane's offset inside an
// entry row is smaller than the row.  Walks rarely go that deep (K = 16 against typical depths of 10-14), so the
// extra cost is a compare per push and per pop.
__device__ __forceinline__ uint32_t lds_addr(const lds_int* p) { return (uint32_t)(size_t)p; }
template <int K>
__device__ __forceinline__ void stack_push(lds_int*& top, int stride, lds_int* lim, int* deep, int v) {
    if (K > 0 && top >= lim) deep[(lds_addr(top) - lds_addr(lim)) / (uint32_t)(stride * 4)] = v;
    else *top = v;
    top += stride;
}
template <int K>
__device__ __forceinline__ int stack_pop(lds_int*& top, int stride, lds_int* lim, int* deep) {
    top -= stride;
    if (K == 0) return *top;
    // the LDS read is unconditional (some entry of row K-1 for a deep index): written as an either-or, the two loads
    // are merged into one flat load through a selected generic address
    int v = *(top < lim ? top : lim - stride);
    if (top >= lim) v = deep[(lds_addr(top) - lds_addr(lim)) / (uint32_t)(stride * 4)];
    return v;
}

__device__ __forceinline__ void trav_set_ray(Trav& T, V3 o, V3 d, const float* grid_origin, const float* grid_step) {
    const V3 inv = slab_reciprocal(d);
    T.ga = v3(grid_step[0] * inv.x, grid_step[1] * inv.y, grid_step[2] * inv.z);
    T.gb = v3((grid_origin[0] - o.x) * inv.x, (grid_origin[1] - o.y) * inv.y, (grid_origin[2] - o.z) * inv.z);
}

// (entry plane distance, exit plane distance) of one axis of a child box: `w` holds (min index | max index << 16)
__device__ __forceinline__ void slab_pair(uint32_t w, uint32_t rot, float a, float b, float& t_in, float& t_out) {
    const uint32_t v = __builtin_amdgcn_alignbit(w, w, rot);
    t_in = __builtin_fmaf((float)(v & 0xFFFFu), a, b);
    t_out = __builtin_fmaf((float)(v >> 16), a, b);
}

// Scaled form of the slab arithmetic (SCALED; the render kernels on scenes walked on ONE grid).  A 16-bit plane index
// q, read as the bit pattern of a float, IS the float q * 2^-149 (a denormal), so extracting a half of a node's dword
// (one AND or one shift, full-rate instructions) replaces the integer-to-float convert (a half-rate instruction, twelve
// of them per node step).  The multiplier absorbs the 2^149, which only fits the float range if plane distances are
// formed on a smaller scale: the node step works on tau = t * sigma, sigma a power of two chosen per grid so that the
// largest of step * sigma * 2^149 lies in [2^48, 2^49).  Scaling by a power of two is exact, so
// fma(asfloat(q), A, B) with A = step * sigma * 2^149 * inv and B = (origin - o) * inv * sigma is the unscaled
// fma(float(q), step * inv, (origin - o) * inv) times sigma bit for bit (tools/valu_class.hip checks all 65536
// indices) as long as nothing leaves the normal range: inv is clamped to +-2^78 instead of +-1e30 (A stays below
// 2^127), which is a box test for a ray whose direction component is at least 2^-78 in magnitude: for such a
// near-parallel ray the slab's two distances stay on either side of every t the ray can reach inside the scene, as
// with the larger clamp.  Entry distances are clamped to [0, 1] by the fma itself (tau = 1 is 2^100 grid steps away),
// which replaces the max with t_min by the weaker, still conservative t >= 0.
// Valid while tau stays clear of the ends of the float range: for rays whose direction is between 2^-40 and 2^40 long,
// i.e. the render kernels' rays (the host checks the camera's, scattered directions have length 1 or at most 2) and
// not the ray batches of rrt_intersect, whose kernels keep the convert form.  Either form only has to be conservative:
// hits are decided by the primitive tests, so images do not depend on which one ran.
static constexpr float SLAB_INV_MAX_SCALED = 0x1p78f;
#ifndef RRT_NESTED_STEPS
#define RRT_NESTED_STEPS 1  // 1: the steps of a group as nested conditions (one restore of the execution mask per group); 0: one condition per step
#endif
#ifndef RRT_SLAB_BFE
#define RRT_SLAB_BFE 1  // 1: plane indices taken out of a node's dwords by bit-field extracts; 0: by a rotate, a mask and a shift
#endif
__device__ __forceinline__ float slab_rcp_scaled(float d) {
    return fmaxf(fminf(__builtin_amdgcn_rcpf(d), SLAB_INV_MAX_SCALED), -SLAB_INV_MAX_SCALED);
}
__device__ __forceinline__ void trav_set_ray_scaled(Trav& T, V3 o, V3 d, const SceneDev& S) {
    const V3 inv = v3(slab_rcp_scaled(d.x), slab_rcp_scaled(d.y), slab_rcp_scaled(d.z));
    T.ga = v3(S.step_scaled[0] * inv.x, S.step_scaled[1] * inv.y, S.step_scaled[2] * inv.z);
    // (scaled before the multiplication by inv: a distant origin times a clamped inv must not overflow)
    T.gb = v3(((S.grid.origin[0] - o.x) * S.sigma) * inv.x, ((S.grid.origin[1] - o.y) * S.sigma) * inv.y,
              ((S.grid.origin[2] - o.z) * S.sigma) * inv.z);
}
__device__ __forceinline__ void slab_pair_scaled(uint32_t w, uint32_t rot, float a, float b, float& t_in, float& t_out) {
#if RRT_SLAB_BFE
    // the half the ray enters through and the other one, each by one bit-field extract at a per-lane offset (two
    // instructions per axis instead of a rotate, a mask and a shift)
    t_in = __builtin_amdgcn_fmed3f(__builtin_fmaf(__uint_as_float(__builtin_amdgcn_ubfe(w, rot, 16u)), a, b), 0.0f, 1.0f);
    t_out = __builtin_fmaf(__uint_as_float(__builtin_amdgcn_ubfe(w, rot ^ 16u, 16u)), a, b);
#else
    const uint32_t v = __builtin_amdgcn_alignbit(w, w, rot);
    t_in = __builtin_amdgcn_fmed3f(__builtin_fmaf(__uint_as_float(v & 0xFFFFu), a, b), 0.0f, 1.0f);
    t_out = __builtin_fmaf(__uint_as_float(v >> 16), a, b);
#endif
}

template <int MODE, bool SCALED = false>
__device__ __forceinline__ void trav_begin(const SceneDev& S, V3 o, V3 d, float tmax, lds_int* stack, int stride, Trav& T) {
    T.best.t = tmax; T.best.u = 0.f; T.best.v = 0.f; T.best.what = WHAT_NONE; T.best.inst = -1;
    if (SCALED) trav_set_ray_scaled(T, o, d, S);
    else trav_set_ray(T, o, d, S.grid.origin, S.grid.step);
    T.cur_inst = -1;
    stack[0] = REF_DONE;
    T.top = stack + stride;
    T.cur = S.root;
}

// Instanced scenes: the reference is one a lane changes space on (an instance, the marker that ends one) or skips (an
// empty child): kinds 3..6.  Such a lane is neither descending nor at a leaf; the node pass resolves it.
__device__ __forceinline__ bool trav_due(int cur) { return (uint32_t)cur - (REF_LEAF | (KIND_INST << 28)) < (4u << 28); }

// Node pass: walks down while the current reference is an inner node (near child first, far child pushed).
// Lanes that reach a leaf wait outside the loop; once fewer than `min_lanes` lanes are still descending the
// pass ends for everybody, so that the waiting lanes get their primitive test instead of watching a handful
// of lanes run node steps (min_lanes == 0: every lane descends until it is at a leaf).
// In instanced scenes a lane that comes upon an instance (or upon the marker that ends one) changes its ray's space
// right here and keeps descending: as a leaf step of its own the change would cost the wavefront a whole pass for
// the few lanes that need it.  `o`, `d`: the lane's ray in the space it is walking (only touched when INST: replaced
// by the object-space ray on entering an instance and by the parked world-space ray on leaving it); `wray`: the lane's
// slot of the parked world rays (INST only).
template <int MODE, bool COUNT, int K = 0, int STEPS = 1, bool SCALED = false>
__device__ __forceinline__ void trav_nodes(const SceneDev& S, V3& o, V3& d, float tmin, int stride, Trav& T,
                                           TraceCounters& cnt, uint32_t min_lanes, lds_int* lim = nullptr, int* deep = nullptr,
                                           lds_float* wray = nullptr) {
    constexpr bool INST = MODE == MODE_TWO_LEVEL;
    static_assert(!(SCALED && INST), "the scaled slab arithmetic is for scenes walked on one grid");
    V3 ga = T.ga, gb = T.gb;
    const float bt_scaled = T.best.t * S.sigma;  // the closest hit so far on the node step's scale (a pass never changes it)
    // rotate count of a node's plane pair: 16 where the ray runs against the axis (the sign of ga, the grid step
    // being positive), else 0
    uint32_t rx = (__float_as_uint(ga.x) >> 31) << 4, ry = (__float_as_uint(ga.y) >> 31) << 4, rz = (__float_as_uint(ga.z) >> 31) << 4;
    int cur = T.cur;
    lds_int* top = T.top;
    // The loop is uniform over the wavefront: one compare per trip decides both which lanes take the step and whether
    // the pass goes on.
    bool desc = cur >= 0;
    const uint32_t min_go = min_lanes > 1u ? min_lanes : 1u;  // the pass goes on while at least this many lanes descend
    // One node step of the lanes it is called for.  `in_window` (a compile-time flag): the caller has established that
    // every stepping lane's stack top is, and stays for this step, inside the LDS part.
    auto step = [&](auto in_window) {
        constexpr bool WINDOW = decltype(in_window)::value;
        // The entry a pop would take is read before the node is fetched: its address does not depend on the step's
        // outcome, and its LDS round trip then runs beside the node's fetch instead of between this step's last
        // compare and the next step's fetch (a wavefront's steps are one dependent chain; at seven wavefronts per SIMD
        // the chain's length still shows in the kernel's time).
        int popped = 0;
        if (K == 0 || WINDOW) popped = *(top - stride);
        // an inner reference is the node's 32-bit byte offset from the uniform base, which the load takes from scalar registers
        const uint4* np = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(S.nodes) + (uint32_t)cur);
        const uint4 a = np[0], b = np[1];  // a = (left, right, L.x, L.y)  b = (L.z, R.x, R.y, R.z)
        if (COUNT) {
            cnt.nodes++;
            if (first_active_lane()) cnt.wave_nodes++;
        }
        float ix, iy, iz, ox, oy, oz;
        float ln, lf, rn, rf;
        if (SCALED) {
            slab_pair_scaled(a.z, rx, ga.x, gb.x, ix, ox);
            slab_pair_scaled(a.w, ry, ga.y, gb.y, iy, oy);
            slab_pair_scaled(b.x, rz, ga.z, gb.z, iz, oz);
            ln = fmaxf(fmaxf(ix, iy), iz);
            lf = fminf(fminf(fminf(ox, oy), oz), bt_scaled);
            slab_pair_scaled(b.y, rx, ga.x, gb.x, ix, ox);
            slab_pair_scaled(b.z, ry, ga.y, gb.y, iy, oy);
            slab_pair_scaled(b.w, rz, ga.z, gb.z, iz, oz);
            rn = fmaxf(fmaxf(ix, iy), iz);
            rf = fminf(fminf(fminf(ox, oy), oz), bt_scaled);
        } else {
            // left box
            slab_pair(a.z, rx, ga.x, gb.x, ix, ox);
            slab_pair(a.w, ry, ga.y, gb.y, iy, oy);
            slab_pair(b.x, rz, ga.z, gb.z, iz, oz);
            ln = fmaxf(fmaxf(fmaxf(ix, iy), iz), tmin);
            lf = fminf(fminf(fminf(ox, oy), oz), T.best.t);
            // right box
            slab_pair(b.y, rx, ga.x, gb.x, ix, ox);
            slab_pair(b.z, ry, ga.y, gb.y, iy, oy);
            slab_pair(b.w, rz, ga.z, gb.z, iz, oz);
            rn = fmaxf(fmaxf(fmaxf(ix, iy), iz), tmin);
            rf = fminf(fminf(fminf(ox, oy), oz), T.best.t);
        }
        int left = (int)a.x, right = (int)a.y;
        if (K == 0 || WINDOW) {
            // Every lane's stack top is inside the LDS part: the step is finished without a branch.  Scalar and
            // branch instructions take issue slots like vector ones, and the four-way case analysis below costs about
            // 25 of them per step.  Here the far child (or, with fewer than two hits, a value nobody will read) is
            // written above the top, the top moves up if both children are hit and down if neither is, and the entry
            // read back is the popped one exactly when neither is hit.  Entries touched: 0 .. K-1 (K == 0: the whole
            // stack is in LDS and has one entry more than the deepest walk needs).
            // The hit flags are combined as wavefront masks (scalar instructions): written on per-lane booleans, the
            // negation of a compare is compiled as a second vector compare, and compares are half-rate instructions.
            const unsigned long long hl_m = __builtin_amdgcn_ballot_w64(ln <= lf), hr_m = __builtin_amdgcn_ballot_w64(rn <= rf);
            const unsigned long long closer_m = __builtin_amdgcn_ballot_w64(rn < ln);
            // the right child first if it is hit and the left one is not, or is entered later
            const bool right_first = __builtin_amdgcn_inverse_ballot_w64(hr_m & ~(hl_m & ~closer_m));
            const bool any = __builtin_amdgcn_inverse_ballot_w64(hl_m | hr_m), both = __builtin_amdgcn_inverse_ballot_w64(hl_m & hr_m);
            const int near = right_first ? right : left;
            const int far = right_first ? left : right;
            *top = far;
            lds_int* down = any ? top : top - stride;
            top = both ? top + stride : down;
            cur = any ? near : popped;
        } else if (ln <= lf && rn <= rf) {
            if (rn < ln) { int tmp = left; left = right; right = tmp; }
            stack_push<K>(top, stride, lim, deep, right);
            cur = left;
        } else if (ln <= lf) {
            cur = left;
        } else if (rn <= rf) {
            cur = right;
        } else {
            cur = stack_pop<K>(top, stride, lim, deep);
        }
    };
    // Instanced scenes: a lane whose step ended on an instance reference (or on the marker that ends one, or on an
    // empty child) changes space here, once per group of steps: written into the step itself, the transition is
    // repeated in the kernel text for every step of the group and its registers are live across all of them.
    auto transition = [&]() {
        if (trav_due(cur)) {
            const uint32_t kind = ((uint32_t)cur >> 28) & 7u;
            if (kind == KIND_INST) {
                const uint32_t first = (uint32_t)cur & REF_OFFSET_MASK;
                const InstRec in = S.insts[first];
                wray_park(wray, stride, o, d);
                to_object_space(in, o, d, o, d);
                trav_set_ray(T, o, d, in.grid_origin, in.grid_step);
                T.cur_inst = (int32_t)first;
                stack_push<K>(top, stride, lim, deep, REF_INST_EXIT);
                cur = in.root;
            } else {
                if (kind == KIND_INST_EXIT) {
                    wray_restore(wray, stride, o, d);
                    trav_set_ray(T, o, d, S.grid.origin, S.grid.step);
                    T.cur_inst = -1;
                }
                cur = stack_pop<K>(top, stride, lim, deep);
            }
            ga = T.ga; gb = T.gb;
            rx = (__float_as_uint(ga.x) >> 31) << 4; ry = (__float_as_uint(ga.y) >> 31) << 4; rz = (__float_as_uint(ga.z) >> 31) << 4;
        }
    };
    // STEPS steps between two votes on whether the pass goes on (loop control is a tenth of a step: measured on the
    // 1M-triangle scene, 1 / 2 / 3 / 4 steps per vote run in 175.6 / 172.1 / 170.3 / 170.9 ms).  Whether the steps of a
    // group may use the branch-free ending is decided once per group, not once per step: a step raises a lane's top
    // by at most one entry, so if every descending lane's top lies at least STEPS - 1 entries below the end of the LDS
    // part when the group starts, it is inside the part for all of the group's steps.
    unsigned long long desc_m = __builtin_amdgcn_ballot_w64(desc);
    for (;;) {
      if (INST) {
          // lanes due for a transition take it first and descend (or wait at a leaf) in their new space
          transition();
          desc = cur >= 0;
          desc_m = __builtin_amdgcn_ballot_w64(desc);
      }
      const bool window = K == 0 || (__builtin_amdgcn_ballot_w64(top + (STEPS - 1) * stride >= lim) & desc_m) == 0ull;
      if (__builtin_expect(window, 1)) {
#if RRT_NESTED_STEPS
          // the lanes of a group's later steps are a subset of those of its earlier ones: written as nested
          // conditions, the group restores the execution mask once at its end instead of after every step
          auto nest = [&](auto self, auto n) -> void {
              constexpr int N = decltype(n)::value;
              if constexpr (N > 0) {
                  if (cur >= 0) {
                      step(std::true_type{});
                      self(self, std::integral_constant<int, N - 1>{});
                  }
              }
          };
          if (desc) {
              step(std::true_type{});
              nest(nest, std::integral_constant<int, STEPS - 1>{});
          }
#else
          if (desc) step(std::true_type{});
#pragma unroll
          for (int extra = 1; extra < STEPS; ++extra) {
              if (cur >= 0) step(std::true_type{});
          }
#endif
      } else {
          if (desc) step(std::false_type{});
#pragma unroll
          for (int extra = 1; extra < STEPS; ++extra) {
              if (cur >= 0) step(std::false_type{});
          }
      }
      // the pass goes on for the lanes that descend and for those due for a transition (which descend after it)
      desc = cur >= 0;
      if (INST) {
          if ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(desc || trav_due(cur))) < min_go) break;
      } else {
          desc_m = __builtin_amdgcn_ballot_w64(desc);
          if ((uint32_t)__popcll(desc_m) < min_go) break;
      }
    }
    T.cur = cur;
    T.top = top;
}

// One leaf step (or stack marker): tests ONE primitive of the leaf with the reference's arithmetic; a triangle
// leaf with more triangles stays current (reference advanced by one record), otherwise the next entry is popped.
// One primitive per step keeps the cost of a pass the same for every lane, whatever its leaf holds.
// Precondition: T.cur is negative and not REF_DONE.
// PAIR: a triangle leaf with at least two records left has two of them tested in one step.
// xf_lds (MODE_UNROLLED): the workgroup's LDS copy of the instance transforms, two float4 per slot (slot 0: identity), or
// nullptr when the table is read from device memory.
template <int MODE, bool COUNT, int K = 0, bool PAIR = false>
__device__ __forceinline__ void trav_leaf(const SceneDev& S, V3& o, V3& d, float tmin, int stride, Trav& T,
                                          TraceCounters& cnt, lds_int* lim = nullptr, int* deep = nullptr,
                                          lds_float* wray = nullptr, const lds_float4* xf_lds = nullptr) {
    constexpr bool INST = MODE == MODE_TWO_LEVEL;
    const uint32_t ref = (uint32_t)T.cur;
    const uint32_t kind = (ref >> 28) & 7u;
    uint32_t first = ref & REF_OFFSET_MASK;
    V3 ro = o, rd = d;  // the ray in the space being walked
    int32_t tri_inst = T.cur_inst;  // instance the primitives of this leaf belong to
    uint32_t first_mask = REF_OFFSET_MASK;
    if (MODE == MODE_UNROLLED) {
        // a leaf names its instance: the ray goes into that instance's space for the test, exactly as
        // Translate::hit and RotateY::hit move it (transform.rs:24,101-108); slot 0 (inst -1) reads the identity
        // record in front of the array and keeps the ray as it is
        first_mask = (1u << S.unroll_shift) - 1u;
        tri_inst = (int32_t)(first >> S.unroll_shift) - 1;
        first &= first_mask;
        // (the record comes from the workgroup's LDS copy of the table when the kernel keeps one: a load through the
        // vector L1 takes its address unit 16 cycles whatever the number of lanes, and that unit is as busy as the SIMDs)
        float4 x0, x1;
        if (xf_lds != nullptr) {
            const lds_float4* e = xf_lds + 2 * (tri_inst + 1);
            const vec4f e0 = e[0], e1 = e[1];
            x0 = make_float4(e0.x, e0.y, e0.z, e0.w);
            x1 = make_float4(e1.x, e1.y, e1.z, e1.w);
        } else {
            x0 = reinterpret_cast<const float4*>(&S.insts[tri_inst])[0];
            x1 = reinterpret_cast<const float4*>(&S.insts[tri_inst])[1];
        }
        const V3 m = o - v3(x0.x, x0.y, x0.z);
        const V3 to = v3(x1.y * m.x - x1.x * m.z, m.y, x1.x * m.x + x1.y * m.z);
        const V3 td = v3(x1.y * d.x - x1.x * d.z, d.y, x1.x * d.x + x1.y * d.z);
        const bool in_inst = tri_inst >= 0;
        ro = v3(in_inst ? to.x : o.x, in_inst ? to.y : o.y, in_inst ? to.z : o.z);
        rd = v3(in_inst ? td.x : d.x, in_inst ? td.y : d.y, in_inst ? td.z : d.z);
    }
    // The common case, decided once for the wavefront: every lane of the pass is at a triangle and the entry below
    // its stack top is in LDS.  The step then ends without a branch: the entry below the top is read in any case and
    // becomes current, and the top moves down, only if the leaf has no triangle left.
    if (__builtin_expect(__ballot(kind != KIND_TRI || (K > 0 && T.top - stride >= lim)) == 0ull, 1)) {
        test_triangle(S, first, ro, rd, tmin, tri_inst, T.best);
        if (COUNT) {
            cnt.prims++;
            if (first_active_lane()) cnt.wave_prims++;
        }
        // A second triangle of the same leaf in the same pass: the builder's leaves mostly hold the two triangles of
        // a quad, and the pass's fixed cost (dispatch, pop, the loop around it) is then paid once for both.  (On the
        // large scenes 2-3 % faster; on the 968-triangle scene, whose passes hold a handful of lanes, 17 % slower,
        // hence the switch.)
        uint32_t r = ref;
        bool more = ((r >> 25) & 7u) != 0u;
        if (PAIR && more) {
            r += 1u - (1u << 25);  // next record, one triangle fewer
            test_triangle(S, r & first_mask, ro, rd, tmin, tri_inst, T.best);
            if (COUNT) {
                cnt.prims++;
                if (first_active_lane()) cnt.wave_prims++;
            }
            more = ((r >> 25) & 7u) != 0u;
        }
        lds_int* down = T.top - stride;
        const int popped = *down;
        T.cur = more ? (int)(r + 1u - (1u << 25)) : popped;  // the leaf's next record, or the popped entry
        T.top = more ? T.top : down;
        return;
    }
    if (kind == KIND_TRI) {
        test_triangle(S, first, ro, rd, tmin, tri_inst, T.best);
        if (COUNT) {
            cnt.prims++;
            if (first_active_lane()) cnt.wave_prims++;
        }
        if (((ref >> 25) & 7u) != 0u) {
            T.cur = (int)(ref + 1u - (1u << 25));  // next record, one triangle fewer
            return;
        }
    } else if (kind == KIND_SPHERE) {
        test_sphere(S, first, ro, rd, tmin, tri_inst, T.best);
        if (COUNT) cnt.prims++;
    } else if (kind == KIND_RECT) {
        test_rect(S, first, ro, rd, tmin, tri_inst, T.best);
        if (COUNT) cnt.prims++;
    } else if (INST && kind == KIND_INST) {
        const InstRec in = S.insts[first];
        wray_park(wray, stride, o, d);
        to_object_space(in, o, d, o, d);
        trav_set_ray(T, o, d, in.grid_origin, in.grid_step);
        T.cur_inst = (int32_t)first;
        stack_push<K>(T.top, stride, lim, deep, REF_INST_EXIT);
        T.cur = in.root;
        return;
    } else if (INST && kind == KIND_INST_EXIT) {
        wray_restore(wray, stride, o, d);
        trav_set_ray(T, o, d, S.grid.origin, S.grid.step);
        T.cur_inst = -1;
    }
    T.cur = stack_pop<K>(T.top, stride, lim, deep);
}

// Complete walk of one ray (used by the ray-batch kernel): the same passes as the render kernel, run to the end.
template <int MODE, bool COUNT>
__device__ __forceinline__ void trace(const SceneDev& S, V3 o, V3 d, float tmin, float tmax, lds_int* stack, int stride,
                                      Best& best, TraceCounters& cnt, lds_float* wray = nullptr) {
    Trav T;
    trav_begin<MODE>(S, o, d, tmax, stack, stride, T);
    while (T.cur != REF_DONE) {
        trav_nodes<MODE, COUNT>(S, o, d, tmin, stride, T, cnt, 0u, nullptr, nullptr, wray);  // 0: descend until this lane is at a leaf
        if (T.cur != REF_DONE) trav_leaf<MODE, COUNT>(S, o, d, tmin, stride, T, cnt, nullptr, nullptr, wray);
    }
    best = T.best;
}

// ---------------------------------------------------------------------------------------------
// Surface record of the closest hit -- what the reference's HitRecord carries
// (src/material/mod.rs:18-25): p, normal, material.
// ---------------------------------------------------------------------------------------------
struct Surface {
    V3 p, n;
    uint32_t material;
};

__device__ __forceinline__ Surface surface_at(const SceneDev& S, const Best& best, V3 o, V3 d) {
    Surface s;
    V3 ro = o, rd = d;
    InstRec in;
    const bool instanced = best.inst >= 0;
    if (instanced) {
        in = S.insts[best.inst];
        to_object_space(in, o, d, ro, rd);
    }
    const uint32_t kind = best.what >> 28, idx = best.what & REF_OFFSET_MASK;
    s.p = ro + best.t * rd;  // Ray::at, src/ray.rs:41-43
    if (kind == KIND_TRI) {
        const float4 q1 = *reinterpret_cast<const float4*>(&S.tris[idx].e1[0]);
        const float4 q2 = *reinterpret_cast<const float4*>(&S.tris[idx].e2[0]);
        // det > 1e-5 on every accepted hit, so det.signum() == 1 (triangle.rs:42,58)
        s.n = unit(cross(v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z)));
        s.material = __float_as_uint(q1.w);
    } else if (kind == KIND_SPHERE) {
        const float4 q0 = *reinterpret_cast<const float4*>(&S.spheres[idx].c[0]);
        s.n = (s.p - v3(q0.x, q0.y, q0.z)) / q0.w;  // sphere.rs:65,86
        s.material = S.spheres[idx].material;
    } else {
        const float4 q1 = *reinterpret_cast<const float4*>(&S.rects[idx].k);
        const uint32_t af = __float_as_uint(q1.y);
        const uint32_t axis = af & 3u;
        const float sg = signum(comp(ro, axis) - q1.x);  // aarect.rs:49,104,180
        s.n = v3(axis == 0 ? sg : 0.0f, axis == 1 ? sg : 0.0f, axis == 2 ? sg : 0.0f);
        if (af & 0x100u) s.n.y = -fabsf(s.n.y);  // FlipFace, geometry/mod.rs:226-230
        s.material = __float_as_uint(q1.w);
    }
    if (instanced) {
        // RotateY::hit, transform.rs:113-124 (the sign test mixes the rotated-back normal with the
        // object-space direction, exactly as the reference does), then Translate::hit, transform.rs:27
        const V3 p = v3(in.cos_t * s.p.x + in.sin_t * s.p.z, s.p.y, -in.sin_t * s.p.x + in.cos_t * s.p.z);
        const V3 n = v3(in.cos_t * s.n.x + in.sin_t * s.n.z, s.n.y, -in.sin_t * s.n.x + in.cos_t * s.n.z);
        s.n = (-n) * signum(dot(n, rd));
        s.p = p + v3(in.offset[0], in.offset[1], in.offset[2]);
    }
    return s;
}

// ---------------------------------------------------------------------------------------------
// Camera -- reference src/camera.rs:56-69.  (x, y) already contain the pixel jitter; the second
// half-pixel offset below is the reference's.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void camera_ray(const CameraRec& cam, float x, float y, uint32_t w, uint32_t h, V3& o, V3& d) {
    const float px = (2.0f * (x + 0.5f) / (float)w - 1.0f) * cam.scale * cam.aspect;
    const float py = (2.0f * (y + 0.5f) / (float)h - 1.0f) * cam.scale;
    const V3 c0 = v3(cam.c0[0], cam.c0[1], cam.c0[2]), c1 = v3(cam.c1[0], cam.c1[1], cam.c1[2]);
    const V3 c2 = v3(cam.c2[0], cam.c2[1], cam.c2[2]), tr = v3(cam.tr[0], cam.tr[1], cam.tr[2]);
    const V3 ray_p = ((c0 * px + c1 * py) + c2 * (-1.0f)) + tr;  // Affine3A::transform_point3a
    o = v3(cam.origin[0], cam.origin[1], cam.origin[2]);
    d = ray_p - o;
}

// ---------------------------------------------------------------------------------------------
// Candidate leaves of a pixel.  Every camera ray of one pixel (src/main.rs:92-94: the pixel's corner plus a jitter in
// [0, 1)^2, through src/camera.rs:56-69, which is affine in the jittered coordinates) lies inside the pyramid spanned by
// the rays through the pixel's four corners.  A leaf whose box that pyramid misses cannot hold a primitive any of those
// rays hits (a hit point lies inside its primitive's box, which the leaf's stored box contains with the builder's
// margin to spare), so the closest hit of a primary ray of the pixel is the minimum over the primitive tests of the
// leaves the pyramid touches -- the same minimum the ray's own walk forms, found without a walk.  The render kernel
// collects them in a pass over the frame's pixels (k_pixel_lists) and resolves the pixel's primary rays against
// the list (test_listed_leaf).  Scenes without instances, and scenes whose instances are unrolled into one world-space
// tree (the boxes are world boxes there too; a triangle of an instance is moved to world space for the pyramid's tests).
// ---------------------------------------------------------------------------------------------
struct PixelFrustum {
    V3 o;
    V3 n[4];        // inward normals of the four side planes through o
    V3 d[4];        // the rays through the four corners
    V3 a;           // the ray through the pixel's centre
    float inv_amax; // 1 / the largest a . d over the pyramid's rays d (attained at a corner: a . d is affine over the pixel)
    float d_l1;     // the largest 1-norm of a corner ray
    bool ok;        // false: the camera arithmetic is too coarse for a pyramid of this pixel to mean anything (no list)
};
// err_scale: 1 for rays that are tested as the camera forms them; 3 where a ray is also moved into an instance's space
// before a test (Translate::hit, RotateY::hit: two more roundings of its origin and direction at the same magnitudes).
__device__ __forceinline__ PixelFrustum pixel_frustum(const CameraRec& cam, uint32_t px, uint32_t py, uint32_t w, uint32_t h, float err_scale = 1.0f) {
    // The jittered coordinates of the pixel lie in [px, px + 1] x [py, py + 1].  Slack on every side: 1/64 of a pixel,
    // plus what the rounding of camera_ray can move a ray (and the corner rays formed here) across the image plane:
    // ray_p is rounded at the magnitude of the camera's position, however short the camera's axes are, i.e. each of
    // its components is off by up to an ulp of |translation| + |axes|.  That error vector, written in the basis of the
    // axes, is a shift of the ray's image-plane coordinates (its part along the view axis rescales them).
    const V3 q0 = v3(cam.c0[0], cam.c0[1], cam.c0[2]), q1 = v3(cam.c1[0], cam.c1[1], cam.c1[2]), q2 = v3(cam.c2[0], cam.c2[1], cam.c2[2]);
    const float ax = fabsf(cam.scale * cam.aspect), ay = fabsf(cam.scale);
    const float err = err_scale * 2.5e-7f * fmaxf(fmaxf(fabsf(cam.tr[0]) + fabsf(q0.x) * ax + fabsf(q1.x) * ay + fabsf(q2.x),
                                            fabsf(cam.tr[1]) + fabsf(q0.y) * ax + fabsf(q1.y) * ay + fabsf(q2.y)),
                                      fabsf(cam.tr[2]) + fabsf(q0.z) * ax + fabsf(q1.z) * ay + fabsf(q2.z));
    const V3 r0 = cross(q1, q2), r1 = cross(q2, q0), r2 = cross(q0, q1);  // rows of the inverse of (q0 q1 q2), times its determinant
    const float inv_det = 1.0f / fabsf(dot(q0, r0));
    const float ex = 1.0f / 64.0f + err * inv_det * (norm1(r0) + ax * norm1(r2)) * (float)w / (2.0f * ax);
    const float ey = 1.0f / 64.0f + err * inv_det * (norm1(r1) + ay * norm1(r2)) * (float)h / (2.0f * ay);
    const float x0 = (float)px - ex, x1 = (float)px + 1.0f + ex, y0 = (float)py - ey, y1 = (float)py + 1.0f + ey;
    PixelFrustum F;
    F.ok = ex < 2.0f && ey < 2.0f;  // (also false for NaN: a singular camera matrix)
    V3 d00, d10, d11, d01, o;
    camera_ray(cam, x0, y0, w, h, F.o, d00);
    camera_ray(cam, x1, y0, w, h, o, d10);
    camera_ray(cam, x1, y1, w, h, o, d11);
    camera_ray(cam, x0, y1, w, h, o, d01);
    camera_ray(cam, (float)px + 0.5f, (float)py + 0.5f, w, h, o, F.a);
    // (a side's normal from one corner direction and the DIFFERENCE to the next: the cross product of two nearly
    // parallel vectors would cancel its leading digits)
    F.n[0] = cross(d00, d10 - d00);
    F.n[1] = cross(d10, d11 - d10);
    F.n[2] = cross(d11, d01 - d11);
    F.n[3] = cross(d01, d00 - d01);
    for (int k = 0; k < 4; ++k)
        if (dot(F.n[k], F.a) < 0.0f) F.n[k] = -F.n[k];
    F.inv_amax = 1.0f / fmaxf(fmaxf(dot(F.a, d00), dot(F.a, d10)), fmaxf(dot(F.a, d11), dot(F.a, d01)));
    F.d[0] = d00; F.d[1] = d10; F.d[2] = d11; F.d[3] = d01;
    F.d_l1 = fmaxf(fmaxf(norm1(d00), norm1(d10)), fmaxf(norm1(d11), norm1(d01)));
    return F;
}
// False only if no ray of the pyramid can pass test_triangle for the triangle (v0, v0 + e1, v0 + e2):
//   * every ray sees its back: test_triangle wants det = e1 . (d x e2) = d . (e2 x e1) above 1e-5, det is linear in d
//     and every ray of the pixel is a convex combination of the corner rays, so det <= its largest corner value; or
//   * its three corners lie outside one side plane, by more than rho: the triangle is the convex hull of its corners,
//     so it then lies outside the pyramid by that much.  rho covers what test_triangle's own rounding can add to a
//     triangle's extent: its barycentric coordinates carry an error of a few ulps of |o - v0| / |e|, i.e. it accepts
//     rays that pass the triangle at a few ulps of the distance to it.
// Both with allowances of tens of ulps of the sums' terms (times `slack`: 2 for a triangle of an instance, whose corners
// the caller has moved to world space and whose test runs on the ray moved the other way).
__device__ __forceinline__ bool frustum_touches_triangle(const PixelFrustum& F, V3 v0, V3 e1, V3 e2, float slack = 1.0f) {
    const V3 m = cross(e2, e1);
    const float det_max = fmaxf(fmaxf(dot(F.d[0], m), dot(F.d[1], m)), fmaxf(dot(F.d[2], m), dot(F.d[3], m)));
    const float le1 = norm1(e1), le2 = norm1(e2);
    if (det_max < -4e-6f * slack * (le1 * le2 * F.d_l1)) return false;
    const V3 p0 = v0 - F.o, p1 = p0 + e1, p2 = p0 + e2;
    const float rho = 4e-6f * slack * ((norm1(p0) + le1) + le2);
    bool in = true;
    for (int k = 0; k < 4; ++k) {
        const V3 n = F.n[k];
        const float far_in = fmaxf(fmaxf(dot(n, p0), dot(n, p1)), dot(n, p2));
        in = in && (far_in >= -rho * norm1(n));
    }
    if (!in) return false;
    //   * or the pyramid lies outside one of the three planes through the apex and an edge of the triangle (the
    //     triangle lies on one side of each, the side of its third corner): all four corner rays on the other side,
    //     by more than the triangle's allowance rho seen from where the pyramid can first come near it (the ray
    //     parameter there is at least z_min / amax, with z_min the triangle's least depth along the centre ray).
    //     Together with the side planes these are all separating planes of two convex cones from one apex.
    const float z_min = fminf(fminf(dot(F.a, p0), dot(F.a, p1)), dot(F.a, p2)) - rho * norm1(F.a);
    if (!(z_min > 0.0f)) return true;  // the triangle reaches the apex's own depth: kept
    const V3 m01 = cross(p0, e1), m12 = cross(p1, e2 - e1), m20 = cross(p2, -e2);
    const float vol = dot(m01, p2);  // the same triple product for all three edges (cyclic order)
    if (!(fabsf(vol) > 1e-5f * (norm1(m01) * norm1(p2)))) return true;  // seen edge-on: kept
    const float sgn = vol > 0.0f ? 1.0f : -1.0f;
    const float per_norm = rho * __builtin_amdgcn_rcpf(z_min * F.inv_amax) * 1.0001f + 2e-6f * slack * F.d_l1;
    auto outside = [&](V3 m) {
        const float inmost = fmaxf(fmaxf(sgn * dot(m, F.d[0]), sgn * dot(m, F.d[1])), fmaxf(sgn * dot(m, F.d[2]), sgn * dot(m, F.d[3])));
        return inmost < -per_norm * norm1(m);
    };
    return !(outside(m01) || outside(m12) || outside(m20));
}
// False only if the box [mn, mx] lies outside one of the pyramid's side planes (conservative: the plane test is made
// at the box's corner furthest inside, with a relative allowance for the rounding of the sum).
__device__ __forceinline__ bool frustum_touches_box(const PixelFrustum& F, V3 mn, V3 mx) {
    bool in = true;
    for (int k = 0; k < 4; ++k) {
        const V3 n = F.n[k];
        const V3 p = v3((n.x >= 0.0f ? mx.x : mn.x) - F.o.x, (n.y >= 0.0f ? mx.y : mn.y) - F.o.y, (n.z >= 0.0f ? mx.z : mn.z) - F.o.z);
        const float a = n.x * p.x, b = n.y * p.y, c = n.z * p.z;
        in = in && ((a + b) + c >= -4e-6f * ((fabsf(a) + fabsf(b)) + fabsf(c)));
    }
    return in;
}
// A lower bound of the ray parameter at which any ray of the pyramid can reach the box: a point q = o + t d has
// a . (q - o) = t (a . d), so t >= min over the box of a . (q - o), divided by the largest a . d; a little below it, and
// never below zero.
__device__ __forceinline__ float frustum_box_entry(const PixelFrustum& F, V3 mn, V3 mx) {
    const V3 a = F.a;
    const V3 p = v3((a.x >= 0.0f ? mn.x : mx.x) - F.o.x, (a.y >= 0.0f ? mn.y : mx.y) - F.o.y, (a.z >= 0.0f ? mn.z : mx.z) - F.o.z);
    const float x = a.x * p.x, y = a.y * p.y, z = a.z * p.z;
    const float lo = ((x + y) + z) - 4e-6f * ((fabsf(x) + fabsf(y)) + fabsf(z));
    return fmaxf(lo * F.inv_amax, 0.0f) * 0.9999f;
}
// Inserts (ref, t) into the list refs[0 .. n), entry[0 .. n), which is ascending in entry (lists are a handful of
// entries long).
__device__ __forceinline__ void list_insert(uint32_t* refs, float* entry, uint32_t& n, uint32_t ref, float t) {
    uint32_t k = n;
    while (k > 0u && entry[k - 1u] > t) { refs[k] = refs[k - 1u]; entry[k] = entry[k - 1u]; --k; }
    refs[k] = ref; entry[k] = t;
    n++;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const u32x4 const_u32x4;  // constant address space: a uniform address is fetched through the scalar cache
// Tests the primitives of one listed leaf with the reference's arithmetic (what trav_leaf does for a walking lane).
// MODE_UNROLLED: the reference names its instance slot above the record index; the ray goes into that instance's
// space for the test as in trav_leaf (xf_lds: the workgroup's LDS copy of the transforms, or nullptr).
template <bool COUNT, int MODE = MODE_FLAT>
__device__ __forceinline__ void test_listed_leaf(const SceneDev& S, uint32_t ref, V3 o, V3 d, float tmin, Best& best, TraceCounters& cnt,
                                                 const lds_float4* xf_lds = nullptr) {
    const uint32_t kind = (ref >> 28) & 7u;
    uint32_t first = ref & REF_OFFSET_MASK;
    int32_t inst = -1;
    if (MODE == MODE_UNROLLED) {
        inst = (int32_t)(first >> S.unroll_shift) - 1;
        first &= (1u << S.unroll_shift) - 1u;
        if (inst >= 0) {
            float4 x0, x1;
            if (xf_lds != nullptr) {
                const lds_float4* e = xf_lds + 2 * (inst + 1);
                const vec4f e0 = e[0], e1 = e[1];
                x0 = make_float4(e0.x, e0.y, e0.z, e0.w);
                x1 = make_float4(e1.x, e1.y, e1.z, e1.w);
            } else {
                x0 = reinterpret_cast<const float4*>(&S.insts[inst])[0];
                x1 = reinterpret_cast<const float4*>(&S.insts[inst])[1];
            }
            const V3 m = o - v3(x0.x, x0.y, x0.z);  // Translate::hit, RotateY::hit (transform.rs:24,101-108), as trav_leaf forms them
            o = v3(x1.y * m.x - x1.x * m.z, m.y, x1.x * m.x + x1.y * m.z);
            d = v3(x1.y * d.x - x1.x * d.z, d.y, x1.x * d.x + x1.y * d.z);
        }
    }
    if (kind == KIND_TRI) {
        const uint32_t count = ((ref >> 25) & 7u) + 1u;
        for (uint32_t j = 0; j < count; ++j) {
            test_triangle(S, first + j, o, d, tmin, inst, best);
            if (COUNT) cnt.prims++;
        }
    } else if (kind == KIND_SPHERE) {
        test_sphere(S, first, o, d, tmin, inst, best);
        if (COUNT) cnt.prims++;
    } else {
        test_rect(S, first, o, d, tmin, inst, best);
        if (COUNT) cnt.prims++;
    }
}

// ---------------------------------------------------------------------------------------------
// Light list -- reference src/geometry/mod.rs:111-125 over src/geometry/sphere.rs:101-119,217-228
// and src/geometry/aarect.rs:123-142
// ---------------------------------------------------------------------------------------------
// (Hittables::random, geometry/mod.rs:118-125, is inlined into `shade`, where it shares code with the cosine lobe.)
__device__ __forceinline__ float lights_pdf(const SceneDev& S, V3 origin, V3 v) {
    const float weight = S.light_weight;  // 1 / light count, geometry/mod.rs:112
    float sum = 0.0f;
    for (uint32_t i = 0; i < S.light_count; ++i) {
        const LightRec L = S.lights[i];
        float pdf = 0.0f;
        if (L.type == 0u) {
            const V3 c = v3(L.p0, L.p1, L.p2);
            const float root = sphere_root(c, L.p3, origin, v, 0.0001f, __builtin_inff());
            if (root == root) {
                const float cos_theta_max = sqrt_rn(1.0f - L.p3 * L.p3 / length_squared(c - origin));
                const float solid_angle = 2.0f * RRT_PI * (1.0f - cos_theta_max);
                pdf = recip_rn(solid_angle);
            }
        } else if (L.type == 1u) {
            float t, uu, vv;
            if (rect_hit(L.p0, L.p1, L.p2, L.p3, L.p4, 1u, origin, v, 0.001f, __builtin_inff(), t, uu, vv)) {
                const float area = (L.p1 - L.p0) * (L.p3 - L.p2);
                const float dist2 = t * t * length_squared(v);
                const V3 n = v3(0.0f, signum(origin.y - L.p4), 0.0f);
                const float cosine = fabsf(dot(v, n)) / sqrt_rn(length_squared(v));
                pdf = dist2 / (cosine * area);
            }
        }
        sum += weight * pdf;
    }
    return sum;
}

// reference src/pdf.rs:14-26
__device__ __forceinline__ V3 random_cosine_direction(Rng& rng) {
    const float r1 = rng_f32(rng);
    const float r2 = rng_f32(rng);
    const float z = sqrt_rn(1.0f - r2);
    const float sr2 = sqrt_rn(r2);
    float sn, cs;
    sincos_turn(r1, sn, cs);  // phi = 2 pi r1
    return v3(cs * sr2, sn * sr2, z);
}

// reference src/vec3.rs:9-25
__device__ __forceinline__ V3 random_in_unit_sphere(Rng& rng) {
    for (;;) {
        const float x = rng_range(rng, -1.0f, 1.0f);
        const float y = rng_range(rng, -1.0f, 1.0f);
        const float z = rng_range(rng, -1.0f, 1.0f);
        const V3 p = v3(x, y, z);
        if (length_squared(p) < 1.0f) return p;
    }
}

__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2.0f * dot(v, n)) * n; }  // material/mod.rs:150-152

// x^5 the way llvm.powi / compiler-rt __powisf2 evaluates it: x * ((x*x)*(x*x))
__device__ __forceinline__ float powi5(float x) {
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return x * x4;
}

// One bounce of the reference's recursive `ray_color` (src/ray.rs:78-127), in throughput form:
//   radiance += beta * emitted;  beta *= weight;  ray = scattered;   (returns false when the path ends)
__device__ __forceinline__ bool shade(const SceneDev& S, const Surface& sf, V3& o, V3& d, V3& beta, V3& radiance,
                                      Rng& rng) {
    const float4 m0 = *reinterpret_cast<const float4*>(&S.mats[sf.material].kind);
    const uint32_t kind = __float_as_uint(m0.x);
    const V3 rgb = v3(m0.y, m0.z, m0.w);
    if (kind == 3u) {  // DiffuseLight, material/mod.rs:183-194: no scatter, emits towards the front face only
        if (dot(d, sf.n) < 0.0f) radiance = radiance + mulv(beta, rgb);
        return false;
    }
    if (kind == 0u) {  // Lambertian, material/mod.rs:71-84 + mixture sampling, ray.rs:99-120
        const Onb uvw = onb_from_w(sf.n);
        V3 dir, unit_dir;
        float pdf_val;
        if (S.light_count != 0u) {
            // MixturePDF::generate, pdf.rs:87-94.  The cosine lobe (pdf.rs:14-26,42) and a sphere light
            // (sphere.rs:111-119,217-228) both draw r1, r2, form (cos(phi) * radial, sin(phi) * radial, z) and rotate
            // it into a basis; the lanes of a wavefront that took different branches of the coin flip share that
            // code, each with its own basis, z and radial (same operations in the same order as the two functions).
            const bool use_cosine = rng_bool_half(rng);
            Onb basis = uvw;
            bool in_basis = true;
            float sphere_r2_over_d2 = 0.0f;
            if (!use_cosine) {
                const uint32_t pick = rng_index(rng, S.light_count, S.light_zone);
                const LightRec L = S.lights[pick];
                if (L.type == 0u) {
                    const V3 direction = v3(L.p0, L.p1, L.p2) - sf.p;
                    sphere_r2_over_d2 = L.p3 * L.p3 / length_squared(direction);
                    basis = onb_from_w(direction);
                } else {
                    in_basis = false;
                    if (L.type == 1u) {
                        const float rx = rng_range(rng, L.p0, L.p1);
                        const float rz = rng_range(rng, L.p2, L.p3);
                        dir = v3(rx, L.p4, rz) - sf.p;
                    } else {
                        dir = v3(1.0f, 0.0f, 0.0f);
                    }
                }
            }
            if (in_basis) {
                const float r1 = rng_f32(rng);
                const float r2 = rng_f32(rng);
                float z, radial;
                if (use_cosine) {
                    z = sqrt_rn(1.0f - r2);
                    radial = sqrt_rn(r2);
                } else {
                    z = 1.0f + r2 * (sqrt_rn(1.0f - sphere_r2_over_d2) - 1.0f);
                    radial = sqrt_rn(1.0f - z * z);
                }
                float sn, cs;
                sincos_turn(r1, sn, cs);  // phi = 2 pi r1
                dir = onb_local(basis, v3(cs * radial, sn * radial, z));
            }
            unit_dir = unit(dir);
            const float cos_pdf = fmaxf(dot(unit_dir, uvw.w) * RRT_INV_PI, 0.0f);
            pdf_val = 0.5f * cos_pdf + 0.5f * lights_pdf(S, sf.p, dir);
        } else {
            dir = onb_local(uvw, random_cosine_direction(rng));
            unit_dir = unit(dir);
            pdf_val = fmaxf(dot(unit_dir, uvw.w) * RRT_INV_PI, 0.0f);
        }
        pdf_val = fmaxf(pdf_val, 1e-5f);
        const float s_pdf = fmaxf(dot(sf.n, unit_dir) * RRT_INV_PI, 0.0f);
        // emitted == 0 for Lambertian;  L = s_pdf * (albedo * L_next) / pdf_val
        beta = mulv(beta, (s_pdf * rcp_1ulp(pdf_val)) * rgb);
        o = sf.p;
        d = dir;
        return true;
    }
    if (kind == 1u) {  // Metal, material/mod.rs:91-108
        const float fuzz = S.mats[sf.material].param;
        const V3 reflected = reflect(unit(d), sf.n);
        const V3 dir = reflected + fuzz * random_in_unit_sphere(rng);
        if (dot(dir, sf.n) > 0.0f) {
            beta = mulv(beta, rgb);
            o = sf.p;
            d = dir;
            return true;
        }
        return false;  // absorbed: scatter None, emitted 0
    }
    {  // Dielectric, material/mod.rs:114-148
        const float ir = S.mats[sf.material].param;
        float ratio = 1.0f / ir;
        V3 n = sf.n;
        if (dot(d, sf.n) > 0.0f) {
            ratio = ir;
            n = -sf.n;
        }
        const V3 unit = normalize(d);
        const float cos_theta = -fminf(dot(unit, n), 1.0f);
        // refract(), material/mod.rs:154-163
        const float sin_theta = sqrt_rn(1.0f - cos_theta * cos_theta);
        bool refracted = false;
        V3 out;
        if (!(ratio * sin_theta > 1.0f)) {
            const V3 r_out_ortho = ratio * (unit + cos_theta * n);
            const V3 r_out_para = (-sqrt_rn(fabsf(1.0f - length_squared(r_out_ortho)))) * n;
            const float r0 = (1.0f - ir) / (1.0f + ir);
            const float r0s = r0 * r0;
            const float reflectance = r0s + (1.0f - r0s) * powi5(1.0f - cos_theta);
            if (reflectance < rng_f32(rng)) {
                refracted = true;
                out = r_out_ortho + r_out_para;
            }
        }
        if (!refracted) out = reflect(unit, sf.n);
        o = sf.p;
        d = out;
        return true;  // attenuation (1,1,1)
    }
}

}  // namespace rrt
#endif
