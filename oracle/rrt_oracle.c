/*
 * rrt_oracle.c -- CPU restatement of the reference's path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this
 * file; the product (librrt_hip.so) never does.  PARITY UNPINNED: the reference (Safarte/rust-ray-tracer)
 * ships no tests, golden vectors or images, and its Rust toolchain is absent here, so this restatement is
 * pinned only by analytic known-answer cases derived from the reference's source text.  The arithmetic of
 * the un-vendored crates is restated from their published behaviour: glam 0.20.2 (Vec3A on SSE2:
 * dot = (x*x' + y*y') + z*z'; cross = mul, mul, sub; normalize = v / sqrt(dot); any_orthonormal_pair =
 * Duff et al.), rand 0.8.3 distributions (Standard f32 = 24 bits, gen_range f32 = 23 bits scaled,
 * gen_bool(0.5) = one bit, gen_range(0..n) = unbiased integer).
 *
 * Each function cites the reference file:line it follows (paths relative to the reference repository).
 * Compile with -O2 -ffp-contract=off -fno-fast-math so that every float expression rounds as written.
 *
 * The reference draws from an OS-seeded thread-local ChaCha12 stream (src/main.rs:88 and 20 more call
 * sites), which nobody can reproduce.  The restatement keeps the reference's DISTRIBUTIONS and call ORDER
 * but draws from the counter-based generator the GPU kernels use (keyed by seed, pixel, sample), so that a
 * CPU image and a GPU image of the same seed follow the same paths.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rrt_ffi.h"

/* ------------------------------------------------------------------ vectors (glam Vec3A semantics) */
typedef struct { float x, y, z; } vec3;
static inline vec3 V(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
static inline vec3 vadd(vec3 a, vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 vsub(vec3 a, vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 vneg(vec3 a) { return V(-a.x, -a.y, -a.z); }
static inline vec3 vscale(float s, vec3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline vec3 vdivs(vec3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline vec3 vmul(vec3 a, vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); } /* src/vec3.rs:5-7 */
static inline float vdot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline vec3 vcross(vec3 a, vec3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float vlen2(vec3 a) { return vdot(a, a); }
static inline vec3 vnormalize(vec3 a) { return vdivs(a, sqrtf(vdot(a, a))); }
static inline float vget(vec3 a, unsigned i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline float rust_signum(float x) { return isnan(x) ? x : copysignf(1.0f, x); }
/* Rust f32::max / f32::min ignore a NaN operand */
static inline float rust_max(float a, float b) { return isnan(a) ? b : (isnan(b) ? a : (a > b ? a : b)); }
static inline float rust_min(float a, float b) { return isnan(a) ? b : (isnan(b) ? a : (a < b ? a : b)); }
#define PI_F 3.14159274101257324f

/* ------------------------------------------------------------------ counter-based generator */
typedef struct { uint32_t k0, k1, ctr; } rng_t;
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static rng_t rng_start(uint32_t seed, uint32_t pixel, uint32_t sample) {
    rng_t r;
    uint32_t h0 = mix32(seed ^ 0xA511E9B3U);
    uint32_t h1 = mix32(h0 + pixel);
    r.k0 = mix32(h1 ^ (sample * 0x9E3779B9U));
    r.k1 = mix32(r.k0 ^ 0x68E31DA4U ^ pixel);
    r.ctr = 0;
    return r;
}
static uint32_t rng_next(rng_t* r) {
    uint32_t x = r->k0 + r->ctr * 0x9E3779B9U;
    r->ctr++;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x ^= r->k1; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
/* rand 0.8.3 Standard: f32 from the top 24 bits */
static float gen_f32(rng_t* r) { return (float)(rng_next(r) >> 8) * (1.0f / 16777216.0f); }
/* rand 0.8.3 `gen_range(low..high)` for f32 = UniformFloat::sample_single (rand-0.8.3 src/distributions/uniform.rs):
 * a value in [1,2) from the top 23 bits, minus 1, times scale = high - low, plus low; if rounding carried the result up
 * to `high`, scale is decreased by one ulp and another value is drawn.  (Uniform::new(..).sample(), which the
 * reference does not use, is the other formula: value1_2 * scale + (low - scale).)  rand loops without a bound; here the
 * loop gives up after 8 draws and returns low -- with the decreasing scale the second draw already succeeds unless it
 * is again the largest of the 2^23 values. */
static float gen_range_f32(rng_t* r, float low, float high) {
    float scale = high - low;
    int tries;
    for (tries = 0; tries < 8; ++tries) {
        uint32_t bits = (rng_next(r) >> 9) | 0x3f800000U;
        float v12;
        memcpy(&v12, &bits, 4);
        float res = (v12 - 1.0f) * scale + low;
        if (res < high) return res;
        uint32_t sb;
        memcpy(&sb, &scale, 4);
        sb -= 1u; /* decrease_masked: the next float below */
        memcpy(&scale, &sb, 4);
    }
    return low;
}
static int gen_bool_half(rng_t* r) { return (rng_next(r) >> 31) == 0u; } /* Bernoulli(0.5) */
static uint32_t gen_index(rng_t* r, uint32_t n) { /* gen_range(0..n) */
    uint32_t zone = (0u - n) % n;
    for (;;) {
        uint64_t m = (uint64_t)rng_next(r) * (uint64_t)n;
        if ((uint32_t)m >= zone) return (uint32_t)(m >> 32);
    }
}

/* exported for the distribution tests */
void oracle_rng_fill(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n, uint32_t* out) {
    rng_t r = rng_start(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng_next(&r);
}

/* ------------------------------------------------------------------ flattened world */
enum { P_TRI = 0, P_SPHERE = 1, P_RECT = 2 };
typedef struct {
    int kind;
    uint32_t material;
    /* triangle: src/geometry/triangle.rs:13-33 */
    vec3 v0, v0v1, v0v2;
    /* sphere: src/geometry/sphere.rs:17-22 */
    vec3 center; float radius;
    /* rect: src/geometry/aarect.rs:14-21 */
    unsigned axis; float a0, a1, b0, b1, k; int flip_face;
    /* instance wrapper (Translate(RotateY(..))), src/geometry/transform.rs */
    int instanced; float sin_t, cos_t; vec3 offset;
    /* bounding box as the reference computes it (for the pointer-BVH mode) */
    vec3 bmin, bmax;
} prim_t;

typedef struct bvh_node {
    struct bvh_node *left, *right; /* NULL children mean leaf */
    int leaf_prim;                 /* >= 0 for leaves */
    vec3 bmin, bmax;
} bvh_node;

typedef struct {
    prim_t* prims; uint32_t n_prims;
    rrt_material* mats; uint32_t n_mats;
    rrt_light* lights; uint32_t n_lights;
    rrt_camera cam;
    vec3 background;
    bvh_node* root;       /* pointer BVH in the reference's style, built on demand */
    bvh_node* pool; size_t pool_used, pool_cap;
} oracle_scene;

static float min3f(float a, float b, float c) { return rust_min(a, rust_min(b, c)); }
static float max3f(float a, float b, float c) { return rust_max(a, rust_max(b, c)); }

static void push_tri(prim_t* p, const rrt_triangle* t) {
    memset(p, 0, sizeof *p);
    p->kind = P_TRI; p->material = t->material;
    vec3 v0 = V(t->v0[0], t->v0[1], t->v0[2]), v1 = V(t->v1[0], t->v1[1], t->v1[2]), v2 = V(t->v2[0], t->v2[1], t->v2[2]);
    p->v0 = v0; p->v0v1 = vsub(v1, v0); p->v0v2 = vsub(v2, v0); /* triangle.rs:28-29 */
    /* triangle.rs:82-94 */
    p->bmin = V(min3f(v0.x, v1.x, v2.x) - 0.0001f, min3f(v0.y, v1.y, v2.y) - 0.0001f, min3f(v0.z, v1.z, v2.z) - 0.0001f);
    p->bmax = V(max3f(v0.x, v1.x, v2.x) + 0.0001f, max3f(v0.y, v1.y, v2.y) + 0.0001f, max3f(v0.z, v1.z, v2.z) + 0.0001f);
}
static void push_sphere(prim_t* p, const rrt_sphere* s) {
    memset(p, 0, sizeof *p);
    p->kind = P_SPHERE; p->material = s->material;
    p->center = V(s->center[0], s->center[1], s->center[2]); p->radius = s->radius;
    p->bmin = vsub(p->center, V(s->radius, s->radius, s->radius)); /* sphere.rs:128-133 */
    p->bmax = vadd(p->center, V(s->radius, s->radius, s->radius));
}
static void push_rect(prim_t* p, const rrt_rect* r) {
    memset(p, 0, sizeof *p);
    p->kind = P_RECT; p->material = r->material;
    p->axis = r->axis; p->a0 = r->a0; p->a1 = r->a1; p->b0 = r->b0; p->b1 = r->b1; p->k = r->k;
    p->flip_face = (r->flags & RRT_RECT_FLIP_FACE) != 0;
    /* aarect.rs:60-66 / 115-121 / 191-197 */
    if (r->axis == 2) { p->bmin = V(r->a0, r->b0, r->k - 0.0001f); p->bmax = V(r->a1, r->b1, r->k + 0.0001f); }
    else if (r->axis == 1) { p->bmin = V(r->a0, r->k - 0.0001f, r->b0); p->bmax = V(r->a1, r->k + 0.0001f, r->b1); }
    else { p->bmin = V(r->k - 0.0001f, r->a0, r->b0); p->bmax = V(r->k + 0.0001f, r->a1, r->b1); }
}
static void cuboid_sides(const rrt_cuboid* c, rrt_rect out[6]) { /* cuboid.rs:24-73 */
    const float *mn = c->min, *mx = c->max;
    rrt_rect s[6] = {
        {2u, mn[0], mx[0], mn[1], mx[1], mx[2], c->material, 0u}, {2u, mn[0], mx[0], mn[1], mx[1], mn[2], c->material, 0u},
        {1u, mn[0], mx[0], mn[2], mx[2], mx[1], c->material, 0u}, {1u, mn[0], mx[0], mn[2], mx[2], mn[1], c->material, 0u},
        {0u, mn[1], mx[1], mn[2], mx[2], mx[0], c->material, 0u}, {0u, mn[1], mx[1], mn[2], mx[2], mn[0], c->material, 0u}};
    memcpy(out, s, sizeof s);
}

static uint32_t count_objects(const rrt_scene_desc* d, const rrt_object* objs, uint32_t n) {
    uint32_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const rrt_object* o = &objs[i];
        if (o->type == RRT_OBJ_CUBOIDS) total += 6u * o->count;
        else if (o->type == RRT_OBJ_INSTANCE) {
            const rrt_mesh* m = &d->meshes[d->instances[o->first].mesh];
            total += count_objects(d, d->mesh_objects + m->first_object, m->object_count);
        } else total += o->count;
    }
    return total;
}
/* Appends the primitives of an object list in id order; `inst` wraps them in Translate(RotateY(..)). */
static uint32_t expand_objects(const rrt_scene_desc* d, const rrt_object* objs, uint32_t n, const rrt_instance* inst, prim_t* out) {
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const rrt_object* o = &objs[i];
        uint32_t first_new = k;
        if (o->type == RRT_OBJ_TRIANGLES) for (uint32_t j = 0; j < o->count; ++j) push_tri(&out[k++], &d->triangles[o->first + j]);
        else if (o->type == RRT_OBJ_SPHERES) for (uint32_t j = 0; j < o->count; ++j) push_sphere(&out[k++], &d->spheres[o->first + j]);
        else if (o->type == RRT_OBJ_RECTS) for (uint32_t j = 0; j < o->count; ++j) push_rect(&out[k++], &d->rects[o->first + j]);
        else if (o->type == RRT_OBJ_CUBOIDS) {
            for (uint32_t j = 0; j < o->count; ++j) {
                rrt_rect s[6];
                cuboid_sides(&d->cuboids[o->first + j], s);
                for (int q = 0; q < 6; ++q) push_rect(&out[k++], &s[q]);
            }
        } else if (o->type == RRT_OBJ_INSTANCE) {
            const rrt_instance* in = &d->instances[o->first];
            const rrt_mesh* m = &d->meshes[in->mesh];
            k += expand_objects(d, d->mesh_objects + m->first_object, m->object_count, in, &out[k]);
            continue;
        }
        if (inst)
            for (uint32_t j = first_new; j < k; ++j) {
                out[j].instanced = 1; out[j].sin_t = inst->sin_theta; out[j].cos_t = inst->cos_theta;
                out[j].offset = V(inst->offset[0], inst->offset[1], inst->offset[2]);
            }
    }
    return k;
}

oracle_scene* oracle_scene_create(const rrt_scene_desc* d) {
    oracle_scene* s = (oracle_scene*)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->n_prims = count_objects(d, d->world, d->world_count);
    s->prims = (prim_t*)calloc(s->n_prims ? s->n_prims : 1, sizeof(prim_t));
    expand_objects(d, d->world, d->world_count, NULL, s->prims);
    s->n_mats = d->material_count;
    s->mats = (rrt_material*)malloc(sizeof(rrt_material) * (d->material_count ? d->material_count : 1));
    memcpy(s->mats, d->materials, sizeof(rrt_material) * d->material_count);
    s->n_lights = d->light_count;
    s->lights = (rrt_light*)malloc(sizeof(rrt_light) * (d->light_count ? d->light_count : 1));
    if (d->light_count) memcpy(s->lights, d->lights, sizeof(rrt_light) * d->light_count);
    s->cam = d->camera;
    s->background = V(d->background[0], d->background[1], d->background[2]);
    return s;
}
void oracle_scene_destroy(oracle_scene* s) {
    if (!s) return;
    free(s->prims); free(s->mats); free(s->lights); free(s->pool); free(s);
}
uint32_t oracle_scene_prim_count(const oracle_scene* s) { return s->n_prims; }

/* ------------------------------------------------------------------ primitive tests */
typedef struct { vec3 o, d; } ray_t;
typedef struct {
    int hit; int prim; float t, u, v; vec3 p, normal; uint32_t material;
} hitrec;

static vec3 ray_at(ray_t r, float t) { return vadd(r.o, vscale(t, r.d)); } /* src/ray.rs:41-43 */

/* src/geometry/triangle.rs:38-69 */
static int hit_triangle(const prim_t* p, ray_t r, float t_min, float t_max, hitrec* rec) {
    vec3 pvec = vcross(r.d, p->v0v2);
    float det = vdot(p->v0v1, pvec);
    if (det > 1e-5f) { /* double_sided is always false (triangle.rs:27) */
        float inv_det = 1.0f / det;
        vec3 tvec = vsub(r.o, p->v0);
        float u = vdot(tvec, pvec) * inv_det;
        if (0.0f <= u && u <= 1.0f) {
            vec3 qvec = vcross(tvec, p->v0v1);
            float v = vdot(r.d, qvec) * inv_det;
            if (0.0f <= v && v < 1.0f - u) {
                float t = vdot(p->v0v2, qvec) * inv_det;
                if (t_min <= t && t <= t_max) {
                    rec->hit = 1; rec->t = t; rec->u = u; rec->v = v;
                    rec->p = ray_at(r, t);
                    rec->normal = vscale(rust_signum(det), vnormalize(vcross(p->v0v1, p->v0v2)));
                    rec->material = p->material;
                    return 1;
                }
            }
        }
    }
    return 0;
}
/* src/geometry/sphere.rs:34-40 */
static void sphere_uv(vec3 p, float* u, float* v) {
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + PI_F;
    *u = phi / (2.0f * PI_F);
    *v = theta / PI_F;
}
/* src/geometry/sphere.rs:52-95 */
static int hit_sphere_raw(vec3 center, float radius, ray_t r, float t_min, float t_max, hitrec* rec) {
    vec3 oc = vsub(r.o, center);
    float a = vlen2(r.d);
    float b = vdot(oc, r.d);
    float c = vlen2(oc) - radius * radius;
    float disc = b * b - a * c;
    if (disc > 0.0f) {
        float sqrtd = sqrtf(disc);
        float root = (-b - sqrtd) / a;
        if (t_min <= root && root <= t_max) {
            rec->hit = 1; rec->t = root; rec->p = ray_at(r, root);
            rec->normal = vdivs(vsub(rec->p, center), radius);
            sphere_uv(rec->normal, &rec->u, &rec->v);
            return 1;
        }
        root = (-b + sqrtd) / a;
        if (t_min <= root && root <= t_max) {
            rec->hit = 1; rec->t = root; rec->p = ray_at(r, root);
            sphere_uv(rec->p, &rec->u, &rec->v); /* sphere.rs:82: un-normalised p, as in the reference */
            rec->normal = vdivs(vsub(rec->p, center), radius);
            return 1;
        }
    }
    return 0;
}
/* src/geometry/aarect.rs:39-58, 94-113, 170-189 */
static int hit_rect_raw(unsigned axis, float a0, float a1, float b0, float b1, float k, ray_t r, float t_min, float t_max, hitrec* rec) {
    unsigned ia = axis == 0 ? 1u : 0u, ib = axis == 2 ? 1u : 2u;
    float t = (k - vget(r.o, axis)) / vget(r.d, axis);
    if (t >= t_min && t <= t_max) {
        float a = vget(r.o, ia) + t * vget(r.d, ia);
        float b = vget(r.o, ib) + t * vget(r.d, ib);
        if (a >= a0 && a <= a1 && b >= b0 && b <= b1) {
            float sg = rust_signum(vget(r.o, axis) - k);
            rec->hit = 1; rec->t = t; rec->p = ray_at(r, t);
            rec->normal = V(axis == 0 ? sg : 0.0f, axis == 1 ? sg : 0.0f, axis == 2 ? sg : 0.0f);
            rec->u = (a - a0) / (a1 - a0);
            rec->v = (b - b0) / (b1 - b0);
            return 1;
        }
    }
    return 0;
}

static int hit_prim_local(const prim_t* p, ray_t r, float t_min, float t_max, hitrec* rec) {
    int h = 0;
    if (p->kind == P_TRI) h = hit_triangle(p, r, t_min, t_max, rec);
    else if (p->kind == P_SPHERE) { h = hit_sphere_raw(p->center, p->radius, r, t_min, t_max, rec); if (h) rec->material = p->material; }
    else {
        h = hit_rect_raw(p->axis, p->a0, p->a1, p->b0, p->b1, p->k, r, t_min, t_max, rec);
        if (h) {
            rec->material = p->material;
            if (p->flip_face) rec->normal.y = -fabsf(rec->normal.y); /* src/geometry/mod.rs:226-230 */
        }
    }
    return h;
}
/* Translate::hit (transform.rs:22-31) around RotateY::hit (transform.rs:99-130) around the primitive */
static int hit_prim(const prim_t* p, ray_t r, float t_min, float t_max, hitrec* rec) {
    if (!p->instanced) return hit_prim_local(p, r, t_min, t_max, rec);
    ray_t moved = {vsub(r.o, p->offset), r.d};
    ray_t rot = moved;
    rot.o.x = p->cos_t * moved.o.x - p->sin_t * moved.o.z;
    rot.o.z = p->sin_t * moved.o.x + p->cos_t * moved.o.z;
    rot.d.x = p->cos_t * moved.d.x - p->sin_t * moved.d.z;
    rot.d.z = p->sin_t * moved.d.x + p->cos_t * moved.d.z;
    if (!hit_prim_local(p, rot, t_min, t_max, rec)) return 0;
    vec3 pp = rec->p, nn = rec->normal;
    pp.x = p->cos_t * rec->p.x + p->sin_t * rec->p.z;
    pp.z = -p->sin_t * rec->p.x + p->cos_t * rec->p.z;
    nn.x = p->cos_t * rec->normal.x + p->sin_t * rec->normal.z;
    nn.z = -p->sin_t * rec->normal.x + p->cos_t * rec->normal.z;
    rec->normal = vscale(rust_signum(vdot(nn, rot.d)), vneg(nn));
    rec->p = vadd(pp, p->offset);
    return 1;
}

/* Closest hit by exhaustive search: minimum of (t, primitive id) over all primitives that accept the ray
 * in [t_min, t_max].  The reference resolves equal-t ties by the order of its randomly built tree
 * (src/geometry/mod.rs:142-147,166), which is not reproducible; lowest id is this build's stated rule. */
static int closest_brute(const oracle_scene* s, ray_t r, float t_min, float t_max, hitrec* best) {
    best->hit = 0; best->prim = -1;
    float limit = t_max;
    for (uint32_t i = 0; i < s->n_prims; ++i) {
        hitrec rec;
        memset(&rec, 0, sizeof rec);
        if (hit_prim(&s->prims[i], r, t_min, limit, &rec)) {
            if (!best->hit || rec.t < best->t) { *best = rec; best->prim = (int)i; limit = rec.t; }
        }
    }
    return best->hit;
}

/* ------------------------------------------------------------------ pointer BVH in the reference's style */
/* src/bvh/aabb.rs:14-36 -- note: every axis is clipped against the ORIGINAL [t_min, t_max] */
static int aabb_hit(vec3 bmin, vec3 bmax, ray_t r, float t_min, float t_max) {
    for (unsigned a = 0; a < 3; ++a) {
        float inv_d = 1.0f / vget(r.d, a);
        float t0 = (vget(bmin, a) - vget(r.o, a)) * inv_d;
        float t1 = (vget(bmax, a) - vget(r.o, a)) * inv_d;
        float lo, hi;
        if (inv_d < 0.0f) { lo = rust_max(t1, t_min); hi = rust_min(t0, t_max); }
        else { lo = rust_max(t0, t_min); hi = rust_min(t1, t_max); }
        if (hi <= lo) return 0;
    }
    return 1;
}
/* A slab test that does carry the interval from axis to axis (what the reference's loop presumably meant to be):
 * used only for the second, clearly labelled CPU baseline, which walks the same reference-style tree with a box test
 * that actually culls.  Not the reference's behaviour. */
static int aabb_hit_running(vec3 bmin, vec3 bmax, ray_t r, float t_min, float t_max) {
    for (unsigned a = 0; a < 3; ++a) {
        float inv_d = 1.0f / vget(r.d, a);
        float t0 = (vget(bmin, a) - vget(r.o, a)) * inv_d;
        float t1 = (vget(bmax, a) - vget(r.o, a)) * inv_d;
        if (inv_d < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        t_min = rust_max(t0, t_min);
        t_max = rust_min(t1, t_max);
        if (t_max <= t_min) return 0;
    }
    return 1;
}
typedef struct { uint32_t idx; float key; } sort_item;
static int cmp_item(const void* a, const void* b) {
    float x = ((const sort_item*)a)->key, y = ((const sort_item*)b)->key;
    if (x < y) return -1;
    if (x > y) return 1;
    /* stable like Rust's sort_by */
    uint32_t i = ((const sort_item*)a)->idx, j = ((const sort_item*)b)->idx;
    return i < j ? -1 : (i > j ? 1 : 0);
}
static bvh_node* new_node(oracle_scene* s) { return &s->pool[s->pool_used++]; }
static bvh_node* leaf_for(oracle_scene* s, uint32_t prim) {
    bvh_node* n = new_node(s);
    n->left = n->right = NULL; n->leaf_prim = (int)prim;
    n->bmin = s->prims[prim].bmin; n->bmax = s->prims[prim].bmax;
    return n;
}
/* src/geometry/mod.rs:161-199: random axis in {x, y}; span 1 duplicates the object; span 2 orders the pair;
 * otherwise sort by box.min[axis] and split in the middle. */
static bvh_node* build_ref_bvh(oracle_scene* s, uint32_t* ids, uint32_t n, uint32_t* axis_state) {
    *axis_state = mix32(*axis_state + 0x9E3779B9U);
    unsigned axis = (*axis_state >> 31) & 1u;
    bvh_node* node = new_node(s);
    node->leaf_prim = -1;
    if (n == 1) {
        node->left = leaf_for(s, ids[0]);
        node->right = node->left;
    } else if (n == 2) {
        float k0 = vget(s->prims[ids[0]].bmin, axis), k1 = vget(s->prims[ids[1]].bmin, axis);
        if (k0 < k1) { node->left = leaf_for(s, ids[0]); node->right = leaf_for(s, ids[1]); }
        else { node->left = leaf_for(s, ids[1]); node->right = leaf_for(s, ids[0]); }
    } else {
        sort_item* items = (sort_item*)malloc(sizeof(sort_item) * n);
        for (uint32_t i = 0; i < n; ++i) { items[i].idx = ids[i]; items[i].key = vget(s->prims[ids[i]].bmin, axis); }
        /* keep submission order for equal keys: sort on (key, position) */
        for (uint32_t i = 0; i < n; ++i) items[i].idx = i;
        qsort(items, n, sizeof(sort_item), cmp_item);
        uint32_t* sorted = (uint32_t*)malloc(sizeof(uint32_t) * n);
        for (uint32_t i = 0; i < n; ++i) sorted[i] = ids[items[i].idx];
        memcpy(ids, sorted, sizeof(uint32_t) * n);
        free(sorted); free(items);
        uint32_t mid = n / 2;
        node->left = build_ref_bvh(s, ids, mid, axis_state);
        node->right = build_ref_bvh(s, ids + mid, n - mid, axis_state);
    }
    /* surrounding_box, src/bvh/aabb.rs:39-56 */
    node->bmin = V(rust_min(node->left->bmin.x, node->right->bmin.x), rust_min(node->left->bmin.y, node->right->bmin.y), rust_min(node->left->bmin.z, node->right->bmin.z));
    node->bmax = V(rust_max(node->left->bmax.x, node->right->bmax.x), rust_max(node->left->bmax.y, node->right->bmax.y), rust_max(node->left->bmax.z, node->right->bmax.z));
    return node;
}
/* Instanced primitives carry object-space boxes; the reference-style tree is only offered for flat scenes. */
int oracle_scene_build_ref_bvh(oracle_scene* s, uint32_t axis_seed) {
    if (s->n_prims == 0) return -1;
    for (uint32_t i = 0; i < s->n_prims; ++i) if (s->prims[i].instanced) return -2;
    free(s->pool);
    s->pool_cap = (size_t)4 * s->n_prims + 8; s->pool_used = 0;
    s->pool = (bvh_node*)calloc(s->pool_cap, sizeof(bvh_node));
    uint32_t* ids = (uint32_t*)malloc(sizeof(uint32_t) * s->n_prims);
    for (uint32_t i = 0; i < s->n_prims; ++i) ids[i] = i;
    uint32_t st = axis_seed;
    s->root = build_ref_bvh(s, ids, s->n_prims, &st);
    free(ids);
    return 0;
}
/* src/geometry/mod.rs:136-153 */
static int bvh_hit(const oracle_scene* s, const bvh_node* n, ray_t r, float t_min, float t_max, hitrec* rec, int running) {
    if (n->leaf_prim >= 0) {
        hitrec tmp;
        memset(&tmp, 0, sizeof tmp);
        if (hit_prim(&s->prims[n->leaf_prim], r, t_min, t_max, &tmp)) { *rec = tmp; rec->prim = n->leaf_prim; return 1; }
        return 0;
    }
    if (!(running ? aabb_hit_running(n->bmin, n->bmax, r, t_min, t_max) : aabb_hit(n->bmin, n->bmax, r, t_min, t_max))) return 0;
    hitrec l, rr;
    if (bvh_hit(s, n->left, r, t_min, t_max, &l, running)) {
        if (bvh_hit(s, n->right, r, t_min, l.t, &rr, running)) { *rec = rr; return 1; }
        *rec = l;
        return 1;
    }
    if (bvh_hit(s, n->right, r, t_min, t_max, &rr, running)) { *rec = rr; return 1; }
    return 0;
}

/* HIT_REF_BVH: the reference's tree and box test.  HIT_REF_BVH_RUNNING_SLAB: the same tree with aabb_hit_running. */
enum { HIT_BRUTE = 0, HIT_REF_BVH = 1, HIT_REF_BVH_RUNNING_SLAB = 2 };
static int world_hit(const oracle_scene* s, int mode, ray_t r, float t_min, float t_max, hitrec* rec) {
    if ((mode == HIT_REF_BVH || mode == HIT_REF_BVH_RUNNING_SLAB) && s->root) {
        rec->hit = 0; rec->prim = -1;
        int h = bvh_hit(s, s->root, r, t_min, t_max, rec, mode == HIT_REF_BVH_RUNNING_SLAB);
        rec->hit = h;
        return h;
    }
    return closest_brute(s, r, t_min, t_max, rec);
}

/* ------------------------------------------------------------------ batch intersection (parity hook) */
typedef struct {
    const oracle_scene* s; int mode; const rrt_ray* rays; size_t begin, end; float t_min, t_max; rrt_hit* out;
} isect_job;
static void* isect_worker(void* arg) {
    isect_job* j = (isect_job*)arg;
    for (size_t i = j->begin; i < j->end; ++i) {
        ray_t r = {V(j->rays[i].origin[0], j->rays[i].origin[1], j->rays[i].origin[2]),
                   V(j->rays[i].direction[0], j->rays[i].direction[1], j->rays[i].direction[2])};
        hitrec rec;
        if (world_hit(j->s, j->mode, r, j->t_min, j->t_max, &rec)) {
            j->out[i].prim_id = rec.prim; j->out[i].t = rec.t; j->out[i].u = rec.u; j->out[i].v = rec.v;
        } else {
            j->out[i].prim_id = -1; j->out[i].t = INFINITY; j->out[i].u = 0.f; j->out[i].v = 0.f;
        }
    }
    return NULL;
}
void oracle_intersect(const oracle_scene* s, int mode, const rrt_ray* rays, size_t n, float t_min, float t_max, rrt_hit* out, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64];
    isect_job jobs[64];
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t b = (size_t)t * per, e = b + per < n ? b + per : n;
        if (b >= e) break;
        isect_job j = {s, mode, rays, b, e, t_min, t_max, out};
        jobs[t] = j;
        pthread_create(&th[t], NULL, isect_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------ camera, src/camera.rs:56-69 */
static ray_t camera_get_ray(const rrt_camera* c, float x, float y, uint32_t w, uint32_t h) {
    float px = (2.0f * (x + 0.5f) / (float)w - 1.0f) * c->scale * c->aspect;
    float py = (2.0f * (y + 0.5f) / (float)h - 1.0f) * c->scale;
    vec3 c0 = V(c->c2w[0], c->c2w[1], c->c2w[2]), c1 = V(c->c2w[3], c->c2w[4], c->c2w[5]);
    vec3 c2 = V(c->c2w[6], c->c2w[7], c->c2w[8]), tr = V(c->c2w[9], c->c2w[10], c->c2w[11]);
    /* Affine3A::transform_point3a: ((x_axis*x + y_axis*y) + z_axis*z) + translation */
    vec3 ray_p = vadd(vadd(vadd(vscale(px, c0), vscale(py, c1)), vscale(-1.0f, c2)), tr);
    ray_t r = {V(c->origin[0], c->origin[1], c->origin[2]), V(0, 0, 0)};
    r.d = vsub(ray_p, r.o);
    return r;
}
void oracle_camera_rays(const rrt_camera* c, uint32_t w, uint32_t h, const float* xy, size_t n, rrt_ray* out) {
    for (size_t i = 0; i < n; ++i) {
        ray_t r = camera_get_ray(c, xy[2 * i], xy[2 * i + 1], w, h);
        out[i].origin[0] = r.o.x; out[i].origin[1] = r.o.y; out[i].origin[2] = r.o.z;
        out[i].direction[0] = r.d.x; out[i].direction[1] = r.d.y; out[i].direction[2] = r.d.z;
        out[i].time = 0.f;
    }
}

/* ------------------------------------------------------------------ ONB, src/vec3.rs:55-73 + glam any_orthonormal_pair */
typedef struct { vec3 u, v, w; } onb_t;
static onb_t onb_from_w(vec3 n) {
    onb_t o;
    o.w = vnormalize(n);
    float sign = copysignf(1.0f, o.w.z);
    float a = -1.0f / (sign + o.w.z);
    float b = o.w.x * o.w.y * a;
    o.u = V(1.0f + sign * o.w.x * o.w.x * a, sign * b, -sign * o.w.x);
    o.v = V(b, sign + o.w.y * o.w.y * a, -o.w.y);
    return o;
}
static vec3 onb_local(const onb_t* o, vec3 a) { return vadd(vadd(vscale(a.x, o->u), vscale(a.y, o->v)), vscale(a.z, o->w)); }

/* ------------------------------------------------------------------ light list */
/* src/geometry/sphere.rs:217-228 */
static vec3 random_to_sphere(float radius, float dist_squared, rng_t* rng) {
    float r1 = gen_f32(rng);
    float r2 = gen_f32(rng);
    float z = 1.0f + r2 * (sqrtf(1.0f - radius * radius / dist_squared) - 1.0f);
    float phi = 2.0f * PI_F * r1;
    float x = cosf(phi) * sqrtf(1.0f - z * z);
    float y = sinf(phi) * sqrtf(1.0f - z * z);
    return V(x, y, z);
}
/* Hittables::random, src/geometry/mod.rs:122-125 -> Sphere::random sphere.rs:114-119 / XZRect::random aarect.rs:134-142 */
static vec3 lights_random(const oracle_scene* s, vec3 origin, rng_t* rng) {
    const rrt_light* L = &s->lights[gen_index(rng, s->n_lights)];
    if (L->type == RRT_LIGHT_SPHERE) {
        vec3 direction = vsub(V(L->p[0], L->p[1], L->p[2]), origin);
        float dist_squared = vlen2(direction);
        onb_t uvw = onb_from_w(direction);
        return onb_local(&uvw, random_to_sphere(L->p[3], dist_squared, rng));
    }
    if (L->type == RRT_LIGHT_XZRECT) {
        float rx = gen_range_f32(rng, L->p[0], L->p[1]);
        float rz = gen_range_f32(rng, L->p[2], L->p[3]);
        return vsub(V(rx, L->p[4], rz), origin);
    }
    return V(1.0f, 0.0f, 0.0f); /* trait default, src/geometry/mod.rs:59-61 */
}
/* Hittables::pdf_value, src/geometry/mod.rs:111-120 -> sphere.rs:101-112 / aarect.rs:123-132 */
static float lights_pdf_value(const oracle_scene* s, vec3 origin, vec3 v) {
    float weight = 1.0f / (float)s->n_lights;
    float sum = 0.0f;
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const rrt_light* L = &s->lights[i];
        float pdf = 0.0f;
        ray_t r = {origin, v};
        hitrec rec;
        if (L->type == RRT_LIGHT_SPHERE) {
            vec3 c = V(L->p[0], L->p[1], L->p[2]);
            if (hit_sphere_raw(c, L->p[3], r, 0.0001f, INFINITY, &rec)) {
                float cos_theta_max = sqrtf(1.0f - L->p[3] * L->p[3] / vlen2(vsub(c, origin)));
                float solid_angle = 2.0f * PI_F * (1.0f - cos_theta_max);
                pdf = 1.0f / solid_angle;
            }
        } else if (L->type == RRT_LIGHT_XZRECT) {
            if (hit_rect_raw(1u, L->p[0], L->p[1], L->p[2], L->p[3], L->p[4], r, 0.001f, INFINITY, &rec)) {
                float area = (L->p[1] - L->p[0]) * (L->p[3] - L->p[2]);
                float dist_squared = rec.t * rec.t * vlen2(v);
                float cosine = fabsf(vdot(v, rec.normal)) / sqrtf(vlen2(v));
                pdf = dist_squared / (cosine * area);
            }
        }
        sum += weight * pdf;
    }
    return sum;
}

/* ------------------------------------------------------------------ pdfs, src/pdf.rs */
static vec3 random_cosine_direction(rng_t* rng) { /* pdf.rs:14-26 */
    float r1 = gen_f32(rng);
    float r2 = gen_f32(rng);
    float z = sqrtf(1.0f - r2);
    float phi = 2.0f * PI_F * r1;
    float sr2 = sqrtf(r2);
    float x = cosf(phi) * sr2;
    float y = sinf(phi) * sr2;
    return V(x, y, z);
}
static float cosine_pdf_value(const onb_t* uvw, vec3 direction) { /* pdf.rs:41-44 */
    float cosine = vdot(vnormalize(direction), uvw->w);
    return rust_max(cosine / PI_F, 0.0f);
}

/* ------------------------------------------------------------------ materials, src/material/mod.rs */
static vec3 random_in_unit_sphere(rng_t* rng) { /* src/vec3.rs:9-25 */
    for (;;) {
        float x = gen_range_f32(rng, -1.0f, 1.0f);
        float y = gen_range_f32(rng, -1.0f, 1.0f);
        float z = gen_range_f32(rng, -1.0f, 1.0f);
        vec3 p = V(x, y, z);
        if (vlen2(p) < 1.0f) return p;
    }
}
static vec3 reflect(vec3 v, vec3 n) { return vsub(v, vscale(2.0f * vdot(v, n), n)); } /* mod.rs:150-152 */
static int refract(vec3 uv, vec3 n, float ratio, vec3* out) {                          /* mod.rs:154-163 */
    float cos_theta = -rust_min(vdot(uv, n), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    if (ratio * sin_theta > 1.0f) return 0;
    vec3 r_out_ortho = vscale(ratio, vadd(uv, vscale(cos_theta, n)));
    vec3 r_out_para = vscale(-sqrtf(fabsf(1.0f - vlen2(r_out_ortho))), n);
    *out = vadd(r_out_ortho, r_out_para);
    return 1;
}
static float powi5(float x) { float x2 = x * x; float x4 = x2 * x2; return x * x4; } /* llvm.powi by squaring */
static float reflectance(float cosine, float ref_idx) {                               /* mod.rs:165-169 */
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    float r0s = r0 * r0;
    return r0s + (1.0f - r0s) * powi5(1.0f - cosine);
}

typedef struct { uint64_t rays; } counters_t;

/* src/ray.rs:78-127, literally recursive */
static vec3 ray_color(const oracle_scene* s, int mode, ray_t ray, uint32_t depth, float t_min, rng_t* rng, counters_t* cnt) {
    if (depth == 0) return V(0.f, 0.f, 0.f);
    hitrec rec;
    cnt->rays++;
    if (world_hit(s, mode, ray, t_min, INFINITY, &rec)) {
        const rrt_material* m = &s->mats[rec.material];
        vec3 rgb = V(m->rgb[0], m->rgb[1], m->rgb[2]);
        vec3 emitted = V(0.f, 0.f, 0.f);
        if (m->kind == RRT_MAT_DIFFUSE_LIGHT) { /* mod.rs:183-194 */
            if (vdot(ray.d, rec.normal) < 0.0f) emitted = rgb;
            return emitted; /* scatter is None */
        }
        if (m->kind == RRT_MAT_METAL) { /* mod.rs:91-108 */
            vec3 reflected = reflect(vnormalize(ray.d), rec.normal);
            ray_t scattered = {rec.p, vadd(reflected, vscale(m->param, random_in_unit_sphere(rng)))};
            if (vdot(scattered.d, rec.normal) > 0.0f)
                return vmul(rgb, ray_color(s, mode, scattered, depth - 1, t_min, rng, cnt)); /* ray.rs:93-98 */
            return emitted;
        }
        if (m->kind == RRT_MAT_DIELECTRIC) { /* mod.rs:114-148 */
            float ratio = 1.0f / m->param;
            vec3 n = rec.normal;
            if (vdot(ray.d, rec.normal) > 0.0f) { ratio = m->param; n = vneg(rec.normal); }
            vec3 unit = vnormalize(ray.d);
            float cos_theta = -rust_min(vdot(unit, n), 1.0f);
            vec3 refracted;
            ray_t scattered = {rec.p, V(0, 0, 0)};
            int done = 0;
            if (refract(unit, n, ratio, &refracted)) {
                if (reflectance(cos_theta, m->param) < gen_f32(rng)) { scattered.d = refracted; done = 1; }
            }
            if (!done) scattered.d = reflect(unit, rec.normal);
            return vmul(V(1.f, 1.f, 1.f), ray_color(s, mode, scattered, depth - 1, t_min, rng, cnt));
        }
        /* Lambertian, mod.rs:71-84, with the mixture of ray.rs:99-120 */
        onb_t uvw = onb_from_w(rec.normal);
        vec3 dir;
        float pdf_val;
        if (s->n_lights != 0) {
            if (gen_bool_half(rng)) dir = onb_local(&uvw, random_cosine_direction(rng)); /* pdf.rs:87-94 */
            else dir = lights_random(s, rec.p, rng);
            pdf_val = 0.5f * cosine_pdf_value(&uvw, dir) + 0.5f * lights_pdf_value(s, rec.p, dir);
        } else {
            dir = onb_local(&uvw, random_cosine_direction(rng));
            pdf_val = cosine_pdf_value(&uvw, dir);
        }
        ray_t scattered = {rec.p, dir};
        pdf_val = rust_max(pdf_val, 1e-5f);
        float cosine = vdot(rec.normal, vnormalize(scattered.d)) / PI_F; /* scattering_pdf, mod.rs:80-83 */
        float s_pdf = rust_max(cosine, 0.0f);
        vec3 next = ray_color(s, mode, scattered, depth - 1, t_min, rng, cnt);
        return vadd(emitted, vdivs(vscale(s_pdf, vmul(rgb, next)), pdf_val));
    }
    return s->background;
}

/* ------------------------------------------------------------------ exports for the known-answer tests */
/* n draws of gen_range(low..high) from the stream (seed, pixel, sample) */
void oracle_gen_range_f32(uint32_t seed, uint32_t pixel, uint32_t sample, float low, float high, uint32_t n, float* out) {
    rng_t r = rng_start(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = gen_range_f32(&r, low, high);
}
/* the value gen_range maps a given 32-bit draw to on its first try (before the `< high` test), and whether it passes */
int oracle_gen_range_map(uint32_t draw, float low, float high, float* out) {
    uint32_t bits = (draw >> 9) | 0x3f800000U;
    float v12;
    memcpy(&v12, &bits, 4);
    *out = (v12 - 1.0f) * (high - low) + low;
    return *out < high;
}
/* OrthNormBasis::from_w, src/vec3.rs:59-68: out = u, v, w */
void oracle_onb(const float* n, float* out) {
    onb_t o = onb_from_w(V(n[0], n[1], n[2]));
    out[0] = o.u.x; out[1] = o.u.y; out[2] = o.u.z; out[3] = o.v.x; out[4] = o.v.y; out[5] = o.v.z; out[6] = o.w.x; out[7] = o.w.y; out[8] = o.w.z;
}
float oracle_reflectance(float cosine, float ref_idx) { return reflectance(cosine, ref_idx); }
int oracle_refract(const float* uv, const float* n, float ratio, float* out) {
    vec3 r = V(0.f, 0.f, 0.f);
    int ok = refract(V(uv[0], uv[1], uv[2]), V(n[0], n[1], n[2]), ratio, &r);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return ok;
}
void oracle_reflect(const float* v, const float* n, float* out) {
    vec3 r = reflect(V(v[0], v[1], v[2]), V(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
/* Hittables::pdf_value of the scene's light list for a direction v from origin */
float oracle_lights_pdf(const oracle_scene* s, const float* origin, const float* v) {
    if (s->n_lights == 0) return 0.0f;
    return lights_pdf_value(s, V(origin[0], origin[1], origin[2]), V(v[0], v[1], v[2]));
}
/* n directions of Hittables::random from origin (stream (seed, pixel, sample)) */
void oracle_lights_random(const oracle_scene* s, const float* origin, uint32_t seed, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; ++i) {
        rng_t r = rng_start(seed, i, 0);
        vec3 d = lights_random(s, V(origin[0], origin[1], origin[2]), &r);
        out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
    }
}
/* n cosine-lobe directions around normal (pdf.rs:14-26,46-48) and the lobe's pdf value for each (pdf.rs:41-44) */
void oracle_cosine_samples(const float* normal, uint32_t seed, uint32_t n, float* dirs, float* pdfs) {
    onb_t uvw = onb_from_w(V(normal[0], normal[1], normal[2]));
    for (uint32_t i = 0; i < n; ++i) {
        rng_t r = rng_start(seed, i, 0);
        vec3 d = onb_local(&uvw, random_cosine_direction(&r));
        dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
        pdfs[i] = cosine_pdf_value(&uvw, d);
    }
}

/* ------------------------------------------------------------------ render loop, src/main.rs:86-103 */
typedef struct {
    const oracle_scene* s; int mode; uint32_t w, h, spp, spp_offset, seed, max_depth; float t_min;
    float* accum; uint32_t* next_row; pthread_mutex_t* lock; uint64_t rays;
} render_job;
static void* render_worker(void* arg) {
    render_job* j = (render_job*)arg;
    counters_t cnt = {0};
    for (;;) {
        pthread_mutex_lock(j->lock);
        uint32_t y = (*j->next_row)++;
        pthread_mutex_unlock(j->lock);
        if (y >= j->h) break;
        for (uint32_t x = 0; x < j->w; ++x) {
            vec3 color = V(0.f, 0.f, 0.f);
            uint32_t pixel = y * j->w + x;
            for (uint32_t sidx = 0; sidx < j->spp; ++sidx) {
                rng_t rng = rng_start(j->seed, pixel, j->spp_offset + sidx);
                float u = (float)x + gen_f32(&rng);
                float v = (float)y + gen_f32(&rng);
                ray_t ray = camera_get_ray(&j->s->cam, u, v, j->w, j->h);
                rng.ctr++; /* camera.rs:67: the ray time draw */
                color = vadd(color, ray_color(j->s, j->mode, ray, j->max_depth, j->t_min, &rng, &cnt));
            }
            float* px = j->accum + 3 * (size_t)pixel;
            px[0] = color.x; px[1] = color.y; px[2] = color.z;
        }
    }
    j->rays = cnt.rays;
    return NULL;
}
/* Writes the per-pixel SUM over samples [spp_offset, spp_offset+spp), rows bottom-up like y in main.rs:86. */
uint64_t oracle_render(const oracle_scene* s, int mode, uint32_t w, uint32_t h, uint32_t spp, uint32_t spp_offset, uint32_t seed,
                       uint32_t max_depth, float t_min, float* accum, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    render_job* jobs = (render_job*)malloc(sizeof(render_job) * (size_t)threads);
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    uint32_t next_row = 0;
    for (int t = 0; t < threads; ++t) {
        render_job j = {s, mode, w, h, spp, spp_offset, seed, max_depth, t_min, accum, &next_row, &lock, 0};
        jobs[t] = j;
        pthread_create(&th[t], NULL, render_worker, &jobs[t]);
    }
    uint64_t rays = 0;
    for (int t = 0; t < threads; ++t) { pthread_join(th[t], NULL); rays += jobs[t].rays; }
    free(th); free(jobs);
    return rays;
}

/* ------------------------------------------------------------------ tone map, src/vec3.rs:30-53 + main.rs:104-109 */
static uint8_t rust_as_u8(float x) { return isnan(x) ? 0 : (x <= 0.0f ? 0 : (x >= 255.0f ? 255 : (uint8_t)x)); }
static float rust_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); } /* NaN stays NaN */
void oracle_tonemap(const float* accum, uint32_t w, uint32_t h, uint32_t spp, uint8_t* rgb8) {
    float scale = 1.0f / (float)spp;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const float* c = accum + 3 * ((size_t)y * w + x);
            uint8_t* o = rgb8 + 3 * ((size_t)(h - 1 - y) * w + x);
            for (int k = 0; k < 3; ++k) {
                float g = isnan(c[k]) ? 0.0f : sqrtf(c[k] * scale);
                o[k] = rust_as_u8(256.0f * rust_clamp(g, 0.0f, 0.999f));
            }
        }
}
