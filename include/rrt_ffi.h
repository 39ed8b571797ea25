/*
 * rrt_ffi.h -- C ABI of the MI355X (gfx950) path-tracing library `librrt_hip.so`.
 *
 * This is the drop-in boundary for the per-pixel hot path of Safarte/rust-ray-tracer:
 * everything between "a Scene has been built" (reference src/scene.rs:25-30) and
 * "an RGB8 pixel buffer exists" (reference src/main.rs:84-116).  A Rust `extern "C"`
 * crate (or any other FFI) can bind this header verbatim: plain pointers, sizes and
 * integer status codes only; no exceptions, panics or aborts cross the boundary.
 *
 * Conventions
 *   - every entry point returns rrt_status (0 == success) unless noted;
 *     rrt_last_error(ctx) gives a human-readable message for the last failure on ctx
 *     (ctx may be NULL: then the last error of the calling thread outside any ctx).
 *   - "host" pointers are ordinary process memory, copied during the call (the caller
 *     may free them on return).  "device" pointers (d_ prefix) are HIP device memory
 *     on the ctx's GPU; calls taking them are asynchronous on the given hipStream_t
 *     (passed as void*; NULL == the default stream).
 *   - one ctx per host thread, or external serialisation; one ctx drives one GPU (rrt_group: several).
 *   - at most one device call should be in flight per ctx: a device call issued on another stream than the
 *     previous one waits (on the GPU) for that one, because both use the ctx's counters and workspace.
 *   - all floating point data is IEEE binary32; ids are 32-bit.
 *
 * Primitive ids.  The world is an ordered list of object runs (rrt_object).  Every
 * primitive gets the id of its position in that flattened list, exactly the index it
 * would have in the reference's world `Vec<Arc<dyn Hittable>>` in push order
 * (reference src/gltf.rs:215-240,333-337); a cuboid contributes its 6 rectangles in
 * the order of reference src/geometry/cuboid.rs:24-73, an instance contributes the
 * primitives of its mesh in mesh order.  The closest hit of a ray is the minimum of
 * (t, id) over all primitives accepted by the reference's per-primitive tests.
 */
#ifndef RRT_FFI_H
#define RRT_FFI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_ABI_VERSION 3u  /* 3: rrt_tuning + rrt_debug_{get,set}_tuning, RRT_CFG_GROUP_FORCE_RCCL; 2: rrt_scene_info grew */

typedef enum rrt_status {
    RRT_OK = 0,
    RRT_ERR_INVALID = 1,   /* bad argument (NULL, out-of-range index, zero size ...) */
    RRT_ERR_NO_DEVICE = 2, /* no usable gfx950 device / HIP runtime not functional    */
    RRT_ERR_HIP = 3,       /* a HIP call failed; message carries hipGetErrorString    */
    RRT_ERR_OOM = 4,       /* host or device allocation failed                        */
    RRT_ERR_IO = 5,        /* file could not be read / written                        */
    RRT_ERR_PARSE = 6,     /* glTF / JSON / base64 content not understood             */
    RRT_ERR_STATE = 7      /* call order wrong (e.g. render before scene upload)      */
} rrt_status;

typedef struct rrt_ctx rrt_ctx; /* opaque; owns all device memory of one GPU */

/* ---- configuration (reference src/main.rs:56 MAX_DEPTH, src/ray.rs:89 t_min) ---- */
typedef struct rrt_config {
    int32_t device;     /* HIP device ordinal                                        */
    uint32_t seed;      /* key of the counter-based per-pixel RNG                    */
    uint32_t max_depth; /* reference: 4; 1..64                                       */
    float t_min;        /* reference: 1e-4                                           */
    uint32_t flags;     /* RRT_CFG_* bits                                            */
} rrt_config;
#define RRT_CFG_DEFAULT 0u
#define RRT_CFG_COUNT_VISITS 1u /* also count BVH node visits / primitive tests (slower kernel variant) */
#define RRT_CFG_DEVICE_BUILD 2u /* build large BVHs (>= 4096 primitives) on the GPU: a linear BVH over Morton-sorted
                                 * primitives instead of the host's SAH build (reference src/geometry/mod.rs:161-199 is
                                 * the step it replaces).  Faster to build, slower to traverse; results are identical. */
#define RRT_CFG_NO_UNROLL 4u    /* keep instanced meshes as a two-level BVH.  Default: when every instanced mesh holds
                                 * triangles only and the copies fit a node budget, each instance (reference
                                 * src/geometry/transform.rs:9-12,44-49) is written out as a world-space copy of its mesh's
                                 * tree inside the one top-level BVH; the ray changes space at the triangle test only,
                                 * on the mesh's shared object-space records.  Same hits, faster walk, more node memory. */
#define RRT_CFG_GROUP_FORCE_RCCL 8u /* rrt_group_create only: sum the members' buffers with RCCL whenever the listed devices
                                 * are distinct -- also in a group of ONE member, whose reduce is otherwise skipped.  Lets a
                                 * one-GPU machine run every RCCL call of the multi-GPU path (communicator set-up, grouped
                                 * in-place ncclReduce to member 0, stream ordering); the image is the member's own.        */

/* ---- scene description: the reference's scene/material types as POD ---- */

/* reference src/material/mod.rs:47-194 (Lambertian / Metal / Dielectric / DiffuseLight) */
enum { RRT_MAT_LAMBERTIAN = 0, RRT_MAT_METAL = 1, RRT_MAT_DIELECTRIC = 2, RRT_MAT_DIFFUSE_LIGHT = 3 };
typedef struct rrt_material {
    uint32_t kind;
    float rgb[3]; /* albedo (SolidColor), metal albedo, or emitted radiance; ignored for dielectric */
    float param;  /* metal: fuzziness; dielectric: index of refraction; else unused                 */
} rrt_material;

/* reference src/geometry/triangle.rs:13-33 -- the library derives v0v1, v0v2 and the padded AABB itself */
typedef struct rrt_triangle {
    float v0[3], v1[3], v2[3];
    uint32_t material;
} rrt_triangle;

/* reference src/geometry/sphere.rs:17-32 */
typedef struct rrt_sphere {
    float center[3];
    float radius;
    uint32_t material;
} rrt_sphere;

/* reference src/geometry/aarect.rs:14-34,69-89,145-167.  axis = the axis the plane is perpendicular to:
 * 0: YZRect (a=y, b=z, k on x)   1: XZRect (a=x, b=z, k on y)   2: XYRect (a=x, b=y, k on z)          */
typedef struct rrt_rect {
    uint32_t axis;
    float a0, a1, b0, b1, k;
    uint32_t material;
    uint32_t flags; /* RRT_RECT_FLIP_FACE: normal.y = -|normal.y| (reference src/geometry/mod.rs:222-234) */
} rrt_rect;
#define RRT_RECT_FLIP_FACE 1u

/* reference src/geometry/cuboid.rs:16-77 -- expands to 6 rrt_rect in the reference's order */
typedef struct rrt_cuboid {
    float min[3], max[3];
    uint32_t material;
} rrt_cuboid;

/* One entry of the ordered world list (or of a mesh's object list). */
enum { RRT_OBJ_TRIANGLES = 0, RRT_OBJ_SPHERES = 1, RRT_OBJ_RECTS = 2, RRT_OBJ_CUBOIDS = 3, RRT_OBJ_INSTANCE = 4 };
typedef struct rrt_object {
    uint32_t type;  /* RRT_OBJ_*                                                          */
    uint32_t first; /* index into the typed array (triangles / spheres / rects / cuboids / instances) */
    uint32_t count; /* run length (must be 1 for RRT_OBJ_INSTANCE)                        */
} rrt_object;

/* A mesh = an object list of its own with its own BVH (bottom level), to be instanced.
 * Mirrors `BVHNode::new(objects, ..)` used as the base of a transform wrapper
 * (reference src/scene.rs:385-388).  Meshes may not contain instances.                   */
typedef struct rrt_mesh {
    uint32_t first_object; /* range in rrt_scene_desc.mesh_objects */
    uint32_t object_count;
} rrt_mesh;

/* Translate(RotateY(mesh)) -- reference src/geometry/transform.rs:9-42,44-135 */
typedef struct rrt_instance {
    uint32_t mesh;
    float sin_theta, cos_theta; /* RotateY: sin/cos of the angle (identity: 0, 1) */
    float offset[3];            /* Translate offset                                */
} rrt_instance;

/* Light list entry (reference `Scene.lights`, src/scene.rs:27; sampled by src/pdf.rs:51-70).
 * Only spheres and XZ rectangles implement pdf_value/random in the reference
 * (src/geometry/sphere.rs:101-119, src/geometry/aarect.rs:123-142); RRT_LIGHT_OTHER keeps the
 * trait defaults (pdf 0, direction (1,0,0); src/geometry/mod.rs:56-61).                   */
enum { RRT_LIGHT_SPHERE = 0, RRT_LIGHT_XZRECT = 1, RRT_LIGHT_OTHER = 2 };
typedef struct rrt_light {
    uint32_t type;
    float p[5]; /* sphere: cx,cy,cz,r,-   xzrect: x0,x1,z0,z1,k */
} rrt_light;

/* reference src/camera.rs:6-16 after Camera::new / apply_transform */
typedef struct rrt_camera {
    float c2w[12];   /* camera_to_world: 3 columns of the 3x3 part (x_axis, y_axis, z_axis), then translation */
    float origin[3]; /* ray_origin                                                                            */
    float scale;     /* tan(vertical_fov/2)                                                                   */
    float aspect;    /* the CAMERA's aspect ratio (not the CLI override, reference src/main.rs:76-81)         */
    float time0, time1;
} rrt_camera;

typedef struct rrt_scene_desc {
    const rrt_material* materials;  uint32_t material_count;
    const rrt_triangle* triangles;  uint32_t triangle_count;
    const rrt_sphere* spheres;      uint32_t sphere_count;
    const rrt_rect* rects;          uint32_t rect_count;
    const rrt_cuboid* cuboids;      uint32_t cuboid_count;
    const rrt_instance* instances;  uint32_t instance_count;
    const rrt_mesh* meshes;         uint32_t mesh_count;
    const rrt_object* mesh_objects; uint32_t mesh_object_count;
    const rrt_object* world;        uint32_t world_count;  /* ordered world list: defines primitive ids */
    const rrt_light* lights;        uint32_t light_count;
    rrt_camera camera;
    float background[3];
} rrt_scene_desc;

/* ---- ray batches (parity hook) ---- */
typedef struct rrt_ray { /* reference src/ray.rs:14-18; direction is NOT normalised */
    float origin[3];
    float direction[3];
    float time;
} rrt_ray;

typedef struct rrt_hit {
    int32_t prim_id; /* -1: miss */
    float t;         /* ray parameter of the hit (undefined on miss: written as +inf) */
    float u, v;      /* HitRecord.u/.v: triangle barycentrics, rectangle uv; 0 for spheres */
} rrt_hit;

typedef struct rrt_stats {
    uint64_t rays;        /* closest-hit queries issued by the integrator (one per reference src/ray.rs:89) */
    uint64_t samples;     /* iterations of the spp loop, reference src/main.rs:91                           */
    uint64_t node_visits; /* BVH nodes fetched   (only with RRT_CFG_COUNT_VISITS, else 0)                   */
    uint64_t prim_tests;  /* primitive tests run (only with RRT_CFG_COUNT_VISITS, else 0)                   */
    /* Divergence of the render kernel (only with RRT_CFG_COUNT_VISITS): how often a 64-lane wavefront executed
     * the node step / the primitive test / the shading pass.  node_visits / (64 * node_wave_steps) is the
     * fraction of lanes doing useful work in a node step, and likewise for the other two.               */
    uint64_t node_wave_steps;
    uint64_t prim_wave_steps;
    uint64_t shade_wave_passes;
    uint64_t refill_wave_passes; /* ray batches only: passes in which a wavefront's idle lanes drew new rays  */
    /* Render kernel: lane slots of node steps that went to lanes whose walk was already complete / to lanes
     * parked at a leaf when the node pass began (the rest of the idle slots belongs to lanes that reached a
     * leaf during the pass).  Same units as 64 * node_wave_steps.                                            */
    uint64_t node_slots_lane_done;
    uint64_t node_slots_lane_at_leaf;
    float kernel_ms;      /* HIP-event time of the render kernel on its stream                              */
    uint32_t chunks;      /* sample chunks the render was cut into (0 for a ray batch); a pixel's sum is the sum,
                             in chunk order, of its chunks' sums                                            */
} rrt_stats;

typedef struct rrt_scene_info {
    uint32_t prim_count;     /* number of primitive ids                                   */
    uint32_t node_count;     /* BVH nodes in HBM (all levels)                             */
    uint32_t tri_count;      /* triangle records in HBM                                   */
    uint32_t bvh_depth;      /* deepest leaf (top level + bottom level for instances)     */
    uint64_t device_bytes;   /* bytes of HBM the scene occupies                           */
    float build_ms;          /* BVH build (host, plus the GPU builder where it ran)       */
    float upload_ms;
    uint32_t device_built;   /* number of BVHs the GPU builder built                      */
    uint32_t unrolled_instances; /* instances written out into the top-level BVH (0: none, or RRT_CFG_NO_UNROLL) */
    uint32_t scaled_slab;    /* 1: rrt_render walks this scene with the scaled form of the box arithmetic (one grid,
                                camera ray directions between 2^-24 and 2^24 long); 0: with the convert form.  Same images. */
    uint32_t pixel_list_permille; /* of the last render whose statistics were read and that collected per-pixel leaf
                                lists for its primary rays: the share of the frame's pixels that got one, in 1/1000
                                (0 before such a render).  The library uses it to decide whether collecting them pays. */
} rrt_scene_info;

/* ---- scheduling knobs (tests and experiments) ----  no counterpart in the reference.  How the render and ray-batch
 * kernels cut a frame into work items and order a wavefront's passes.  None of the knobs changes which paths a pixel
 * follows; `chunks` changes the order in which a pixel's partial sums are added (float rounding), every other knob
 * leaves images identical bit for bit.  RRT_TUNE_AUTO in a field = the library's own rule for the scene and frame.
 * rrt_create reads the environment ONCE (RRT_CHUNKS, RRT_SPLIT_ITEMS, RRT_NODE_SHARE, RRT_LEAVE_WALK, RRT_LEAVE_READY,
 * RRT_OVERLAP_ITEMS, RRT_BAND_TILES, RRT_SCALED_SLAB, RRT_REFILL_MIN, RRT_BLOCKS_PER_CU, RRT_INTERSECT_STATIC, RRT_PIXEL_LISTS,
 * RRT_PIXEL_LIST_LEN) into the
 * context's rrt_tuning; the render and intersect entry points never look at the environment.                      */
#define RRT_TUNE_AUTO 0xFFFFFFFFu
typedef struct rrt_tuning {
    uint32_t chunks;           /* sample chunks a frame is cut into (1 .. spp); each chunk has a partial image        */
    uint32_t split_items;      /* 1: work item = a few pixels x many of their chunks; 0: 64 pixels x one chunk          */
    uint32_t node_share_64ths; /* 0..64: share of the walking lanes that must descend for a node pass to run            */
    uint32_t leave_walk;       /* 0..63: a walk phase ends with at most this many lanes still walking ...               */
    uint32_t leave_ready;      /* 1..64: ... once at least this many wait to shade                                      */
    uint32_t overlap_items;    /* 1: two work items in flight per wavefront; 0: one                                     */
    uint32_t band_tiles;       /* tiles are numbered in vertical bands this many 8-pixel tiles wide; 0: row-major       */
    uint32_t scaled_slab;      /* 0: convert form of the box arithmetic even where the scaled form is valid             */
    uint32_t refill_min;       /* 1..64: idle lanes that make the ray-batch kernel draw new rays                        */
    uint32_t blocks_per_cu;    /* resident workgroups per CU of the persistent kernels (at most what the kernel allows) */
    uint32_t intersect_static; /* 1: rrt_intersect runs the plain one-lane-per-ray kernel                               */
    uint32_t pixel_lists;      /* 1: primary rays of split items resolved against per-pixel candidate lists; 0: they
                                  walk the tree                                                                        */
    uint32_t pixel_list_len;   /* 1..11: a pixel whose camera rays can reach more primitives than this gets no list     */
    uint32_t list_waves;       /* 6 or 7: wavefronts per SIMD the render kernel with pixel lists is compiled for (84 or
                                  72 registers).  RRT_TUNE_AUTO: the library times both on the scene's first launches of
                                  a frame size and sample count whose statistics are read, and keeps the faster one     */
    uint32_t reserved[2];      /* RRT_TUNE_AUTO                                                                         */
} rrt_tuning;

/* ---- lifetime ----  replaces: process start-up in reference src/main.rs:26-56 */
uint32_t rrt_abi_version(void);
rrt_status rrt_create(const rrt_config* cfg, rrt_ctx** out);
void rrt_destroy(rrt_ctx* ctx);
const char* rrt_last_error(const rrt_ctx* ctx);
/* The context's scheduling knobs (see rrt_tuning).  set: takes effect from the next render / intersect call; fields
 * out of range are clamped.  Not for use while a call on the context is in flight.                                */
rrt_status rrt_debug_get_tuning(const rrt_ctx* ctx, rrt_tuning* out);
rrt_status rrt_debug_set_tuning(rrt_ctx* ctx, const rrt_tuning* tuning);

/* ---- scene upload ----  replaces: `BVHNode::new(objects, 0., 1.)` (reference src/gltf.rs:343,
 * src/geometry/mod.rs:161-199): the library builds a flat structure-of-arrays BVH on the host
 * and uploads it, the primitives, materials, lights and camera to HBM.                       */
rrt_status rrt_scene_upload(rrt_ctx* ctx, const rrt_scene_desc* scene);
rrt_status rrt_scene_get_info(const rrt_ctx* ctx, rrt_scene_info* out);
/* Host-only dry run of the same validation + BVH build (no GPU needed, nothing uploaded): fills `out`
 * (device_bytes = what an upload would occupy, upload_ms = 0).  Errors are reported through
 * rrt_last_error(NULL).                                                                         */
rrt_status rrt_scene_check(const rrt_scene_desc* scene, rrt_scene_info* out);

/* ---- closest hit on a ray batch ----  replaces: `world.hit(ray, t_min, t_max)`
 * (reference src/ray.rs:89 -> src/geometry/mod.rs:136-153) for n rays at once.               */
rrt_status rrt_intersect(rrt_ctx* ctx, const rrt_ray* rays, size_t n, float t_min, float t_max, rrt_hit* out);
rrt_status rrt_intersect_device(rrt_ctx* ctx, const rrt_ray* d_rays, size_t n, float t_min, float t_max,
                                rrt_hit* d_out, void* stream);

/* ---- render ----  replaces: the pixel loop of reference src/main.rs:86-103 (rows x columns x
 * samples; camera.get_ray + ray_color, summed per pixel).  Sample indices
 * [spp_offset, spp_offset + spp) of every pixel are rendered and their SUM is written to
 * accum (w*h*3 floats, row y as in the reference's loop, i.e. bottom-up; x fastest, RGB
 * interleaved).  Disjoint sample ranges on different GPUs add up to the full image.          */
rrt_status rrt_render(rrt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                      float* accum, rrt_stats* stats);
rrt_status rrt_render_device(rrt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                             float* d_accum, void* stream);
/* The same render for callers that cannot allocate.  No counterpart in the reference: it exists for hosts that may not
 * call hipMalloc while they render (code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation).  It launches the same kernels as rrt_render_device and gives bit-identical results, but its
 * counters, work-queue heads and per-chunk partial images live in a static workspace in device memory
 * (256 bytes + RRT_NOALLOC_PARTIAL_MAX_BYTES, held by every GPU the library runs on until the process ends) instead of
 * buffers the context grows on demand; the call allocates and frees nothing (no device memory; on the success path no
 * host memory either).  A frame whose partial images (chunks x width x height x 12 bytes; a frame is cut into up to 32
 * sample chunks) exceed RRT_NOALLOC_PARTIAL_MAX_BYTES is cut into fewer, longer chunks until they fit -- the image then
 * equals rrt_render_device's with the same chunk count (rrt_stats.chunks reports it; another count adds the same samples
 * in another order) -- and a frame that needs chunks but of which not even two partial images fit (more than
 * RRT_NOALLOC_PARTIAL_MAX_BYTES / 24 = 2,796,202 pixels) is refused with RRT_ERR_INVALID.  All contexts on one GPU share
 * that GPU's workspace: calls that overlap on one GPU race (also with rrt_render_stats of an earlier such call still
 * unread).                                                                                                         */
#define RRT_NOALLOC_PARTIAL_MAX_BYTES (64u << 20)
rrt_status rrt_render_device_noalloc(rrt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t spp_offset,
                                     float* d_accum, void* stream);
/* Blocks until the stream of the last rrt_render_device or rrt_intersect_device call has drained, then reports
 * its counters (for a ray batch: rays = n, samples = 0). */
rrt_status rrt_render_stats(rrt_ctx* ctx, rrt_stats* stats);

/* ---- multi-GPU ----  replaces: the rayon pool over image rows (reference src/main.rs:42-49,86), the reference's
 * only parallelism, across the GPUs of one node.  A group is one context per listed device inside one host process.
 * Every sample is independent (reference src/main.rs:91), so rrt_group_render gives member g a contiguous slice of
 * the sample indices [spp_offset, spp_offset + spp_total) of every pixel (remainder to the low members), renders the
 * slices side by side, sums the f32 accumulation buffers on member 0 (RCCL ncclReduce over xGMI when the devices
 * are distinct; librccl is opened at run time, only then; a copy plus an add kernel when devices repeat or RCCL is
 * unavailable) and copies the sum to `accum` (host, layout as rrt_render).  The scene is built once on the host and
 * uploaded to every member.  Listing one device twice is allowed (two members share the GPU): a rehearsal of the
 * partition on a one-GPU machine, not a speed-up.  stats: counters summed over members, kernel_ms = the slowest.   */
typedef struct rrt_group rrt_group;
rrt_status rrt_group_create(const rrt_config* cfg /* .device ignored */, const int32_t* devices, uint32_t n_devices,
                            rrt_group** out);
void rrt_group_destroy(rrt_group* group);
const char* rrt_group_last_error(const rrt_group* group);
uint32_t rrt_group_size(const rrt_group* group);
const char* rrt_group_reduce_path(const rrt_group* group); /* "none" (one member), "rccl" or "copy+add" */
rrt_status rrt_group_scene_upload(rrt_group* group, const rrt_scene_desc* scene);
rrt_status rrt_group_scene_get_info(const rrt_group* group, rrt_scene_info* out);
rrt_status rrt_group_render(rrt_group* group, uint32_t width, uint32_t height, uint32_t spp_total, uint32_t spp_offset,
                            float* accum, rrt_stats* stats);
rrt_status rrt_group_tonemap(rrt_group* group, const float* accum, uint32_t width, uint32_t height, uint32_t spp_total,
                             uint8_t* rgb8);

/* ---- tone map ----  replaces: `get_color(color, samples)` + the flipped `put_pixel`
 * (reference src/vec3.rs:30-53, src/main.rs:104-109).  rgb8 is top-down, w*h*3 bytes.        */
rrt_status rrt_tonemap(rrt_ctx* ctx, const float* accum, uint32_t width, uint32_t height, uint32_t spp_total,
                       uint8_t* rgb8);
rrt_status rrt_tonemap_device(rrt_ctx* ctx, const float* d_accum, uint32_t width, uint32_t height,
                              uint32_t spp_total, uint8_t* d_rgb8, void* stream);

/* ---- arithmetic self-test ----  no counterpart in the reference.  The kernels form 1/x and sqrt(x) in the primitive
 * tests and the light pdf (reference src/geometry/triangle.rs:46, src/geometry/sphere.rs:61,107-108) with a short
 * instruction sequence that must round exactly like the IEEE operations the reference's f32 arithmetic uses.  This
 * call runs that sequence and the IEEE operation over ALL 2^32 float bit patterns on the ctx's GPU and reports the
 * number of patterns on which they differ in any bit (both must be 0).                                             */
rrt_status rrt_selftest_arith(rrt_ctx* ctx, uint64_t* recip_mismatches, uint64_t* sqrt_mismatches);

/* ---- front end helpers (host only; not accelerated) ---- */

/* Scene construction from glTF -- replaces `Scene::from_gltf_file` (reference src/gltf.rs:314-351).
 * flags == RRT_GLTF_COMPAT reproduces the reference's contract: base64 data-URI buffers only, u16
 * indices, primitives[0] only, accessor byteOffset / byteStride ignored, mesh-node transforms ignored,
 * camera and lights taken from `camera_light_path` (the reference hard-codes "assets/suzanne.gltf",
 * src/gltf.rs:317; NULL means "the same file as `path`"), lights become radius-0.2 emissive spheres,
 * |metallic| < 1e-5 -> Lambertian else Metal(fuzz = roughness), background 0.051.
 * RRT_GLTF_EXTENDED additionally accepts external .bin buffers, accessor byteOffset / byteStride,
 * u8/u32 indices, bakes node TRS / matrix transforms into the vertices and applies `extended_scale`. */
#define RRT_GLTF_COMPAT 0u
#define RRT_GLTF_EXTENDED 1u
typedef struct rrt_gltf_options {
    uint32_t flags;
    float extended_scale;     /* uniform scale applied to baked vertices (EXTENDED only; 0 == 1.0) */
    float extended_offset[3]; /* translation applied after the scale (EXTENDED only)                */
} rrt_gltf_options;
rrt_status rrt_gltf_load(const char* path, const char* camera_light_path, const rrt_gltf_options* opt,
                         rrt_scene_desc** out);
void rrt_scene_free(rrt_scene_desc* scene); /* frees a scene returned by rrt_gltf_load */

/* Image output -- replaces `img.save(output_file)` (reference src/main.rs:116); unlike the reference,
 * a failure to write is reported.  rgb8 top-down.                                                  */
rrt_status rrt_png_write(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RRT_FFI_H */
