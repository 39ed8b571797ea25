"""Host-side Python binding of ``librrt_hip.so`` (the C ABI declared in ``include/rrt_ffi.h``).

The classes mirror the reference's scene description types for the path-tracing hot path:
``Scene`` (reference ``src/scene.rs:25-30``), ``Camera`` (``src/camera.rs:6-16``), the four materials
(``src/material/mod.rs:47-194``) and the primitives (``src/geometry``).  Everything that computes runs in
the HIP library; there is no CPU fallback -- importing works without a GPU, but creating a ``Renderer``
raises when the library or a gfx950 device is missing.

The directory name contains a hyphen, so the package is imported through ``rrt_pkg.load()`` at the
repository root under the module name ``rust_ray_tracer_amd``.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RRT_LIB_PATH: another build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("RRT_LIB_PATH") or os.path.join(_HERE, "librrt_hip.so")

RRT_OK = 0
STATUS_NAMES = {0: "RRT_OK", 1: "RRT_ERR_INVALID", 2: "RRT_ERR_NO_DEVICE", 3: "RRT_ERR_HIP", 4: "RRT_ERR_OOM",
                5: "RRT_ERR_IO", 6: "RRT_ERR_PARSE", 7: "RRT_ERR_STATE"}

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT = 0, 1, 2, 3
OBJ_TRIANGLES, OBJ_SPHERES, OBJ_RECTS, OBJ_CUBOIDS, OBJ_INSTANCE = 0, 1, 2, 3, 4
LIGHT_SPHERE, LIGHT_XZRECT, LIGHT_OTHER = 0, 1, 2
RECT_FLIP_FACE = 1
CFG_COUNT_VISITS = 1
CFG_DEVICE_BUILD = 2
CFG_NO_UNROLL = 4
CFG_GROUP_FORCE_RCCL = 8
TUNE_AUTO = 0xFFFFFFFF
GLTF_COMPAT, GLTF_EXTENDED = 0, 1


class RrtError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


# ---------------------------------------------------------------------------------------------- POD
class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("seed", C.c_uint32), ("max_depth", C.c_uint32), ("t_min", C.c_float),
                ("flags", C.c_uint32)]


class MaterialPOD(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rgb", C.c_float * 3), ("param", C.c_float)]


class TrianglePOD(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3), ("material", C.c_uint32)]


class SpherePOD(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", C.c_uint32)]


class RectPOD(C.Structure):
    _fields_ = [("axis", C.c_uint32), ("a0", C.c_float), ("a1", C.c_float), ("b0", C.c_float), ("b1", C.c_float),
                ("k", C.c_float), ("material", C.c_uint32), ("flags", C.c_uint32)]


class CuboidPOD(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3), ("material", C.c_uint32)]


class ObjectPOD(C.Structure):
    _fields_ = [("type", C.c_uint32), ("first", C.c_uint32), ("count", C.c_uint32)]


class MeshPOD(C.Structure):
    _fields_ = [("first_object", C.c_uint32), ("object_count", C.c_uint32)]


class InstancePOD(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("sin_theta", C.c_float), ("cos_theta", C.c_float), ("offset", C.c_float * 3)]


class LightPOD(C.Structure):
    _fields_ = [("type", C.c_uint32), ("p", C.c_float * 5)]


class CameraPOD(C.Structure):
    _fields_ = [("c2w", C.c_float * 12), ("origin", C.c_float * 3), ("scale", C.c_float), ("aspect", C.c_float),
                ("time0", C.c_float), ("time1", C.c_float)]


class SceneDescPOD(C.Structure):
    _fields_ = [
        ("materials", C.POINTER(MaterialPOD)), ("material_count", C.c_uint32),
        ("triangles", C.POINTER(TrianglePOD)), ("triangle_count", C.c_uint32),
        ("spheres", C.POINTER(SpherePOD)), ("sphere_count", C.c_uint32),
        ("rects", C.POINTER(RectPOD)), ("rect_count", C.c_uint32),
        ("cuboids", C.POINTER(CuboidPOD)), ("cuboid_count", C.c_uint32),
        ("instances", C.POINTER(InstancePOD)), ("instance_count", C.c_uint32),
        ("meshes", C.POINTER(MeshPOD)), ("mesh_count", C.c_uint32),
        ("mesh_objects", C.POINTER(ObjectPOD)), ("mesh_object_count", C.c_uint32),
        ("world", C.POINTER(ObjectPOD)), ("world_count", C.c_uint32),
        ("lights", C.POINTER(LightPOD)), ("light_count", C.c_uint32),
        ("camera", CameraPOD),
        ("background", C.c_float * 3),
    ]


class RayPOD(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("time", C.c_float)]


class HitPOD(C.Structure):
    _fields_ = [("prim_id", C.c_int32), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64),
                ("node_wave_steps", C.c_uint64), ("prim_wave_steps", C.c_uint64), ("shade_wave_passes", C.c_uint64),
                ("refill_wave_passes", C.c_uint64), ("node_slots_lane_done", C.c_uint64),
                ("node_slots_lane_at_leaf", C.c_uint64),
                ("kernel_ms", C.c_float), ("chunks", C.c_uint32)]


class Tuning(C.Structure):
    """``rrt_tuning``: the scheduling knobs of a context (``TUNE_AUTO`` = the library's rule)."""
    _fields_ = [("chunks", C.c_uint32), ("split_items", C.c_uint32), ("node_share_64ths", C.c_uint32),
                ("leave_walk", C.c_uint32), ("leave_ready", C.c_uint32), ("overlap_items", C.c_uint32),
                ("band_tiles", C.c_uint32), ("scaled_slab", C.c_uint32), ("refill_min", C.c_uint32),
                ("blocks_per_cu", C.c_uint32), ("intersect_static", C.c_uint32), ("pixel_lists", C.c_uint32),
                ("pixel_list_len", C.c_uint32), ("list_waves", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class SceneInfo(C.Structure):
    _fields_ = [("prim_count", C.c_uint32), ("node_count", C.c_uint32), ("tri_count", C.c_uint32),
                ("bvh_depth", C.c_uint32), ("device_bytes", C.c_uint64), ("build_ms", C.c_float),
                ("upload_ms", C.c_float), ("device_built", C.c_uint32), ("unrolled_instances", C.c_uint32),
                ("scaled_slab", C.c_uint32), ("pixel_list_permille", C.c_uint32)]


class GltfOptions(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("extended_scale", C.c_float), ("extended_offset", C.c_float * 3)]


RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("direction", np.float32, 3), ("time", np.float32)])
HIT_DTYPE = np.dtype([("prim_id", np.int32), ("t", np.float32), ("u", np.float32), ("v", np.float32)])
TRI_DTYPE = np.dtype([("v0", np.float32, 3), ("v1", np.float32, 3), ("v2", np.float32, 3), ("material", np.uint32)])
assert RAY_DTYPE.itemsize == C.sizeof(RayPOD) and HIT_DTYPE.itemsize == C.sizeof(HitPOD)
assert TRI_DTYPE.itemsize == C.sizeof(TrianglePOD)

# every symbol include/rrt_ffi.h declares
ABI_SYMBOLS = [
    "rrt_abi_version", "rrt_create", "rrt_destroy", "rrt_last_error", "rrt_debug_get_tuning", "rrt_debug_set_tuning",
    "rrt_scene_upload", "rrt_scene_get_info",
    "rrt_scene_check",
    "rrt_intersect", "rrt_intersect_device", "rrt_render", "rrt_render_device", "rrt_render_stats", "rrt_tonemap",
    "rrt_tonemap_device", "rrt_selftest_arith", "rrt_gltf_load", "rrt_scene_free", "rrt_png_write",
    "rrt_render_device_noalloc",
    "rrt_group_create", "rrt_group_destroy", "rrt_group_last_error", "rrt_group_size", "rrt_group_reduce_path",
    "rrt_group_scene_upload", "rrt_group_scene_get_info", "rrt_group_render", "rrt_group_tonemap",
]
NOALLOC_PARTIAL_MAX_BYTES = 64 << 20  # RRT_NOALLOC_PARTIAL_MAX_BYTES

_lib = None


def load_library() -> C.CDLL:
    """Loads ``librrt_hip.so`` (built in-tree by ``make``) and declares the prototypes.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
    lib.rrt_abi_version.restype = u32
    lib.rrt_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.rrt_destroy.argtypes = [vp]
    lib.rrt_destroy.restype = None
    lib.rrt_last_error.argtypes = [vp]
    lib.rrt_last_error.restype = C.c_char_p
    lib.rrt_debug_get_tuning.argtypes = [vp, C.POINTER(Tuning)]
    lib.rrt_debug_set_tuning.argtypes = [vp, C.POINTER(Tuning)]
    lib.rrt_scene_upload.argtypes = [vp, C.POINTER(SceneDescPOD)]
    lib.rrt_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
    lib.rrt_scene_check.argtypes = [C.POINTER(SceneDescPOD), C.POINTER(SceneInfo)]
    lib.rrt_intersect.argtypes = [vp, vp, C.c_size_t, f32, f32, vp]
    lib.rrt_intersect_device.argtypes = [vp, vp, C.c_size_t, f32, f32, vp, vp]
    lib.rrt_render.argtypes = [vp, u32, u32, u32, u32, vp, C.POINTER(Stats)]
    lib.rrt_render_device.argtypes = [vp, u32, u32, u32, u32, vp, vp]
    lib.rrt_render_device_noalloc.argtypes = [vp, u32, u32, u32, u32, vp, vp]
    lib.rrt_render_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.rrt_group_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int32), u32, C.POINTER(vp)]
    lib.rrt_group_destroy.argtypes = [vp]
    lib.rrt_group_destroy.restype = None
    lib.rrt_group_last_error.argtypes = [vp]
    lib.rrt_group_last_error.restype = C.c_char_p
    lib.rrt_group_size.argtypes = [vp]
    lib.rrt_group_size.restype = u32
    lib.rrt_group_reduce_path.argtypes = [vp]
    lib.rrt_group_reduce_path.restype = C.c_char_p
    lib.rrt_group_scene_upload.argtypes = [vp, C.POINTER(SceneDescPOD)]
    lib.rrt_group_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
    lib.rrt_group_render.argtypes = [vp, u32, u32, u32, u32, vp, C.POINTER(Stats)]
    lib.rrt_group_tonemap.argtypes = [vp, vp, u32, u32, u32, vp]
    lib.rrt_tonemap.argtypes = [vp, vp, u32, u32, u32, vp]
    lib.rrt_tonemap_device.argtypes = [vp, vp, u32, u32, u32, vp, vp]
    lib.rrt_selftest_arith.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.rrt_gltf_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(GltfOptions), C.POINTER(C.POINTER(SceneDescPOD))]
    lib.rrt_scene_free.argtypes = [C.POINTER(SceneDescPOD)]
    lib.rrt_scene_free.restype = None
    lib.rrt_png_write.argtypes = [C.c_char_p, vp, u32, u32]
    _lib = lib
    return lib


def _check(status: int, ctx=None):
    if status != RRT_OK:
        msg = load_library().rrt_last_error(ctx)
        raise RrtError(status, msg.decode() if msg else "")


# ---------------------------------------------------------------------------------------------- scene types
class Camera:
    """reference ``src/camera.rs``: ``Camera::new`` (19-39), ``default`` (41-54), ``apply_transform`` (72-77)."""

    def __init__(self, aspect_ratio: float, vertical_fov_deg: float, camera_to_world: np.ndarray, time0=0.0, time1=1.0):
        m = np.asarray(camera_to_world, dtype=np.float32)
        assert m.shape == (3, 4), "camera_to_world is the 3x4 affine (3 columns of the 3x3 part, then translation)"
        self.aspect_ratio = np.float32(aspect_ratio)
        self.vertical_fov = np.float32(vertical_fov_deg)
        radians = np.float32(vertical_fov_deg) * (np.float32(math.pi) / np.float32(180.0))
        self.scale = np.float32(math.tan(float(radians * np.float32(0.5))))
        self.c2w = m.copy()
        self.origin = m[:, 3].copy()  # camera_to_world.transform_point3a(0)
        self.time0, self.time1 = np.float32(time0), np.float32(time1)

    @staticmethod
    def default() -> "Camera":
        m = np.zeros((3, 4), np.float32)
        m[0, 0] = m[1, 1] = m[2, 2] = 1.0
        return Camera(1.0, 30.0, m)

    def pod(self) -> CameraPOD:
        p = CameraPOD()
        cols = [self.c2w[:, 0], self.c2w[:, 1], self.c2w[:, 2], self.c2w[:, 3]]
        flat = np.concatenate(cols).astype(np.float32)
        p.c2w[:] = flat.tolist()
        p.origin[:] = self.origin.tolist()
        p.scale, p.aspect, p.time0, p.time1 = float(self.scale), float(self.aspect_ratio), float(self.time0), float(self.time1)
        return p

    @staticmethod
    def from_pod(p: CameraPOD) -> "Camera":
        cam = Camera.default()
        flat = np.array(p.c2w[:], np.float32)
        cam.c2w = np.stack([flat[0:3], flat[3:6], flat[6:9], flat[9:12]], axis=1)
        cam.origin = np.array(p.origin[:], np.float32)
        cam.scale, cam.aspect_ratio = np.float32(p.scale), np.float32(p.aspect)
        cam.time0, cam.time1 = np.float32(p.time0), np.float32(p.time1)
        return cam


def Lambertian(r, g, b):
    return (MAT_LAMBERTIAN, (r, g, b), 0.0)


def Metal(albedo, fuzziness):
    return (MAT_METAL, tuple(albedo), fuzziness)


def Dielectric(ir):
    return (MAT_DIELECTRIC, (1.0, 1.0, 1.0), ir)


def DiffuseLight(r, g, b):
    return (MAT_DIFFUSE_LIGHT, (r, g, b), 0.0)


class Scene:
    """The reference's ``Scene {camera, lights, world, background}`` (``src/scene.rs:25-30``) as a builder of the
    POD description handed to ``rrt_scene_upload``.  Objects get primitive ids in the order they are added."""

    def __init__(self):
        self.materials: list = []
        self._tri_chunks: list = []
        self._tri_count = 0
        self.spheres: list = []
        self.rects: list = []
        self.cuboids: list = []
        self.instances: list = []
        self.meshes: list = []       # (first_object, object_count)
        self.mesh_objects: list = []
        self.world: list = []
        self.lights: list = []
        self.camera = Camera.default()
        self.background = (0.0, 0.0, 0.0)
        self._keep = None
        self._owned_ptr = None

    # -- materials
    def add_material(self, mat) -> int:
        self.materials.append(mat)
        return len(self.materials) - 1

    # -- primitives (target: the world list, or a mesh under construction)
    def _target(self, mesh):
        return self.world if mesh is None else mesh

    def add_triangles(self, verts: np.ndarray, material: int, mesh=None):
        """verts: (n, 3, 3) float32 -- v0, v1, v2 per triangle (reference ``Triangle::new``, triangle.rs:23-32)."""
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3, 3)
        arr = np.zeros(len(v), TRI_DTYPE)
        arr["v0"], arr["v1"], arr["v2"] = v[:, 0], v[:, 1], v[:, 2]
        arr["material"] = material
        self._tri_chunks.append(arr)
        self._target(mesh).append((OBJ_TRIANGLES, self._tri_count, len(v)))
        self._tri_count += len(v)

    def add_triangle_records(self, arr: np.ndarray, mesh=None):
        assert arr.dtype == TRI_DTYPE
        self._tri_chunks.append(np.ascontiguousarray(arr))
        self._target(mesh).append((OBJ_TRIANGLES, self._tri_count, len(arr)))
        self._tri_count += len(arr)

    def add_sphere(self, center, radius, material: int, mesh=None):
        self.spheres.append((tuple(center), radius, material))
        self._target(mesh).append((OBJ_SPHERES, len(self.spheres) - 1, 1))

    def add_rect(self, axis: int, a0, a1, b0, b1, k, material: int, flip_face=False, mesh=None):
        """axis 0: YZRect, 1: XZRect, 2: XYRect (reference ``src/geometry/aarect.rs``)."""
        self.rects.append((axis, a0, a1, b0, b1, k, material, RECT_FLIP_FACE if flip_face else 0))
        self._target(mesh).append((OBJ_RECTS, len(self.rects) - 1, 1))

    def add_cuboid(self, pmin, pmax, material: int, mesh=None):
        self.cuboids.append((tuple(pmin), tuple(pmax), material))
        self._target(mesh).append((OBJ_CUBOIDS, len(self.cuboids) - 1, 1))

    def new_mesh(self) -> list:
        return []

    def finish_mesh(self, mesh: list) -> int:
        self.meshes.append((len(self.mesh_objects), len(mesh)))
        self.mesh_objects.extend(mesh)
        return len(self.meshes) - 1

    def add_instance(self, mesh_index: int, angle_deg: float = 0.0, offset=(0.0, 0.0, 0.0)):
        """``Translate(RotateY(mesh, angle), offset)`` -- reference ``src/geometry/transform.rs:14-17,54-57``."""
        radians = np.float32(angle_deg) * (np.float32(math.pi) / np.float32(180.0))
        s, c = np.float32(math.sin(float(radians))), np.float32(math.cos(float(radians)))
        self.instances.append((mesh_index, float(s), float(c), tuple(offset)))
        self.world.append((OBJ_INSTANCE, len(self.instances) - 1, 1))

    def add_sphere_light(self, center, radius):
        self.lights.append((LIGHT_SPHERE, (center[0], center[1], center[2], radius, 0.0)))

    def add_xzrect_light(self, x0, x1, z0, z1, k):
        self.lights.append((LIGHT_XZRECT, (x0, x1, z0, z1, k)))

    @property
    def triangle_count(self) -> int:
        return self._tri_count

    # -- POD
    def desc(self) -> SceneDescPOD:
        """Builds (and keeps alive) the ctypes description."""
        d = SceneDescPOD()
        keep = []

        def arr(ctype, items, fill):
            a = (ctype * max(len(items), 1))()
            for i, it in enumerate(items):
                fill(a[i], it)
            keep.append(a)
            return a

        def fmat(p, m):
            p.kind = m[0]; p.rgb[:] = [float(x) for x in m[1]]; p.param = float(m[2])

        mats = arr(MaterialPOD, self.materials, fmat)
        d.materials, d.material_count = C.cast(mats, C.POINTER(MaterialPOD)), len(self.materials)
        if self._tri_chunks:
            tris = np.ascontiguousarray(np.concatenate(self._tri_chunks))
        else:
            tris = np.zeros(1, TRI_DTYPE)
        keep.append(tris)
        d.triangles, d.triangle_count = C.cast(C.c_void_p(tris.ctypes.data), C.POINTER(TrianglePOD)), self._tri_count

        def fsph(p, s):
            p.center[:] = [float(x) for x in s[0]]; p.radius = float(s[1]); p.material = s[2]

        sph = arr(SpherePOD, self.spheres, fsph)
        d.spheres, d.sphere_count = C.cast(sph, C.POINTER(SpherePOD)), len(self.spheres)

        def frect(p, r):
            p.axis, p.a0, p.a1, p.b0, p.b1, p.k, p.material, p.flags = r[0], float(r[1]), float(r[2]), float(r[3]), float(r[4]), float(r[5]), r[6], r[7]

        rc = arr(RectPOD, self.rects, frect)
        d.rects, d.rect_count = C.cast(rc, C.POINTER(RectPOD)), len(self.rects)

        def fcub(p, c):
            p.min[:] = [float(x) for x in c[0]]; p.max[:] = [float(x) for x in c[1]]; p.material = c[2]

        cb = arr(CuboidPOD, self.cuboids, fcub)
        d.cuboids, d.cuboid_count = C.cast(cb, C.POINTER(CuboidPOD)), len(self.cuboids)

        def finst(p, i):
            p.mesh, p.sin_theta, p.cos_theta = i[0], i[1], i[2]; p.offset[:] = [float(x) for x in i[3]]

        ins = arr(InstancePOD, self.instances, finst)
        d.instances, d.instance_count = C.cast(ins, C.POINTER(InstancePOD)), len(self.instances)

        def fmesh(p, m):
            p.first_object, p.object_count = m

        ms = arr(MeshPOD, self.meshes, fmesh)
        d.meshes, d.mesh_count = C.cast(ms, C.POINTER(MeshPOD)), len(self.meshes)

        def fobj(p, o):
            p.type, p.first, p.count = o

        mo = arr(ObjectPOD, self.mesh_objects, fobj)
        d.mesh_objects, d.mesh_object_count = C.cast(mo, C.POINTER(ObjectPOD)), len(self.mesh_objects)
        wo = arr(ObjectPOD, self.world, fobj)
        d.world, d.world_count = C.cast(wo, C.POINTER(ObjectPOD)), len(self.world)

        def flight(p, l):
            p.type = l[0]; p.p[:] = [float(x) for x in l[1]]

        lt = arr(LightPOD, self.lights, flight)
        d.lights, d.light_count = C.cast(lt, C.POINTER(LightPOD)), len(self.lights)
        d.camera = self.camera.pod()
        d.background[:] = [float(x) for x in self.background]
        self._keep = keep
        return d

    def check(self) -> SceneInfo:
        """Host-only validation + BVH build (``rrt_scene_check``); works without a GPU."""
        d = self.desc()
        info = SceneInfo()
        _check(load_library().rrt_scene_check(C.byref(d), C.byref(info)), None)
        return info

    # -- glTF front end (reference ``Scene::from_gltf_file``, src/gltf.rs:314-351)
    @staticmethod
    def from_gltf_file(path: str, camera_light_path: Optional[str] = None, extended: bool = False, scale: float = 1.0,
                       offset: Sequence[float] = (0.0, 0.0, 0.0)) -> "Scene":
        lib = load_library()
        opt = GltfOptions()
        opt.flags = GLTF_EXTENDED if extended else GLTF_COMPAT
        opt.extended_scale = scale
        opt.extended_offset[:] = [float(x) for x in offset]
        out = C.POINTER(SceneDescPOD)()
        st = lib.rrt_gltf_load(path.encode(), camera_light_path.encode() if camera_light_path else None, C.byref(opt), C.byref(out))
        _check(st, None)
        try:
            return Scene.from_pod(out.contents)
        finally:
            lib.rrt_scene_free(out)

    @staticmethod
    def from_pod(d: SceneDescPOD) -> "Scene":
        """Copies a POD description (triangles, spheres, world list, lights, camera; as the glTF loader emits)."""
        s = Scene()
        for i in range(d.material_count):
            m = d.materials[i]
            s.materials.append((m.kind, tuple(m.rgb[:]), m.param))
        if d.triangle_count:
            buf = C.string_at(C.cast(d.triangles, C.c_void_p), d.triangle_count * C.sizeof(TrianglePOD))
            tris = np.frombuffer(buf, TRI_DTYPE).copy()
            s._tri_chunks.append(tris)
            s._tri_count = len(tris)
        for i in range(d.sphere_count):
            p = d.spheres[i]
            s.spheres.append((tuple(p.center[:]), p.radius, p.material))
        for i in range(d.world_count):
            o = d.world[i]
            s.world.append((o.type, o.first, o.count))
        for i in range(d.light_count):
            l = d.lights[i]
            s.lights.append((l.type, tuple(l.p[:])))
        s.camera = Camera.from_pod(d.camera)
        s.background = tuple(d.background[:])
        return s


# ---------------------------------------------------------------------------------------------- renderer
class Renderer:
    """One GPU context (``rrt_create``).  Mirrors the reference's render stage: ``upload`` stands for
    ``BVHNode::new`` (src/gltf.rs:343), ``render`` for the pixel loop (src/main.rs:86-103), ``tonemap`` for
    ``get_color`` + ``put_pixel`` (src/vec3.rs:30-53, src/main.rs:104-109)."""

    def __init__(self, device: int = 0, seed: int = 1, max_depth: int = 4, t_min: float = 1e-4, count_visits: bool = False,
                 device_build: bool = False, unroll_instances: bool = True):
        self.lib = load_library()
        cfg = Config(device, seed, max_depth, t_min,
                     (CFG_COUNT_VISITS if count_visits else 0) | (CFG_DEVICE_BUILD if device_build else 0) |
                     (0 if unroll_instances else CFG_NO_UNROLL))
        self.ctx = C.c_void_p()
        _check(self.lib.rrt_create(C.byref(cfg), C.byref(self.ctx)), None)
        self._scene_keep = None

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx:
            self.lib.rrt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene: Scene) -> SceneInfo:
        d = scene.desc()
        _check(self.lib.rrt_scene_upload(self.ctx, C.byref(d)), self.ctx)
        return self.info()

    def tuning(self) -> Tuning:
        t = Tuning()
        _check(self.lib.rrt_debug_get_tuning(self.ctx, C.byref(t)), self.ctx)
        return t

    def set_tuning(self, **knobs) -> Tuning:
        """Sets the named scheduling knobs (``rrt_debug_set_tuning``; ``None`` = back to the library's rule) and leaves
        the others as they are: ``r.set_tuning(chunks=12, split_items=0)``.  Returns the knobs now in force."""
        t = self.tuning()
        names = {f[0] for f in Tuning._fields_} - {"reserved"}
        for k, v in knobs.items():
            if k not in names:
                raise KeyError(k)
            setattr(t, k, TUNE_AUTO if v is None else int(v))
        _check(self.lib.rrt_debug_set_tuning(self.ctx, C.byref(t)), self.ctx)
        return self.tuning()

    def info(self) -> SceneInfo:
        info = SceneInfo()
        _check(self.lib.rrt_scene_get_info(self.ctx, C.byref(info)), self.ctx)
        return info

    def intersect(self, rays: np.ndarray, t_min: float = 1e-4, t_max: float = float("inf")) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        _check(self.lib.rrt_intersect(self.ctx, rays.ctypes.data, len(rays), t_min, t_max, hits.ctypes.data), self.ctx)
        return hits

    def intersect_device(self, d_rays: int, n: int, d_hits: int, t_min: float = 1e-4, t_max: float = float("inf"), stream: int = 0):
        _check(self.lib.rrt_intersect_device(self.ctx, d_rays, n, t_min, t_max, d_hits, stream), self.ctx)

    def render(self, width: int, height: int, spp: int, spp_offset: int = 0):
        accum = np.zeros((height, width, 3), np.float32)
        st = Stats()
        _check(self.lib.rrt_render(self.ctx, width, height, spp, spp_offset, accum.ctypes.data, C.byref(st)), self.ctx)
        return accum, st

    def render_device(self, width: int, height: int, spp: int, spp_offset: int, d_accum: int, stream: int = 0):
        _check(self.lib.rrt_render_device(self.ctx, width, height, spp, spp_offset, d_accum, stream), self.ctx)

    def render_device_noalloc(self, width: int, height: int, spp: int, spp_offset: int, d_accum: int, stream: int = 0):
        """``rrt_render_device`` out of the library's static workspace: allocates nothing, refuses large frames."""
        _check(self.lib.rrt_render_device_noalloc(self.ctx, width, height, spp, spp_offset, d_accum, stream), self.ctx)

    def render_stats(self) -> Stats:
        st = Stats()
        _check(self.lib.rrt_render_stats(self.ctx, C.byref(st)), self.ctx)
        return st

    def tonemap(self, accum: np.ndarray, spp_total: int) -> np.ndarray:
        accum = np.ascontiguousarray(accum, dtype=np.float32)
        h, w, _ = accum.shape
        rgb = np.zeros((h, w, 3), np.uint8)
        _check(self.lib.rrt_tonemap(self.ctx, accum.ctypes.data, w, h, spp_total, rgb.ctypes.data), self.ctx)
        return rgb

    def tonemap_device(self, d_accum: int, width: int, height: int, spp_total: int, d_rgb8: int, stream: int = 0):
        _check(self.lib.rrt_tonemap_device(self.ctx, d_accum, width, height, spp_total, d_rgb8, stream), self.ctx)

    def selftest_arith(self):
        """(reciprocal mismatches, square-root mismatches) of the kernels' short forms against the IEEE operations
        over all 2^32 float bit patterns."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self.lib.rrt_selftest_arith(self.ctx, C.byref(a), C.byref(b)), self.ctx)
        return a.value, b.value


class RendererGroup:
    """Several GPUs in one process (``rrt_group_*``): the sample range of a frame is split over the members, the
    accumulation buffers are summed on member 0.  Stands for the reference's rayon pool (``src/main.rs:42-49,86``)."""

    def __init__(self, devices: Sequence[int], seed: int = 1, max_depth: int = 4, t_min: float = 1e-4,
                 count_visits: bool = False, device_build: bool = False, unroll_instances: bool = True,
                 force_rccl: bool = False):
        self.lib = load_library()
        cfg = Config(0, seed, max_depth, t_min,
                     (CFG_COUNT_VISITS if count_visits else 0) | (CFG_DEVICE_BUILD if device_build else 0) |
                     (0 if unroll_instances else CFG_NO_UNROLL) | (CFG_GROUP_FORCE_RCCL if force_rccl else 0))
        devs = (C.c_int32 * len(devices))(*devices)
        self.grp = C.c_void_p()
        _check(self.lib.rrt_group_create(C.byref(cfg), devs, len(devices), C.byref(self.grp)), None)

    def _check(self, status: int):
        if status != RRT_OK:
            msg = self.lib.rrt_group_last_error(self.grp)
            raise RrtError(status, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "grp", None) is not None and self.grp:
            self.lib.rrt_group_destroy(self.grp)
            self.grp = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self) -> int:
        return self.lib.rrt_group_size(self.grp)

    @property
    def reduce_path(self) -> str:
        return self.lib.rrt_group_reduce_path(self.grp).decode()

    def upload(self, scene: Scene) -> SceneInfo:
        d = scene.desc()
        self._check(self.lib.rrt_group_scene_upload(self.grp, C.byref(d)))
        info = SceneInfo()
        self._check(self.lib.rrt_group_scene_get_info(self.grp, C.byref(info)))
        return info

    def render(self, width: int, height: int, spp_total: int, spp_offset: int = 0):
        accum = np.zeros((height, width, 3), np.float32)
        st = Stats()
        self._check(self.lib.rrt_group_render(self.grp, width, height, spp_total, spp_offset, accum.ctypes.data, C.byref(st)))
        return accum, st

    def tonemap(self, accum: np.ndarray, spp_total: int) -> np.ndarray:
        accum = np.ascontiguousarray(accum, dtype=np.float32)
        h, w, _ = accum.shape
        rgb = np.zeros((h, w, 3), np.uint8)
        self._check(self.lib.rrt_group_tonemap(self.grp, accum.ctypes.data, w, h, spp_total, rgb.ctypes.data))
        return rgb


def png_write(path: str, rgb8: np.ndarray):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    _check(load_library().rrt_png_write(path.encode(), rgb8.ctypes.data, w, h), None)


def image_width(height: int, aspect_ratio: float) -> int:
    """``((height as f32) * aspect_ratio) as u32`` -- reference src/main.rs:81."""
    return int(np.float32(height) * np.float32(aspect_ratio))
