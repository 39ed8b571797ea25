"""ctypes wrapper of ``oracle/liboracle.so`` -- the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``; never by the product package.  Parity unpinned: the reference ships no golden vectors, so the
restatement is pinned by analytic known-answer cases only (see the header of ``rrt_oracle.c``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

HIT_BRUTE, HIT_REF_BVH, HIT_REF_BVH_RUNNING_SLAB = 0, 1, 2

_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
    lib.oracle_scene_create.argtypes = [vp]
    lib.oracle_scene_create.restype = vp
    lib.oracle_scene_destroy.argtypes = [vp]
    lib.oracle_scene_destroy.restype = None
    lib.oracle_scene_prim_count.argtypes = [vp]
    lib.oracle_scene_prim_count.restype = u32
    lib.oracle_scene_build_ref_bvh.argtypes = [vp, u32]
    lib.oracle_scene_build_ref_bvh.restype = C.c_int
    lib.oracle_intersect.argtypes = [vp, C.c_int, vp, C.c_size_t, f32, f32, vp, C.c_int]
    lib.oracle_intersect.restype = None
    lib.oracle_camera_rays.argtypes = [vp, u32, u32, vp, C.c_size_t, vp]
    lib.oracle_camera_rays.restype = None
    lib.oracle_render.argtypes = [vp, C.c_int, u32, u32, u32, u32, u32, u32, f32, vp, C.c_int]
    lib.oracle_render.restype = C.c_uint64
    lib.oracle_tonemap.argtypes = [vp, u32, u32, u32, vp]
    lib.oracle_tonemap.restype = None
    lib.oracle_rng_fill.argtypes = [u32, u32, u32, u32, vp]
    lib.oracle_rng_fill.restype = None
    lib.oracle_gen_range_f32.argtypes = [u32, u32, u32, f32, f32, u32, vp]
    lib.oracle_gen_range_f32.restype = None
    lib.oracle_gen_range_map.argtypes = [u32, f32, f32, C.POINTER(f32)]
    lib.oracle_gen_range_map.restype = C.c_int
    lib.oracle_onb.argtypes = [vp, vp]
    lib.oracle_onb.restype = None
    lib.oracle_reflectance.argtypes = [f32, f32]
    lib.oracle_reflectance.restype = f32
    lib.oracle_refract.argtypes = [vp, vp, f32, vp]
    lib.oracle_refract.restype = C.c_int
    lib.oracle_reflect.argtypes = [vp, vp, vp]
    lib.oracle_reflect.restype = None
    lib.oracle_lights_pdf.argtypes = [vp, vp, vp]
    lib.oracle_lights_pdf.restype = f32
    lib.oracle_lights_random.argtypes = [vp, vp, u32, u32, vp]
    lib.oracle_lights_random.restype = None
    lib.oracle_cosine_samples.argtypes = [vp, u32, u32, vp, vp]
    lib.oracle_cosine_samples.restype = None
    _lib = lib
    return lib


def default_threads() -> int:
    """Threads for the row-parallel loops: the CPUs this process may run on (its affinity mask), bounded by
    OMP_NUM_THREADS where the environment sets it (a GPU box exports its per-job CPU share that way; the machine has
    many more cores than the job may use)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, min(n, 256))


class OracleScene:
    """Flattened world of a scene description (any object exposing ``desc()`` returning the ctypes POD)."""

    def __init__(self, scene):
        self.lib = load()
        self._desc = scene.desc()  # keeps the arrays alive
        self._scene = scene
        self.handle = self.lib.oracle_scene_create(C.byref(self._desc))
        if not self.handle:
            raise MemoryError("oracle_scene_create failed")
        self._has_ref_bvh = False

    def close(self):
        if self.handle:
            self.lib.oracle_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def prim_count(self) -> int:
        return self.lib.oracle_scene_prim_count(self.handle)

    def build_ref_bvh(self, axis_seed: int = 1):
        rc = self.lib.oracle_scene_build_ref_bvh(self.handle, axis_seed)
        if rc != 0:
            raise ValueError(f"reference-style BVH unavailable for this scene (rc={rc})")
        self._has_ref_bvh = True

    def intersect(self, rays: np.ndarray, t_min=1e-4, t_max=float("inf"), mode=HIT_BRUTE, threads=None) -> np.ndarray:
        hit_dtype = np.dtype([("prim_id", np.int32), ("t", np.float32), ("u", np.float32), ("v", np.float32)])
        rays = np.ascontiguousarray(rays)
        assert rays.dtype.itemsize == 28
        out = np.zeros(len(rays), hit_dtype)
        if mode in (HIT_REF_BVH, HIT_REF_BVH_RUNNING_SLAB) and not self._has_ref_bvh:
            self.build_ref_bvh()
        self.lib.oracle_intersect(self.handle, mode, rays.ctypes.data, len(rays), t_min, t_max, out.ctypes.data,
                                  threads or default_threads())
        return out

    def camera_rays(self, width: int, height: int, xy: np.ndarray) -> np.ndarray:
        ray_dtype = np.dtype([("origin", np.float32, 3), ("direction", np.float32, 3), ("time", np.float32)])
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        out = np.zeros(len(xy), ray_dtype)
        self.lib.oracle_camera_rays(C.byref(self._desc.camera), width, height, xy.ctypes.data, len(xy), out.ctypes.data)
        return out

    def render(self, width, height, spp, spp_offset=0, seed=1, max_depth=4, t_min=1e-4, mode=HIT_BRUTE, threads=None):
        accum = np.zeros((height, width, 3), np.float32)
        if mode in (HIT_REF_BVH, HIT_REF_BVH_RUNNING_SLAB) and not self._has_ref_bvh:
            self.build_ref_bvh()
        rays = self.lib.oracle_render(self.handle, mode, width, height, spp, spp_offset, seed, max_depth, t_min,
                                      accum.ctypes.data, threads or default_threads())
        return accum, int(rays)


def tonemap(accum: np.ndarray, spp: int) -> np.ndarray:
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    h, w, _ = accum.shape
    out = np.zeros((h, w, 3), np.uint8)
    load().oracle_tonemap(accum.ctypes.data, w, h, spp, out.ctypes.data)
    return out


def rng_stream(seed: int, pixel: int, sample: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint32)
    load().oracle_rng_fill(seed, pixel, sample, n, out.ctypes.data)
    return out


# ---- single functions of the restatement, for the known-answer tests
def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32).reshape(3)


def gen_range(seed: int, pixel: int, sample: int, low: float, high: float, n: int) -> np.ndarray:
    out = np.zeros(n, np.float32)
    load().oracle_gen_range_f32(seed, pixel, sample, low, high, n, out.ctypes.data)
    return out


def gen_range_map(draw: int, low: float, high: float):
    """(value the 32-bit draw maps to on the first try, whether it is accepted)."""
    v = C.c_float(0)
    ok = load().oracle_gen_range_map(draw, low, high, C.byref(v))
    return np.float32(v.value), bool(ok)


def onb(n) -> np.ndarray:
    """rows u, v, w of OrthNormBasis::from_w(n)."""
    a, out = _f3(n), np.zeros(9, np.float32)
    load().oracle_onb(a.ctypes.data, out.ctypes.data)
    return out.reshape(3, 3)


def reflectance(cosine: float, ref_idx: float) -> float:
    return float(load().oracle_reflectance(cosine, ref_idx))


def refract(uv, n, ratio: float):
    a, b, out = _f3(uv), _f3(n), np.zeros(3, np.float32)
    ok = load().oracle_refract(a.ctypes.data, b.ctypes.data, ratio, out.ctypes.data)
    return (out if ok else None)


def reflect(v, n) -> np.ndarray:
    a, b, out = _f3(v), _f3(n), np.zeros(3, np.float32)
    load().oracle_reflect(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def lights_pdf(oscene: OracleScene, origin, v) -> float:
    a, b = _f3(origin), _f3(v)
    return float(load().oracle_lights_pdf(oscene.handle, a.ctypes.data, b.ctypes.data))


def lights_random(oscene: OracleScene, origin, seed: int, n: int) -> np.ndarray:
    a, out = _f3(origin), np.zeros((n, 3), np.float32)
    load().oracle_lights_random(oscene.handle, a.ctypes.data, seed, n, out.ctypes.data)
    return out


def cosine_samples(normal, seed: int, n: int):
    a, dirs, pdfs = _f3(normal), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    load().oracle_cosine_samples(a.ctypes.data, seed, n, dirs.ctypes.data, pdfs.ctypes.data)
    return dirs, pdfs
