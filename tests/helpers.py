"""Shared scene and ray-batch builders for the parity tests (numpy only; deterministic)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")


# ------------------------------------------------------------------------------------------------ scenes
def scene_default(rrt):
    """config 1: assets/default.gltf (12 triangles) + camera and light of assets/suzanne.gltf (reference src/gltf.rs:317)."""
    return rrt.Scene.from_gltf_file(os.path.join(ASSETS, "default.gltf"), os.path.join(ASSETS, "suzanne.gltf"))


def scene_suzanne(rrt):
    """config 2: assets/suzanne.gltf (968 triangles, Metal)."""
    return rrt.Scene.from_gltf_file(os.path.join(ASSETS, "suzanne.gltf"), os.path.join(ASSETS, "suzanne.gltf"))


def scene_mixed(rrt):
    """Every primitive and material kind in one flat scene: a small displaced mesh, spheres (Lambertian, Metal,
    Dielectric, light), rectangles of the three orientations, a cuboid, one sphere light."""
    from rust_ray_tracer_amd import scenes

    s = rrt.Scene()
    grey = s.add_material(rrt.Lambertian(0.7, 0.7, 0.7))
    metal = s.add_material(rrt.Metal((0.8, 0.6, 0.2), 0.3))
    glass = s.add_material(rrt.Dielectric(1.5))
    red = s.add_material(rrt.Lambertian(0.65, 0.05, 0.05))
    light = s.add_material(rrt.DiffuseLight(60.0, 60.0, 60.0))
    verts = scenes.height_field(24, 16, 0.5, 1.5, seed=5)
    s.add_triangles(scenes.grid_triangles(verts), grey)
    s.add_sphere((0.0, 1.6, 0.0), 1.0, glass)
    s.add_sphere((-2.5, 1.2, 1.0), 0.8, metal)
    s.add_sphere((2.4, 1.0, -0.5), 0.7, red)
    s.add_rect(2, -6.0, 6.0, -1.0, 5.0, -4.5, red)        # XY wall at z = -4.5
    s.add_rect(0, -1.0, 5.0, -4.5, 4.5, -6.0, metal)      # YZ wall at x = -6
    s.add_rect(1, -6.0, 6.0, -4.5, 4.5, 5.0, grey)        # XZ ceiling at y = 5
    s.add_cuboid((3.5, -0.5, 1.0), (5.0, 1.5, 2.5), grey)
    lp = (1.0, 4.0, 2.0)
    s.add_sphere(lp, 0.4, light)
    s.add_sphere_light(lp, 0.4)
    s.camera = rrt.Camera(16.0 / 9.0, 45.0, scenes.look_at_c2w((1.0, 3.0, 9.0), (0.0, 1.0, 0.0)))
    s.background = (0.05, 0.06, 0.08)
    return s


def scene_instanced(rrt):
    """A small two-level scene: one mesh (blob + sphere + cuboid) instanced 9 times as Translate(RotateY(mesh)),
    next to top-level triangles and a light."""
    from rust_ray_tracer_amd import scenes

    s = rrt.Scene()
    grey = s.add_material(rrt.Lambertian(0.7, 0.7, 0.7))
    metal = s.add_material(rrt.Metal((0.8, 0.8, 0.9), 0.1))
    light = s.add_material(rrt.DiffuseLight(300.0, 300.0, 300.0))
    m = s.new_mesh()
    s.add_triangles(scenes.blob_mesh(14, 18, 0.8, 0.4, seed=3), grey, mesh=m)
    s.add_sphere((0.0, 1.1, 0.0), 0.3, metal, mesh=m)
    s.add_cuboid((-0.9, -1.0, -0.9), (0.9, -0.8, 0.9), metal, mesh=m)
    mi = s.finish_mesh(m)
    g = np.array([[[-9, -1.2, -9], [-9, -1.2, 9], [9, -1.2, -9]], [[9, -1.2, -9], [-9, -1.2, 9], [9, -1.2, 9]]], np.float32)
    s.add_triangles(g, grey)
    k = 0
    for i in range(3):
        for j in range(3):
            s.add_instance(mi, angle_deg=37.0 * k - 90.0, offset=((i - 1) * 2.6 + 0.1 * k, 0.0, (j - 1) * 2.6 - 0.05 * k))
            k += 1
    lp = (0.0, 6.0, 1.0)
    s.add_sphere(lp, 0.5, light)
    s.add_sphere_light(lp, 0.5)
    s.camera = rrt.Camera(16.0 / 9.0, 42.0, scenes.look_at_c2w((0.5, 5.0, 9.0), (0.0, 0.0, 0.0)))
    s.background = (0.051, 0.051, 0.051)
    return s


def scene_instanced_tris(rrt):
    """Triangle-only meshes instanced as Translate(RotateY(mesh)): the kind of scene the library unrolls into one
    world-space tree (two meshes, 12 instances, top-level triangles, a light)."""
    from rust_ray_tracer_amd import scenes

    s = rrt.Scene()
    grey = s.add_material(rrt.Lambertian(0.7, 0.7, 0.7))
    metal = s.add_material(rrt.Metal((0.8, 0.8, 0.9), 0.1))
    glass = s.add_material(rrt.Dielectric(1.5))
    light = s.add_material(rrt.DiffuseLight(300.0, 300.0, 300.0))
    m = s.new_mesh()
    tris = scenes.blob_mesh(16, 20, 0.8, 0.4, seed=4)
    s.add_triangles(tris[: len(tris) // 2], grey, mesh=m)
    s.add_triangles(tris[len(tris) // 2:], metal, mesh=m)
    blob = s.finish_mesh(m)
    m = s.new_mesh()
    s.add_triangles(scenes.blob_mesh(6, 7, 0.45, 0.2, seed=9), glass, mesh=m)
    small = s.finish_mesh(m)
    g = np.array([[[-9, -1.2, -9], [-9, -1.2, 9], [9, -1.2, -9]], [[9, -1.2, -9], [-9, -1.2, 9], [9, -1.2, 9]]], np.float32)
    s.add_triangles(g, grey)
    k = 0
    for i in range(3):
        for j in range(3):
            s.add_instance(blob, angle_deg=41.0 * k - 120.0, offset=((i - 1) * 2.5 + 0.1 * k, 0.0, (j - 1) * 2.5 - 0.05 * k))
            k += 1
    for k in range(3):
        s.add_instance(small, angle_deg=100.0 * k, offset=(-2.0 + 2.0 * k, 1.6, 0.4 * k))
    lp = (0.0, 6.0, 1.0)
    s.add_sphere(lp, 0.5, light)
    s.add_sphere_light(lp, 0.5)
    s.camera = rrt.Camera(16.0 / 9.0, 42.0, scenes.look_at_c2w((0.5, 5.0, 9.0), (0.0, 0.0, 0.0)))
    s.background = (0.051, 0.051, 0.051)
    return s


def scene_cornell(rrt):
    from rust_ray_tracer_amd import scenes

    return scenes.cornell_box(rrt, aspect_ratio=1.0)


def scene_synthetic_small(rrt):
    """The benchmark's displaced-mesh scene at 2*64*32 = 4096 triangles, through the glTF front end."""
    from rust_ray_tracer_amd import scenes

    return scenes.synthetic_mesh(rrt, ASSETS, subdiv=32)


SCENES = {
    "default": scene_default,
    "suzanne": scene_suzanne,
    "mixed": scene_mixed,
    "instanced": scene_instanced,
    "instanced_tris": scene_instanced_tris,
    "cornell": scene_cornell,
    "synthetic": scene_synthetic_small,
}


# ------------------------------------------------------------------------------------------------ ray batches
RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("direction", np.float32, 3), ("time", np.float32)])


def make_rays(origins, directions):
    o = np.asarray(origins, np.float32).reshape(-1, 3)
    d = np.asarray(directions, np.float32).reshape(-1, 3)
    if len(o) == 1 and len(d) > 1:
        o = np.repeat(o, len(d), axis=0)
    r = np.zeros(len(d), RAY_DTYPE)
    r["origin"], r["direction"] = o, d
    return r


def primary_rays(oscene, width, height, n, seed):
    """n jittered camera rays (reference src/main.rs:92-94) drawn with a numpy generator."""
    g = np.random.default_rng(seed)
    xy = np.stack([g.uniform(0, width, n), g.uniform(0, height, n)], axis=1).astype(np.float32)
    return oscene.camera_rays(width, height, xy)


def bounce_rays(oscene, prim_rays, seed):
    """Secondary rays: from the oracle's hit points of `prim_rays` into uniformly random directions."""
    g = np.random.default_rng(seed)
    hits = oscene.intersect(prim_rays)
    ok = hits["prim_id"] >= 0
    p = prim_rays["origin"][ok] + hits["t"][ok, None] * prim_rays["direction"][ok]
    d = g.normal(size=(len(p), 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return make_rays(p.astype(np.float32), d.astype(np.float32))


def adversarial_rays(tri_records, origin, seed, n_tris=96):
    """Rays aimed exactly at vertices, edge midpoints and centroids of triangles (grazing u = 0, v = 0, u + v = 1),
    axis-parallel rays, rays with zero direction components, and rays starting on triangle planes."""
    g = np.random.default_rng(seed)
    origin = np.asarray(origin, np.float32)
    rays = []
    if len(tri_records):
        pick = g.choice(len(tri_records), size=min(n_tris, len(tri_records)), replace=False)
        t = tri_records[pick]
        v0, v1, v2 = t["v0"], t["v1"], t["v2"]
        targets = [v0, v1, v2, (v0 + v1) * np.float32(0.5), (v1 + v2) * np.float32(0.5), (v0 + v2) * np.float32(0.5),
                   (v0 + v1 + v2) / np.float32(3.0)]
        for tg in targets:
            rays.append(make_rays(origin[None, :], (tg - origin).astype(np.float32)))
            # the same targets seen from the front side of each triangle, along its normal
            n = np.cross(v1 - v0, v2 - v0)
            n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
            rays.append(make_rays((tg + n * np.float32(2.0)).astype(np.float32), (-n).astype(np.float32)))
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
                     [1, 1, 0], [0, -1, 1], [-1, 0, -1], [1e-30, -1, 1e-30], [0.0, -1.0, -0.0]], np.float32)
    starts = g.uniform(-3, 3, size=(len(axes) * 8, 3)).astype(np.float32)
    rays.append(make_rays(starts, np.tile(axes, (8, 1))))
    rays.append(make_rays(origin[None, :], axes))
    return np.concatenate(rays)


def scene_tri_records(scene):
    if scene._tri_chunks:
        return np.concatenate(scene._tri_chunks)
    return np.zeros(0, dtype=[("v0", np.float32, 3), ("v1", np.float32, 3), ("v2", np.float32, 3), ("material", np.uint32)])


def ray_suite(scene, oscene, width=160, height=90, n=1024, seed=11):
    """The three ray classes of the parity gate: primary, bounce, adversarial."""
    prim = primary_rays(oscene, width, height, n, seed)
    bounce = bounce_rays(oscene, prim, seed + 1)
    adv = adversarial_rays(scene_tri_records(scene), scene.camera.origin, seed + 2)
    return {"primary": prim, "bounce": bounce, "adversarial": adv}


def assert_hits_match(got, want, rel_t=1e-5, abs_uv=1e-5):
    """ids bit-exact; t within rel_t relative; u, v within abs_uv."""
    bad = np.nonzero(got["prim_id"] != want["prim_id"])[0]
    assert len(bad) == 0, f"{len(bad)} closest-hit ids differ, first at ray {bad[:5]}: got {got[bad[:5]]} want {want[bad[:5]]}"
    hit = want["prim_id"] >= 0
    tw, tg = want["t"][hit], got["t"][hit]
    assert np.all(np.abs(tg - tw) <= rel_t * np.abs(tw)), "hit t differs by more than the relative tolerance"
    # a sphere's far-root uv is NaN on both sides (acos of an un-normalised coordinate, reference sphere.rs:82)
    assert np.all(np.isclose(got["u"][hit], want["u"][hit], rtol=0, atol=abs_uv, equal_nan=True))
    assert np.all(np.isclose(got["v"][hit], want["v"][hit], rtol=0, atol=abs_uv, equal_nan=True))
    assert np.all(np.isinf(got["t"][~hit]))


# ------------------------------------------------------------------------------------------------ device memory
class DeviceArray:
    """A float32 array in GPU memory for the tests of the device-pointer entry points: hipMalloc / hipMemcpy / hipFree
    of the HIP runtime the library itself is linked to (the loader hands back the copy that is already mapped), so that
    no second GPU runtime is initialised inside the test process."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes as C

            lib = C.CDLL("libamdhip64.so.7")
            lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            lib.hipFree.argtypes = [C.c_void_p]
            lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
            lib.hipDeviceSynchronize.argtypes = []
            lib.hipSetDevice.argtypes = [C.c_int]
            cls._hip = lib
        return cls._hip

    def __init__(self, shape, fill_byte=0xFF):
        import ctypes as C

        self.shape = tuple(shape)
        self.nbytes = int(np.prod(self.shape)) * 4
        hip = self.hip()
        assert hip.hipSetDevice(0) == 0
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), self.nbytes) == 0, "hipMalloc failed"
        self.ptr = p.value
        assert hip.hipMemset(self.ptr, fill_byte, self.nbytes) == 0  # 0xFF bytes: NaN in every float

    def numpy(self):
        hip = self.hip()
        assert hip.hipDeviceSynchronize() == 0
        out = np.empty(self.shape, np.float32)
        assert hip.hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def free(self):
        if self.ptr:
            self.hip().hipFree(self.ptr)
            self.ptr = None
