"""Parity gate 1: closest hit of the HIP path (through the C ABI, rrt_intersect) against the CPU oracle and the committed
golden vectors.  Primitive ids bit-exact, t within 1e-5 relative (north star), u and v within 1e-5 absolute."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

SCENE_NAMES = ["default", "suzanne", "mixed", "instanced", "instanced_tris", "cornell", "synthetic"]


@pytest.fixture(scope="module")
def renderer_cache(rrt):
    cache = {}
    yield cache
    for r, _ in cache.values():
        r.close()


def get(rrt, cache, name):
    if name not in cache:
        scene = helpers.SCENES[name](rrt)
        r = rrt.Renderer(device=0, seed=1)
        r.upload(scene)
        cache[name] = (r, scene)
    return cache[name]


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_ray_classes_match_the_oracle(rrt, oracle, renderer_cache, name):
    r, scene = get(rrt, renderer_cache, name)
    o = oracle.OracleScene(scene)
    suite = helpers.ray_suite(scene, o, n=2048, seed=33)
    for kind, rays in suite.items():
        got, want = r.intersect(rays), o.intersect(rays)
        helpers.assert_hits_match(got, want)
        if kind == "primary" and name != "default":
            assert (want["prim_id"] >= 0).mean() > 0.05, "the batch should hit something"


@pytest.mark.parametrize("name", ["default", "suzanne", "mixed", "instanced", "cornell"])
def test_golden_vectors(rrt, renderer_cache, golden_dir, name):
    r, _ = get(rrt, renderer_cache, name)
    g = np.load(os.path.join(golden_dir, f"rays_{name}.npz"))
    for kind in ("primary", "bounce", "adversarial"):
        helpers.assert_hits_match(r.intersect(g[f"{kind}_rays"]), g[f"{kind}_hits"])


def test_unrolled_and_two_level_instances_give_the_same_hits(rrt, oracle, renderer_cache, monkeypatch):
    """Instanced meshes are written out as world-space copies of their tree inside the top-level BVH by
    default; RRT_CFG_NO_UNROLL keeps the two-level walk of the reference (src/geometry/transform.rs:22-31,99-130).  Both
    run the same primitive tests on the same operands: records identical bit for bit, and equal to the oracle's."""
    r, scene = get(rrt, renderer_cache, "instanced_tris")
    assert r.info().unrolled_instances == 12
    two = rrt.Renderer(unroll_instances=False)
    info = two.upload(scene)
    assert info.unrolled_instances == 0 and info.node_count < r.info().node_count
    o = oracle.OracleScene(scene)
    for kind, rays in helpers.ray_suite(scene, o, n=4096, seed=5).items():
        a, b = r.intersect(rays), two.intersect(rays)
        assert np.array_equal(a, b), kind
        helpers.assert_hits_match(a, o.intersect(rays))
    two.close()
    # the plain one-lane-per-ray kernel (an experiment knob) walks the unrolled tree as well
    rays = helpers.primary_rays(o, 160, 90, 2048, seed=9)
    want = r.intersect(rays)
    r.set_tuning(intersect_static=1)
    assert np.array_equal(r.intersect(rays), want)
    r.set_tuning(intersect_static=None)
    # meshes that hold spheres and rectangles (the reference's own use: Translate(RotateY(Cuboid)), src/scene.rs:216-236)
    # are unrolled the same way, their leaves naming the instance like triangle leaves do
    for name, n_inst in (("instanced", 9), ("cornell", 2)):
        r2, scene2 = get(rrt, renderer_cache, name)
        assert r2.info().unrolled_instances == n_inst
        two = rrt.Renderer(unroll_instances=False)
        assert two.upload(scene2).unrolled_instances == 0
        o2 = oracle.OracleScene(scene2)
        for kind, rays in helpers.ray_suite(scene2, o2, n=4096, seed=6).items():
            a, b = r2.intersect(rays), two.intersect(rays)
            assert a.tobytes() == b.tobytes(), (name, kind)  # (bytes: a sphere's far-root uv is NaN on both sides)
            helpers.assert_hits_match(a, o2.intersect(rays))
        two.close()


def test_an_instance_listed_twice_gets_two_trees(rrt, oracle):
    """Two world objects that name the same rrt_instance: ids, tree and instance slot per object, hits and image as the
    two-level form and the oracle give them."""
    from rust_ray_tracer_amd import scenes

    s = rrt.Scene()
    m = s.add_material(rrt.Lambertian(0.6, 0.6, 0.6))
    light = s.add_material(rrt.DiffuseLight(100.0, 100.0, 100.0))
    mesh = s.new_mesh()
    s.add_triangles(scenes.blob_mesh(8, 9, 0.6, 0.3, seed=1), m, mesh=mesh)
    s.add_sphere((0.0, 0.9, 0.0), 0.2, m, mesh=mesh)
    mi = s.finish_mesh(mesh)
    s.add_instance(mi, angle_deg=30.0, offset=(1.0, 0.0, 0.0))
    s.world.append((rrt.OBJ_INSTANCE, 0, 1))
    s.add_instance(mi, angle_deg=-50.0, offset=(-1.5, 0.2, 0.3))
    s.world.append((rrt.OBJ_INSTANCE, 0, 1))
    s.add_sphere((0.0, 5.0, 0.0), 0.5, light)
    s.add_sphere_light((0.0, 5.0, 0.0), 0.5)
    s.camera = rrt.Camera(16.0 / 9.0, 45.0, scenes.look_at_c2w((0.0, 2.0, 6.0), (0.0, 0.0, 0.0)))
    a, b = rrt.Renderer(seed=3), rrt.Renderer(seed=3, unroll_instances=False)
    assert a.upload(s).unrolled_instances == 4 and b.upload(s).unrolled_instances == 0
    o = oracle.OracleScene(s)
    for kind, rays in helpers.ray_suite(s, o, n=2048, seed=8).items():
        ga, gb = a.intersect(rays), b.intersect(rays)
        assert ga.tobytes() == gb.tobytes(), kind
        helpers.assert_hits_match(ga, o.intersect(rays))
    ia, sa = a.render(96, 54, 4)
    ib, sb = b.render(96, 54, 4)
    assert sa.rays == sb.rays and np.array_equal(ia, ib)
    a.close()
    b.close()


def test_t_range_arguments(rrt, oracle, renderer_cache):
    r, scene = get(rrt, renderer_cache, "mixed")
    o = oracle.OracleScene(scene)
    rays = helpers.primary_rays(o, 160, 90, 1024, seed=2)
    for t_min, t_max in ((1e-4, 8.0), (7.0, 9.5), (0.0, float("inf")), (10.0, 10.0)):
        helpers.assert_hits_match(r.intersect(rays, t_min=t_min, t_max=t_max), o.intersect(rays, t_min=t_min, t_max=t_max))


def test_ties_resolve_to_the_lowest_id(rrt, oracle):
    s = rrt.Scene()
    m = s.add_material(rrt.Lambertian(0.5, 0.5, 0.5))
    tri = np.array([[(0, 0, 0), (1, 0, 0), (0, 1, 0)]], np.float32)
    s.add_triangles(np.repeat(tri, 37, axis=0), m)  # coincident triangles in several leaves
    s.add_rect(2, -1, 1, -1, 1, 0.0, m)             # and a rectangle in the same plane (id 37)
    r = rrt.Renderer()
    r.upload(s)
    g = np.random.default_rng(0)
    p = g.uniform(0.05, 0.45, size=(256, 2)).astype(np.float32)
    rays = helpers.make_rays(np.concatenate([p, np.ones((256, 1), np.float32)], 1), np.tile(np.float32([0, 0, -1]), (256, 1)))
    got = r.intersect(rays)
    assert np.all(got["prim_id"] == 0) and np.all(got["t"] == 1.0)
    helpers.assert_hits_match(got, oracle.OracleScene(s).intersect(rays))
    r.close()


def test_empty_and_ragged_batches(rrt, oracle, renderer_cache):
    r, scene = get(rrt, renderer_cache, "default")
    o = oracle.OracleScene(scene)
    assert len(r.intersect(np.zeros(0, rrt.RAY_DTYPE))) == 0
    for n in (1, 63, 64, 65, 1000):
        rays = helpers.primary_rays(o, 71, 40, n, seed=n)
        helpers.assert_hits_match(r.intersect(rays), o.intersect(rays))
    # degenerate rays: zero direction, NaN direction, infinite origin -> misses, never a fault
    bad = helpers.make_rays([(0, 0, 5), (0, 0, 5), (np.inf, 0, 0), (0, 0, 5)], [(0, 0, 0), (np.nan, 0, -1), (0, 0, -1), (0, 0, -np.inf)])
    got, want = r.intersect(bad), o.intersect(bad)
    assert np.array_equal(got["prim_id"], want["prim_id"])


def test_empty_scene_and_single_primitive(rrt, oracle):
    s = rrt.Scene()
    m = s.add_material(rrt.Lambertian(0.5, 0.5, 0.5))
    r = rrt.Renderer()
    r.upload(s)
    rays = helpers.make_rays([(0, 0, 5)] * 3, [(0, 0, -1), (1, 0, 0), (0, 1, 0)])
    assert np.all(r.intersect(rays)["prim_id"] == -1)
    s.add_sphere((0, 0, 0), 1.0, m)
    r.upload(s)  # re-upload replaces the scene
    got = r.intersect(rays)
    assert got["prim_id"][0] == 0 and got["t"][0] == 4.0 and got["prim_id"][1] == -1
    helpers.assert_hits_match(got, oracle.OracleScene(s).intersect(rays))
    r.close()


def test_large_mesh_random_rays_match_exhaustive_search(rrt, oracle):
    """Full-size geometry (the 1,048,576-triangle benchmark mesh): a random subset of rays against brute force, and
    size-independent properties on a large batch."""
    from rust_ray_tracer_amd import scenes

    scene = scenes.synthetic_mesh(rrt, helpers.ASSETS, subdiv=512)
    assert scene.triangle_count == 1048576
    r = rrt.Renderer()
    info = r.upload(scene)
    assert info.prim_count == 1048577
    o = oracle.OracleScene(scene)
    big = helpers.primary_rays(o, 3840, 2160, 1 << 20, seed=77)
    got = r.intersect(big)
    hit = got["prim_id"] >= 0
    assert 0.3 < hit.mean() <= 1.0
    assert np.all(got["t"][hit] >= 1e-4) and np.all(np.isfinite(got["t"][hit]))
    assert np.all((got["u"][hit] >= 0) & (got["u"][hit] <= 1) & (got["v"][hit] >= 0) & (got["u"][hit] + got["v"][hit] <= 1.0 + 1e-6) | (got["prim_id"][hit] == 1048576))
    assert np.array_equal(got, r.intersect(big))  # deterministic
    # shrinking t_max to the found t still finds the same primitive (inclusive upper bound), shrinking below loses it
    sub = np.nonzero(hit)[0][:4096]
    again = r.intersect(big[sub], t_max=float(got["t"][sub].max()))
    assert np.array_equal(again["prim_id"], got["prim_id"][sub])
    # exhaustive oracle on a subset (1M triangles x 1536 rays)
    pick = np.random.default_rng(3).choice(len(big), 1024, replace=False)
    bounce = helpers.bounce_rays(o, big[pick[:512]], seed=5)
    sample = np.concatenate([big[pick], bounce])
    helpers.assert_hits_match(r.intersect(sample), o.intersect(sample))
    r.close()


def test_api_errors(rrt):
    r = rrt.Renderer()
    with pytest.raises(rrt.RrtError) as e:
        r.intersect(np.zeros(4, rrt.RAY_DTYPE))
    assert e.value.status == 7  # RRT_ERR_STATE: no scene uploaded
    with pytest.raises(rrt.RrtError) as e:
        r.render(8, 8, 1)
    assert e.value.status == 7
    s = rrt.Scene()
    s.add_material(rrt.Lambertian(0.5, 0.5, 0.5))
    s.add_sphere((0, 0, 0), 1.0, 0)
    r.upload(s)
    for w, h, spp in ((0, 8, 1), (8, 0, 1), (8, 8, 0)):
        with pytest.raises(rrt.RrtError) as e:
            r.render(w, h, spp)
        assert e.value.status == 1
    bad = rrt.Scene()
    with pytest.raises(rrt.RrtError):
        r.upload(bad)  # no materials
    assert r.intersect(helpers.make_rays([(0, 0, 5)], [(0, 0, -1)]))["prim_id"][0] == 0  # the old scene is still there
    with pytest.raises(rrt.RrtError) as e:
        rrt.Renderer(device=99)
    assert e.value.status == 1
    with pytest.raises(rrt.RrtError):
        rrt.Renderer(max_depth=0)
    r.close()


def test_streaming_batch_sizes_order_and_counters(rrt, oracle, renderer_cache, monkeypatch):
    """The batch kernel hands ray indices to wavefronts in blocks of 512 and to lanes as they fall idle: results must
    not depend on the batch size relative to the block, on the order of the rays, or on the refill threshold."""
    r, scene = get(rrt, renderer_cache, "synthetic")
    o = oracle.OracleScene(scene)
    base = helpers.primary_rays(o, 160, 90, 4096, seed=9)
    want = o.intersect(base)
    for n in (511, 512, 513, 1023, 1025, 4096):
        helpers.assert_hits_match(r.intersect(base[:n]), want[:n])
    perm = np.random.default_rng(4).permutation(len(base))
    assert np.array_equal(r.intersect(base[perm]), r.intersect(base)[perm])
    # many more rays than the resident grid holds lanes: every wavefront draws several blocks
    big = np.tile(base, 700)[: 512 * 5120 + 7]
    got = r.intersect(big)
    ref = np.tile(r.intersect(base), 700)[: len(big)]
    assert np.array_equal(got, ref)
    # refill threshold 1 (every idle lane refills at once) and 64 (a wave only refills when it is empty)
    for thr in (1, 64):
        r.set_tuning(refill_min=thr)
        assert np.array_equal(r.intersect(base), ref[: len(base)])
    r.set_tuning(refill_min=None)
    # the plain one-lane-per-ray kernel is the same function of the rays
    r.set_tuning(intersect_static=1)
    assert np.array_equal(r.intersect(base), ref[: len(base)])
    r.set_tuning(intersect_static=None)
    # counting variant: same hits, counters consistent with the walk
    rc = rrt.Renderer(count_visits=True)
    rc.upload(scene)
    assert np.array_equal(rc.intersect(base), ref[: len(base)])
    st = rc.render_stats()
    assert st.rays == len(base) and st.samples == 0
    assert st.node_visits >= st.rays and st.node_wave_steps * 64 >= st.node_visits
    assert st.prim_wave_steps * 64 >= st.prim_tests > 0 and st.refill_wave_passes > 0
    rc.close()


@pytest.mark.parametrize("device_build", [False, True])
def test_random_soups_with_hostile_coordinates(rrt, oracle, monkeypatch, device_build):
    """The box test is the only inexact step of the walk (child boxes live on a 16-bit grid, plane distances come from
    one fma with a hardware reciprocal); it must never cull a hit the exhaustive search finds.  Triangle soups far from
    the origin, tiny or huge, flat in one axis, with rays along the axes, from inside and from far outside."""
    monkeypatch.setenv("RRT_DEVICE_BUILD_MIN", "2")
    g = np.random.default_rng(2024)
    cases = [(1.0, 0.0, None), (1e-3, 0.0, None), (1e4, 0.0, None), (1.0, 1e4, None), (1e-2, 3e3, None), (50.0, -2e5, None),
             (1.0, 0.0, 0), (1.0, 0.0, 1), (10.0, 500.0, 2), (1.0, 7.0, None)]
    for scale, offset, flat_axis in cases:
        n = 1500
        centres = g.uniform(-1, 1, (n, 1, 3)) * scale
        tris = centres + g.normal(0, 0.03 * scale, (n, 3, 3))
        if flat_axis is not None:
            tris[..., flat_axis] = 0.25 * scale  # every triangle in one plane: boxes of zero thickness
            tris[n // 2:, 1, :], tris[n // 2:, 2, :] = tris[n // 2:, 2, :].copy(), tris[n // 2:, 1, :].copy()  # both facings
        tris = (tris + offset).astype(np.float32)
        s = rrt.Scene()
        m = s.add_material(rrt.Lambertian(0.5, 0.5, 0.5))
        s.add_triangles(tris, m)
        s.add_sphere((offset + 0.3 * scale, offset, offset - 0.2 * scale), 0.2 * scale, m)
        s.camera = rrt.Camera(1.0, 40.0, np.float32([[1, 0, 0, offset], [0, 1, 0, offset], [0, 0, 1, offset + 4 * scale]]))
        r = rrt.Renderer(device_build=device_build)
        r.upload(s)
        k = 1024
        inside = g.uniform(-1, 1, (k, 3)) * scale + offset
        outside = g.normal(0, 1, (k, 3))
        outside = outside / np.linalg.norm(outside, axis=1, keepdims=True) * 30 * scale + offset
        target = g.uniform(-1, 1, (2 * k, 3)) * scale + offset
        origins = np.concatenate([inside, outside])
        dirs = target - origins
        dirs[:k] = g.normal(0, 1, (k, 3))
        axis = np.zeros((384, 3))
        axis[np.arange(384), g.integers(0, 3, 384)] = g.choice([-1.0, 1.0], 384)
        axis_o = tris[g.integers(0, n, 384)].mean(1) - axis * 3 * scale + g.normal(0, 0.01 * scale, (384, 3))  # along an axis, through a triangle
        rays = helpers.make_rays(np.concatenate([origins, axis_o]).astype(np.float32), np.concatenate([dirs, axis]).astype(np.float32))
        got, want = r.intersect(rays, t_min=1e-4 * scale), oracle.OracleScene(s).intersect(rays, t_min=1e-4 * scale)
        assert (want["prim_id"] >= 0).mean() > 0.02, (scale, offset, flat_axis)  # (tiny triangles fail the reference's det > 1e-5)
        helpers.assert_hits_match(got, want)
        r.close()


def test_unrolled_scene_with_more_instances_than_the_kernels_keep_in_lds(rrt, oracle):
    """The kernels keep the transforms of up to 127 instances in LDS and read the table from device memory beyond that:
    150 instances of a small mesh, hits and image against the two-level form and the oracle."""
    from rust_ray_tracer_amd import scenes

    g = np.random.default_rng(5)
    s = rrt.Scene()
    grey = s.add_material(rrt.Lambertian(0.6, 0.6, 0.6))
    light = s.add_material(rrt.DiffuseLight(200.0, 200.0, 200.0))
    mesh = s.new_mesh()
    s.add_triangles(scenes.blob_mesh(5, 6, 0.35, 0.3, seed=2), grey, mesh=mesh)
    mi = s.finish_mesh(mesh)
    for k in range(150):
        s.add_instance(mi, angle_deg=float(g.uniform(0, 360)), offset=(float((k % 15) - 7) * 0.9, float(g.uniform(-0.3, 0.3)), float((k // 15) - 5) * 0.9))
    s.add_sphere((0.0, 6.0, 0.0), 0.5, light)
    s.add_sphere_light((0.0, 6.0, 0.0), 0.5)
    s.camera = rrt.Camera(16.0 / 9.0, 45.0, scenes.look_at_c2w((0.0, 5.0, 9.0), (0.0, 0.0, 0.0)))
    s.background = (0.05, 0.05, 0.06)
    a, b = rrt.Renderer(seed=3), rrt.Renderer(seed=3, unroll_instances=False)
    assert a.upload(s).unrolled_instances == 150 and b.upload(s).unrolled_instances == 0
    o = oracle.OracleScene(s)
    for kind, rays in helpers.ray_suite(s, o, n=2048, seed=8).items():
        ga, gb = a.intersect(rays), b.intersect(rays)
        assert np.array_equal(ga, gb), kind
        helpers.assert_hits_match(ga, o.intersect(rays))
    ia, sa = a.render(128, 72, 4)
    ib, sb = b.render(128, 72, 4)
    assert sa.rays == sb.rays and np.array_equal(ia, ib)
    a.close()
    b.close()


def test_unrolled_instances_with_hostile_coordinates(rrt, oracle):
    """Unrolled instances put world-space boxes around primitives that are tested in object space, with the ray moved
    there in f32: the boxes' pad has to cover the rounding of that move.  Small meshes far from the origin, far from
    each other and rotated by awkward angles, rays from inside, from far outside and along the axes; the two-level
    form on the same rays must give the same records, and both the oracle's."""
    g = np.random.default_rng(77)
    for scale, spread in ((1.0, 30.0), (0.05, 2000.0), (20.0, 1e4), (1e-2, 5.0), (3.0, 150.0)):
        s = rrt.Scene()
        m = s.add_material(rrt.Lambertian(0.5, 0.5, 0.5))
        mesh = s.new_mesh()
        n = 600
        centres = g.uniform(-1, 1, (n, 1, 3)) * scale
        tris = (centres + g.normal(0, 0.05 * scale, (n, 3, 3))).astype(np.float32)
        s.add_triangles(tris, m, mesh=mesh)
        # ... and the other primitives a mesh may hold: their leaves are unrolled the same way
        # (a sphere only where the reference's own test can see its radius: sphere.rs:56-58 forms |o - c|^2 - r^2 and
        # b^2 - a c in f32, and from more than a few thousand radii away r^2 is below the rounding of |o - c|^2 -- the
        # test's answer is then rounding noise, "hits" up to many radii beside the sphere included, which no box covers)
        n_other = 7
        if spread / scale <= 100.0:
            s.add_sphere((0.3 * scale, -0.2 * scale, 0.1 * scale), 0.4 * scale, m, mesh=mesh)
            n_other = 8
        s.add_cuboid((-1.2 * scale, -1.2 * scale, -1.2 * scale), (-0.6 * scale, 0.9 * scale, -0.8 * scale), m, mesh=mesh)
        s.add_rect(1, -scale, scale, -scale, scale, -1.3 * scale, m, mesh=mesh)
        mi = s.finish_mesh(mesh)
        offsets = g.uniform(-1, 1, (10, 3)) * spread
        angles = [0.0, 90.0, 180.0, 270.0, 45.0, 1e-3, 359.999, 33.3, 123.4, 271.7]
        for off, ang in zip(offsets, angles):
            s.add_instance(mi, angle_deg=ang, offset=tuple(float(x) for x in off))
        s.add_triangles((g.normal(0, 1, (4, 3, 3)) * spread).astype(np.float32), m)  # a few huge top-level triangles
        s.camera = rrt.Camera(1.0, 40.0, np.float32([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 3 * spread]]))
        a, b = rrt.Renderer(), rrt.Renderer(unroll_instances=False)
        assert a.upload(s).unrolled_instances == 10 and b.upload(s).unrolled_instances == 0
        k = 512
        which = g.integers(0, 10, 3 * k)
        target = offsets[which] + g.uniform(-1, 1, (3 * k, 3)) * scale
        inside = offsets[which[:k]] + g.uniform(-1, 1, (k, 3)) * scale
        far = g.normal(0, 1, (k, 3))
        far = far / np.linalg.norm(far, axis=1, keepdims=True) * 4 * spread
        axis = np.zeros((k, 3))
        axis[np.arange(k), g.integers(0, 3, k)] = g.choice([-1.0, 1.0], k)
        origins = np.concatenate([inside, far, target[2 * k:] - axis * 3 * scale])
        dirs = np.concatenate([g.normal(0, 1, (k, 3)), target[k:2 * k] - far, axis])
        rays = helpers.make_rays(origins.astype(np.float32), dirs.astype(np.float32))
        ga, gb, want = a.intersect(rays, t_min=1e-4 * scale), b.intersect(rays, t_min=1e-4 * scale), oracle.OracleScene(s).intersect(rays, t_min=1e-4 * scale)
        assert (want["prim_id"] >= 0).mean() > 0.02, (scale, spread)
        helpers.assert_hits_match(ga, want)
        helpers.assert_hits_match(gb, want)
        assert ga.tobytes() == gb.tobytes(), (scale, spread)
        hit_ids = want["prim_id"][want["prim_id"] >= 0]
        in_inst = hit_ids[hit_ids < 10 * (n + n_other)]
        assert np.isin(in_inst % (n + n_other), np.arange(n, n + n_other)).sum() > 20  # spheres / rectangles of the instances are hit
        a.close()
        b.close()


def test_scene_fed_to_both_sides_is_itself_checked_on_the_box(rrt, oracle):
    """The oracle and the GPU path both consume the scene the product's glTF loader emits, so a loader error would be
    invisible to the comparisons above.  Here the loader's triangles are compared, on the box that runs the GPU
    gates, with an independent reading of the same file (json + base64 + numpy, following reference src/gltf.rs:118-244),
    and that independent scene -- built without the loader -- must give the same closest hits as the loaded one."""
    import base64
    import json
    import os

    path = os.path.join(helpers.ASSETS, "suzanne.gltf")
    d = json.load(open(path))
    bufs = [base64.b64decode(b["uri"].split(",")[1]) for b in d["buffers"]]
    views = [bufs[v["buffer"]][v["byteOffset"]:v["byteOffset"] + v["byteLength"]] for v in d["bufferViews"]]
    tris = []
    for mesh in d["meshes"]:
        p = mesh["primitives"][0]
        ia, pa = d["accessors"][p["indices"]], d["accessors"][p["attributes"]["POSITION"]]
        idx = np.frombuffer(views[ia["bufferView"]], np.uint16, ia["count"]).astype(np.int64)
        pos = np.frombuffer(views[pa["bufferView"]], np.float32, pa["count"] * 3).reshape(-1, 3)
        tris.append(pos[idx[:3 * (len(idx) // 3)]].reshape(-1, 3, 3))
    want = np.concatenate(tris)
    loaded = helpers.scene_suzanne(rrt)
    t = np.concatenate(loaded._tri_chunks)
    assert np.array_equal(np.stack([t["v0"], t["v1"], t["v2"]], 1), want)
    # the same scene assembled by hand: triangles from the independent reading, light and camera as documented
    # (reference src/gltf.rs:287-299: radius 0.2 emissive sphere at the light node's translation)
    s = rrt.Scene()
    m = s.add_material(loaded.materials[0])
    s.add_triangles(want, m)
    lm = s.add_material(loaded.materials[loaded.spheres[0][2]])
    s.add_sphere(loaded.spheres[0][0], 0.2, lm)
    s.add_sphere_light(loaded.spheres[0][0], 0.2)
    s.camera = loaded.camera
    s.background = (0.051, 0.051, 0.051)
    assert abs(loaded.spheres[0][0][0] - 4.076245) < 1e-5 and abs(loaded.spheres[0][1] - 0.2) < 1e-7
    a, b = rrt.Renderer(seed=3), rrt.Renderer(seed=3)
    a.upload(loaded)
    b.upload(s)
    o = oracle.OracleScene(s)
    rays = np.concatenate(list(helpers.ray_suite(s, o, n=2048, seed=41).values()))
    ha, hb = a.intersect(rays), b.intersect(rays)
    assert ha.tobytes() == hb.tobytes()
    helpers.assert_hits_match(hb, o.intersect(rays))
    ia, _ = a.render(96, 54, 4)
    ib, _ = b.render(96, 54, 4)
    assert np.array_equal(ia, ib)
    a.close()
    b.close()
