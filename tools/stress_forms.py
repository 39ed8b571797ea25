#!/usr/bin/env python3
"""Random scenes and cameras: every form of the render kernel must give the same image, bit for bit.
    python tools/stress_forms.py [--cases 300] [--seed 1]      (on the GPU box)
Flat scenes (triangle soups of very different sizes with spheres and rectangles): the convert and the scaled form of
the box arithmetic, one item at a time and two in flight, whole tiles and items of a few pixels x many chunks.
Instanced scenes (meshes of triangles, spheres and cuboids under Translate(RotateY)): the unrolled and the two-level
form.  Every fifth scene: rrt_render_device_noalloc against rrt_render_device, and a group of two contexts against the
sum of its two sample slices.  Cameras inside, touching and far away, fields of view of 0.5 .. 175 degrees."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def camera(rrt, np, g, centre, scale):
    rot, _ = np.linalg.qr(g.normal(0, 1, (3, 3)))
    dist = scale * float(g.choice([0.0, 0.01, 0.5, 1.5, 4.0, 40.0]))
    cam_len = float(10 ** g.uniform(-2, 2))
    pos = centre + rot[:, 2] * dist
    c2w = np.concatenate([rot * cam_len, pos[:, None]], axis=1).astype(np.float32)
    return rrt.Camera(float(g.uniform(0.5, 2.0)), float(g.choice([0.5, 5.0, 40.0, 90.0, 150.0, 175.0])), c2w)


def random_scene(rrt, np, g, instanced):
    """A random scene (with instanced meshes if asked) under a random camera; returns (scene, triangles drawn, scale)."""
    scale = 10 ** g.uniform(-2, 3)
    offset = g.uniform(-1, 1, 3) * scale * 10 ** g.uniform(-1, 2)
    s = rrt.Scene()
    mats = [s.add_material(rrt.Lambertian(0.7, 0.6, 0.5)), s.add_material(rrt.Metal((0.8, 0.8, 0.9), 0.1)),
            s.add_material(rrt.Dielectric(1.5)), s.add_material(rrt.DiffuseLight(3.0, 3.0, 3.0))]
    n = int(10 ** g.uniform(1.3, 3.7 if instanced else 4.3))
    size = scale * 10 ** g.uniform(-2.5, 0.3, (n, 1, 1))
    tris = g.uniform(-1, 1, (n, 1, 3)) * scale + g.normal(0, 1, (n, 3, 3)) * size
    if instanced:
        # a mesh around its own origin, instanced a few times around `offset`
        mesh = s.new_mesh()
        s.add_triangles(tris.astype(np.float32) * 0.3, mats[int(g.integers(0, 3))], mesh=mesh)
        if g.integers(0, 2):
            s.add_sphere(tuple(g.uniform(-0.3, 0.3, 3) * scale), float(0.1 * scale), mats[int(g.integers(0, 3))], mesh=mesh)
        if g.integers(0, 2):
            lo = g.uniform(-0.3, 0.0, 3) * scale
            s.add_cuboid(tuple(lo), tuple(lo + g.uniform(0.05, 0.3, 3) * scale), mats[0], mesh=mesh)
        m = s.finish_mesh(mesh)
        for _ in range(int(g.integers(1, 6))):
            s.add_instance(m, float(g.uniform(-180, 180)), tuple(offset + g.uniform(-1, 1, 3) * scale))
        s.add_triangles((tris[: max(1, n // 8)] + offset).astype(np.float32), mats[1])
    else:
        tris = (tris + offset).astype(np.float32)
        cut = sorted(g.integers(0, n + 1, 2))
        s.add_triangles(tris[: cut[0]], mats[0])
        s.add_triangles(tris[cut[0]: cut[1]], mats[1])
        s.add_triangles(tris[cut[1]:], mats[2])
        for _ in range(int(g.integers(0, 4))):
            s.add_sphere(tuple(offset + g.uniform(-1, 1, 3) * scale), float(scale * 10 ** g.uniform(-2, -0.3)), mats[int(g.integers(0, 4))])
    light_c = offset + np.float32([0, 2.5 * scale, 0])
    s.add_sphere(tuple(light_c), float(0.3 * scale), mats[3])
    s.add_sphere_light(tuple(light_c), float(0.3 * scale))
    s.background = (0.5, 0.6, 0.8)
    s.camera = camera(rrt, np, g, offset, scale)
    return s, n, scale


def run(cases, seed, verbose=True):
    import numpy as np
    import rrt_pkg

    rrt = rrt_pkg.load()
    g = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        instanced = case % 3 == 2
        s, n, scale = random_scene(rrt, np, g, instanced)
        w, h = int(g.integers(1, 91)), int(g.integers(1, 91))
        spp, chunks = (12, 4) if g.integers(0, 2) else (37, 16)
        case_seed = int(g.integers(1, 1 << 30))
        r = rrt.Renderer(seed=case_seed)
        info = r.upload(s)
        r.set_tuning(overlap_items=0, split_items=0, chunks=chunks, scaled_slab=0, pixel_lists=0)
        base, s0 = r.render(w, h, spp)
        forms = [dict(overlap_items=1), dict(scaled_slab=1), dict(overlap_items=1, scaled_slab=1)]
        if not instanced:
            forms += [dict(split_items=1), dict(split_items=1, overlap_items=1, scaled_slab=1)]
        for knobs in forms:
            r.set_tuning(**{**dict(overlap_items=0, split_items=0, scaled_slab=0), **knobs})
            img, s1 = r.render(w, h, spp)
            if not (s0.rays == s1.rays and np.array_equal(base, img)):
                bad += 1
                print(f"MISMATCH case {case}: {knobs} instanced {instanced} n {n} scale {scale:.3g} {w}x{h} spp {spp} scaled_ok {info.scaled_slab}: rays {s0.rays} / {s1.rays}, "
                      f"{int((base != img).any(axis=-1).sum())} pixels differ", flush=True)
        # every fifth scene: the entry point that allocates nothing, and a group of two contexts on this GPU whose
        # members render half the samples each
        if case % 5 == 0:
            import helpers
            r.set_tuning(overlap_items=None, split_items=None, scaled_slab=None, pixel_lists=None, chunks=None)
            a, b = helpers.DeviceArray((h, w, 3)), helpers.DeviceArray((h, w, 3))
            r.render_device(w, h, spp, 3, a.ptr, 0)
            sa = r.render_stats()
            r.render_device_noalloc(w, h, spp, 3, b.ptr, 0)
            sb = r.render_stats()
            if not (sa.rays == sb.rays and sa.chunks == sb.chunks and np.array_equal(a.numpy(), b.numpy())):
                bad += 1
                print(f"MISMATCH case {case}: render_device_noalloc against render_device, instanced {instanced} n {n} {w}x{h} spp {spp}: rays {sa.rays} / {sb.rays}", flush=True)
            whole = a.numpy().copy()
            a.free()
            b.free()
            if spp >= 2:
                n_first = spp - spp // 2  # (the remainder of an uneven split goes to the first members)
                first, s_first = r.render(w, h, n_first, spp_offset=3)
                second, s_second = r.render(w, h, spp - n_first, spp_offset=3 + n_first)
                grp = rrt.RendererGroup([0, 0], seed=case_seed)
                grp.upload(s)
                summed, s_group = grp.render(w, h, spp, spp_offset=3)
                grp.close()
                if not (s_group.rays == s_first.rays + s_second.rays == sa.rays and np.array_equal(summed, first + second)):
                    bad += 1
                    print(f"MISMATCH case {case}: group of two against the sum of its slices, instanced {instanced} n {n} {w}x{h} spp {spp}: rays {s_group.rays} / {s_first.rays + s_second.rays} / {sa.rays}", flush=True)
        r.close()
        if instanced:
            r2 = rrt.Renderer(seed=case_seed, unroll_instances=False)
            r2.upload(s)
            r2.set_tuning(overlap_items=0, chunks=chunks)
            img, s1 = r2.render(w, h, spp)
            if not (s0.rays == s1.rays and np.array_equal(base, img)):
                bad += 1
                print(f"MISMATCH case {case}: two-level against unrolled ({info.unrolled_instances} instances), n {n} scale {scale:.3g} {w}x{h} spp {spp}: rays {s0.rays} / {s1.rays}, "
                      f"{int((base != img).any(axis=-1).sum())} pixels differ", flush=True)
            r2.close()
        if verbose and case % 25 == 24:
            print(f"{case + 1} cases, {bad} mismatches", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    sys.exit(1 if run(args.cases, args.seed) else 0)


if __name__ == "__main__":
    main()
