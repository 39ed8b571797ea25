#!/usr/bin/env python3
"""Random scenes and cameras: the render with per-pixel candidate lists against the walked one, bit for bit.
    python tools/stress_lists.py [--cases 300] [--seed 1]      (on the GPU box)
Triangle soups of 20 .. 20000 triangles of very different sizes (needles, slivers, triangles larger than the scene),
with spheres and rectangles thrown in, seen from outside, from inside and from touching distance, at fields of view
from 0.5 to 175 degrees, on frames of 1 .. 90 pixels a side; both forms of the kernel's loop over items.
--instanced: the scenes of tools/stress_forms.py with meshes of triangles, spheres and cuboids under Translate(RotateY),
unrolled into one tree (the lists then name a primitive together with its instance)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def random_flat_scene(rrt, np, g):
    """A random scene without instances under a random camera; returns (scene, description of the draw)."""
    n = int(10 ** g.uniform(1.3, 4.3))
    scale = 10 ** g.uniform(-2, 3)
    offset = g.uniform(-1, 1, 3) * scale * 10 ** g.uniform(-1, 2)
    centres = g.uniform(-1, 1, (n, 1, 3)) * scale
    size = scale * 10 ** g.uniform(-2.5, 0.3, (n, 1, 1))
    tris = centres + g.normal(0, 1, (n, 3, 3)) * size
    kind = g.integers(0, 4)
    if kind == 1:  # needles: two corners close together
        tris[:, 1] = tris[:, 0] + (tris[:, 1] - tris[:, 0]) * 1e-3
    elif kind == 2:  # a flat sheet with holes
        tris[..., g.integers(0, 3)] *= 1e-4
    tris = (tris + offset).astype(np.float32)
    s = rrt.Scene()
    mats = [s.add_material(rrt.Lambertian(0.7, 0.6, 0.5)), s.add_material(rrt.Metal((0.8, 0.8, 0.9), 0.1)),
            s.add_material(rrt.Dielectric(1.5)), s.add_material(rrt.DiffuseLight(3.0, 3.0, 3.0))]
    cut = sorted(g.integers(0, n + 1, 2))
    s.add_triangles(tris[: cut[0]], mats[0])
    s.add_triangles(tris[cut[0]: cut[1]], mats[1])
    s.add_triangles(tris[cut[1]:], mats[2])
    for _ in range(int(g.integers(0, 4))):
        s.add_sphere(tuple(offset + g.uniform(-1, 1, 3) * scale), float(scale * 10 ** g.uniform(-2, -0.3)), mats[int(g.integers(0, 4))])
    if g.integers(0, 2):
        k = float(offset[1] - scale)
        s.add_rect(1, float(offset[0] - scale), float(offset[0] + scale), float(offset[2] - scale), float(offset[2] + scale), k, mats[0])
    light_c = offset + np.float32([0, 2.5 * scale, 0])
    s.add_sphere(tuple(light_c), float(0.3 * scale), mats[3])
    s.add_sphere_light(tuple(light_c), float(0.3 * scale))
    s.background = (0.5, 0.6, 0.8)
    # camera: a random rotation, at a random distance (0: inside, at the centre of the soup)
    q = g.normal(0, 1, (3, 3))
    rot, _ = np.linalg.qr(q)
    dist = scale * float(g.choice([0.0, 0.01, 0.5, 1.5, 4.0, 40.0]))
    cam_len = float(10 ** g.uniform(-2, 2))
    pos = offset + rot[:, 2] * dist
    c2w = np.concatenate([rot * cam_len, pos[:, None]], axis=1).astype(np.float32)
    fov = float(g.choice([0.5, 5.0, 40.0, 90.0, 150.0, 175.0]))
    s.camera = rrt.Camera(float(g.uniform(0.5, 2.0)), fov, c2w)
    return s, f"n {n} scale {scale:.3g} kind {kind} dist {dist:.3g} fov {fov} cam_len {cam_len:.3g}"


def run(cases, seed, only=-1, verbose=True, instanced=False):
    import numpy as np
    import rrt_pkg
    import stress_forms

    rrt = rrt_pkg.load()
    g = np.random.default_rng(seed)
    bad = 0
    listed_total = 0
    for case in range(cases):
        if instanced:
            s, n_tris, scale = stress_forms.random_scene(rrt, np, g, True)
            what = f"instanced n {n_tris} scale {scale:.3g}"
        else:
            s, what = random_flat_scene(rrt, np, g)
        w, h = int(g.integers(1, 91)), int(g.integers(1, 91))
        spp, chunks = (16, 8) if g.integers(0, 2) else (37, 16)
        case_seed = int(g.integers(1, 1 << 30))
        if only >= 0 and case != only:
            continue
        r = rrt.Renderer(seed=case_seed)
        r.upload(s)
        r.set_tuning(overlap_items=1, split_items=1, chunks=chunks, pixel_lists=0)
        plain, s0 = r.render(w, h, spp)
        for overlap, waves in ((1, 7), (1, 6), (0, None)):
            r.set_tuning(overlap_items=overlap, pixel_lists=1, list_waves=waves)
            img, s1 = r.render(w, h, spp)
            if not (s0.rays == s1.rays and np.array_equal(plain, img)):
                bad += 1
                print(f"MISMATCH case {case}: {what} {w}x{h} spp {spp} overlap {overlap} waves {waves}: "
                      f"rays {s0.rays} / {s1.rays}, {int((plain != img).any(axis=-1).sum())} pixels differ", flush=True)
        listed_total += r.info().pixel_list_permille
        if only >= 0:
            from oracle import oracle
            ref, ref_rays = oracle.OracleScene(s).render(w, h, spp, seed=case_seed)
            near = lambda a: int((~np.isclose(a, ref, rtol=1e-4, atol=1e-5).all(axis=-1)).sum())
            print(f"case {case}: oracle rays {ref_rays}, walked {s0.rays} ({near(plain)} pixels off the oracle's), with lists {s1.rays} ({near(img)} pixels off)")
        r.close()
        if verbose and case % 25 == 24:
            print(f"{case + 1} cases, {bad} mismatches, mean listed share {listed_total / (case + 1) / 10:.1f} %", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="render this case only, and compare both renders with the oracle's")
    ap.add_argument("--instanced", action="store_true", help="scenes with instanced meshes (unrolled)")
    args = ap.parse_args()
    sys.exit(1 if run(args.cases, args.seed, args.only, instanced=args.instanced) else 0)


if __name__ == "__main__":
    main()
