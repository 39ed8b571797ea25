#!/usr/bin/env python3
"""Random scenes and ray batches: rrt_intersect (the BVH walk on the GPU) against the oracle's exhaustive search.
    python tools/stress_intersect.py [--cases 300] [--seed 1]      (on the GPU box)
Flat and instanced scenes as in tools/stress_forms.py, every fourth flat one large enough for the GPU's BVH builder and
built by it; per scene 4096 rays: camera rays, rays between random points
of the scene, axis-parallel rays, rays that start on a triangle.  Gate: primitive ids equal, t within 1e-5 relative.
Prints every batch with differences and what kind they are (a hit the walk lacks / has in excess / another primitive)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(cases, seed, verbose=True):
    import numpy as np
    import rrt_pkg
    from oracle import oracle

    rrt = rrt_pkg.load()
    g = np.random.default_rng(seed)
    ray_dtype = np.dtype([("origin", np.float32, 3), ("direction", np.float32, 3), ("time", np.float32)])
    bad_cases = 0
    bad_rays = 0
    total = 0
    for case in range(cases):
        scale = 10 ** g.uniform(-2, 3)
        offset = g.uniform(-1, 1, 3) * scale * 10 ** g.uniform(-1, 1.5)
        s = rrt.Scene()
        m0 = s.add_material(rrt.Lambertian(0.7, 0.6, 0.5))
        instanced = case % 3 == 2
        device_build = not instanced and case % 4 == 0  # a tree the GPU builder makes (scenes of >= 4096 primitives)
        n = int(10 ** g.uniform(3.7, 4.6)) if device_build else int(10 ** g.uniform(1.0, 3.5))
        size = scale * 10 ** g.uniform(-2.0, 0.3, (n, 1, 1))
        tris = g.uniform(-1, 1, (n, 1, 3)) * scale + g.normal(0, 1, (n, 3, 3)) * size
        if instanced:
            mesh = s.new_mesh()
            s.add_triangles(tris.astype(np.float32) * 0.3, m0, mesh=mesh)
            if g.integers(0, 2):
                s.add_sphere(tuple(g.uniform(-0.3, 0.3, 3) * scale), float(0.1 * scale), m0, mesh=mesh)
            if g.integers(0, 2):
                lo = g.uniform(-0.3, 0.0, 3) * scale
                s.add_cuboid(tuple(lo), tuple(lo + g.uniform(0.05, 0.3, 3) * scale), m0, mesh=mesh)
            m = s.finish_mesh(mesh)
            for _ in range(int(g.integers(1, 6))):
                s.add_instance(m, float(g.uniform(-180, 180)), tuple(offset + g.uniform(-1, 1, 3) * scale))
            world_tris = (tris[: max(1, n // 8)] + offset).astype(np.float32)
            s.add_triangles(world_tris, m0)
        else:
            world_tris = (tris + offset).astype(np.float32)
            s.add_triangles(world_tris, m0)
            for _ in range(int(g.integers(0, 4))):
                s.add_sphere(tuple(offset + g.uniform(-1, 1, 3) * scale), float(scale * 10 ** g.uniform(-1.5, -0.3)), m0)
            if g.integers(0, 2):
                s.add_rect(int(g.integers(0, 3)), float(offset[0] - scale), float(offset[0] + scale), float(offset[1] - scale), float(offset[1] + scale), float(offset[2]), m0)
        s.camera = rrt.Camera.default()
        nr = 4096
        rays = np.zeros(nr, ray_dtype)
        a = offset + g.uniform(-1.5, 1.5, (nr, 3)) * scale
        b = offset + g.uniform(-1.5, 1.5, (nr, 3)) * scale
        kind = g.integers(0, 4, nr)
        # kind 1: axis-parallel; kind 2: from a point on a triangle; kind 3: from far away towards the scene
        d = b - a
        ax = g.integers(0, 3, nr)
        par = np.zeros((nr, 3)); par[np.arange(nr), ax] = g.choice([-1.0, 1.0], nr) * scale
        d = np.where((kind == 1)[:, None], par, d)
        tri = world_tris[g.integers(0, len(world_tris), nr)]
        bary = g.dirichlet([1, 1, 1], nr)
        on = (tri * bary[:, :, None]).sum(1)
        a = np.where((kind == 2)[:, None], on, a)
        far = offset + g.normal(0, 1, (nr, 3)) * scale * 30
        a = np.where((kind == 3)[:, None], far, a)
        d = np.where((kind == 3)[:, None], b - far, d)
        d = d * (10 ** g.uniform(-2, 2, (nr, 1)))
        rays["origin"], rays["direction"] = a.astype(np.float32), d.astype(np.float32)
        for unroll in ((True, False) if instanced else (True,)):
            r = rrt.Renderer(unroll_instances=unroll, device_build=device_build)
            info = r.upload(s)
            if device_build and not info.device_built:
                print(f"NOTE case {case}: {n} triangles, and the upload fell back to the host builder", flush=True)
            got = r.intersect(rays)
            r.close()
            want = oracle.OracleScene(s).intersect(rays)
            total += nr
            hit = want["prim_id"] >= 0
            same_id = got["prim_id"] == want["prim_id"]
            with np.errstate(invalid="ignore"):  # (misses carry t = inf on both sides)
                t_ok = np.abs(got["t"] - want["t"]) <= 1e-5 * np.abs(want["t"])
            wrong = ~same_id | (hit & same_id & ~t_ok)
            if wrong.any():
                bad_cases += 1
                bad_rays += int(wrong.sum())
                lacks = int((wrong & hit & (got["prim_id"] < 0)).sum())
                excess = int((wrong & ~hit).sum())
                other = int(wrong.sum()) - lacks - excess
                i = int(np.flatnonzero(wrong)[0])
                print(f"DIFF case {case} (instanced {instanced}, unrolled {unroll}, device-built {device_build}, n {n}, scale {scale:.3g}): {int(wrong.sum())} rays: walk lacks {lacks}, has in excess {excess}, "
                      f"other primitive or t {other}; first: kind {int(kind[i])} got {got[i]} want {want[i]} |o-hit|/scale {float(want['t'][i]) * float(np.linalg.norm(d[i])) / scale:.3g}", flush=True)
        if verbose and case % 50 == 49:
            print(f"{case + 1} cases, {total} rays, {bad_cases} batches with differences, {bad_rays} rays", flush=True)
    print(f"done: {cases} cases, {total} rays, {bad_cases} batches with differences, {bad_rays} rays")
    return bad_rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    sys.exit(1 if run(args.cases, args.seed) else 0)


if __name__ == "__main__":
    main()
