#!/usr/bin/env python3
"""Random scenes and cameras: small same-seed frames of the HIP path against the CPU oracle's.
    python tools/stress_render.py [--cases 200] [--seed 1] [--flat-only] [--only CASE]      (on the GPU box)
Scenes as in tools/stress_lists.py and, every third one, tools/stress_forms.py's instanced ones (all four material
kinds, a sphere light; unrolled and two-level in turn), at most 2000 triangles so that the oracle's exhaustive search
stays quick.  Gate per frame: ray counts within 0.5 % (+ 2), at least 95 % of the pixels within 1e-4 relative (1e-5
absolute), the tone map bit-exact.  The pixels outside are paths that meet another primitive after an ulp-sized change
of direction (the kernels' 1-ulp reciprocals in shading): frames of one and two rays per path agree in every pixel, and
hostile geometry makes the third and fourth ray chaotic -- measured over 5,000 frames: 99.8 % of them have every pixel
close; the ones below the gate (lowest 89 %) are soups of interpenetrating triangles at coordinates of several hundred,
where the reference's fixed t_min = 1e-4 (src/ray.rs:89) is about an ulp of a hit point and self-intersection is noise."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def run(cases, seed, verbose=True, flat_only=False, only=-1, lists=False):
    import numpy as np
    import rrt_pkg
    from oracle import oracle
    import stress_forms
    import stress_lists

    rrt = rrt_pkg.load()
    g = np.random.default_rng(seed)
    bad = 0
    worst = 1.0
    done = 0
    for case in range(cases):
        if case % 3 == 2 and not flat_only:  # meshes of triangles, a sphere, a cuboid under Translate(RotateY), unrolled and two-level in turn
            s, n, scale = stress_forms.random_scene(rrt, np, g, True)
            what = f"instanced n {n} scale {scale:.3g}"
        else:
            s, what = stress_lists.random_flat_scene(rrt, np, g)
        w, h = int(g.integers(4, 41)), int(g.integers(4, 41))
        spp = int(g.integers(1, 7))
        if lists:
            spp = 8  # one sample in each of the 8 chunks an item with lists needs
        case_seed = int(g.integers(1, 1 << 30))
        if s.triangle_count > 2000 or (only >= 0 and case != only):
            continue
        done += 1
        r = rrt.Renderer(seed=case_seed, unroll_instances=case % 2 == 0)
        r.upload(s)
        if lists:  # primary rays from per-pixel candidate lists (scenes without instances and unrolled ones)
            r.set_tuning(split_items=1, chunks=8, pixel_lists=1, overlap_items=case % 4 < 2)
        img, st = r.render(w, h, spp)
        rgb = r.tonemap(img, spp)
        if lists and only >= 0:  # the same frame with primary rays that walk
            listed_share = r.info().pixel_list_permille / 10.0
            r.set_tuning(pixel_lists=0)
            walked, sw = r.render(w, h, spp)
            print(f"case {case}: {listed_share:.1f} % of the pixels listed; walked frame: rays {sw.rays} (with lists {st.rays}), {int((walked != img).any(axis=-1).sum())} pixels differ from the frame with lists")
        r.close()
        ref, ref_rays = oracle.OracleScene(s).render(w, h, spp, seed=case_seed)
        finite = np.isfinite(ref).all(axis=-1) & np.isfinite(img).all(axis=-1)
        close = (np.isclose(img, ref, rtol=1e-4, atol=1e-5).all(axis=-1) | ~finite).mean()
        rays_ok = abs(int(st.rays) - ref_rays) <= max(2, ref_rays // 200)
        tone_ok = np.array_equal(rgb, oracle.tonemap(img, spp))
        worst = min(worst, close)
        if only >= 0:
            off = ~np.isclose(img, ref, rtol=1e-4, atol=1e-5).all(axis=-1)
            rel = np.abs(img - ref) / np.maximum(np.abs(ref), 1e-5)
            print(f"case {case}: {what}: {int(off.sum())} pixels off; largest relative difference {rel.max():.3g}, median over the pixels off {np.median(rel.max(axis=-1)[off]) if off.any() else 0:.3g}")
            for y, x in np.argwhere(off)[:6]:
                print("   pixel", x, y, "gpu", img[y, x], "oracle", ref[y, x])
        if not (rays_ok and close >= 0.95 and tone_ok and finite.mean() > 0.99):
            bad += 1
            print(f"DIFF case {case}: {what} {w}x{h} spp {spp}: rays {st.rays} / {ref_rays}, pixels close {close:.4f}, finite {finite.mean():.4f}, tone map {'ok' if tone_ok else 'DIFFERS'}", flush=True)
        if verbose and case % 25 == 24:
            print(f"{case + 1} cases ({done} rendered), {bad} outside the gate, lowest share of close pixels {worst:.4f}", flush=True)
    print(f"done: {cases} cases ({done} rendered), {bad} outside the gate, lowest share of close pixels {worst:.4f}")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lists", action="store_true", help="8 samples in 8 chunks, items of 8 pixels, primary rays from per-pixel candidate lists")
    ap.add_argument("--flat-only", action="store_true", help="no instanced scenes")
    ap.add_argument("--only", type=int, default=-1, help="render this case only and print its differing pixels")
    args = ap.parse_args()
    sys.exit(1 if run(args.cases, args.seed, flat_only=args.flat_only, only=args.only, lists=args.lists) else 0)


if __name__ == "__main__":
    main()
