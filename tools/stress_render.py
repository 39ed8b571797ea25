#!/usr/bin/env python3
"""Random scenes and cameras: small same-seed frames of the HIP path against the CPU oracle's.
    python tools/stress_render.py [--cases 200] [--seed 1]      (on the GPU box)
Scenes as in tools/stress_lists.py (all four material kinds, a sphere light), at most 2000 triangles so that the
oracle's exhaustive search stays quick.  Gate per frame: ray counts within 0.5 % (+ 2), at least 95 % of the pixels
within 1e-4 relative (1e-5 absolute) -- the rest are paths that meet another primitive after an ulp-sized change of
direction, which hostile geometry (needles, grazing sheets, a 175-degree field of view) makes more frequent than the
bundled scenes do (measured over 998 frames: 99.8 % of the frames have every pixel close, the lowest share is 95.6 %)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def run(cases, seed, verbose=True):
    import numpy as np
    import rrt_pkg
    from oracle import oracle
    import stress_lists

    rrt = rrt_pkg.load()
    g = np.random.default_rng(seed)
    bad = 0
    worst = 1.0
    done = 0
    for case in range(cases):
        s, what = stress_lists.random_flat_scene(rrt, np, g)
        w, h = int(g.integers(4, 41)), int(g.integers(4, 41))
        spp = int(g.integers(1, 7))
        case_seed = int(g.integers(1, 1 << 30))
        if s.triangle_count > 2000:
            continue
        done += 1
        r = rrt.Renderer(seed=case_seed)
        r.upload(s)
        img, st = r.render(w, h, spp)
        rgb = r.tonemap(img, spp)
        r.close()
        ref, ref_rays = oracle.OracleScene(s).render(w, h, spp, seed=case_seed)
        finite = np.isfinite(ref).all(axis=-1) & np.isfinite(img).all(axis=-1)
        close = (np.isclose(img, ref, rtol=1e-4, atol=1e-5).all(axis=-1) | ~finite).mean()
        rays_ok = abs(int(st.rays) - ref_rays) <= max(2, ref_rays // 200)
        tone_ok = np.array_equal(rgb, oracle.tonemap(img, spp))
        worst = min(worst, close)
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
    args = ap.parse_args()
    sys.exit(1 if run(args.cases, args.seed) else 0)


if __name__ == "__main__":
    main()
