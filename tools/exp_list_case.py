#!/usr/bin/env python3
"""One case of tools/stress_lists.py in detail: which pixels differ between the walked frame, the two forms with lists
and the oracle's frame, and by how much.   python tools/exp_list_case.py --seed 101 --case 7003"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--case", type=int, required=True)
    args = ap.parse_args()
    import numpy as np
    import rrt_pkg
    from stress_lists import random_flat_scene
    from oracle import oracle

    rrt = rrt_pkg.load()
    g = np.random.default_rng(args.seed)
    for case in range(args.case + 1):
        s, what = random_flat_scene(rrt, np, g)
        w, h = int(g.integers(1, 91)), int(g.integers(1, 91))
        spp, chunks = (16, 8) if g.integers(0, 2) else (37, 16)
        case_seed = int(g.integers(1, 1 << 30))
    print(what, w, h, spp, chunks, flush=True)
    r = rrt.Renderer(seed=case_seed)
    r.upload(s)
    frames = {}
    for name, kw in (("walk pipelined", dict(overlap_items=1, pixel_lists=0)), ("walk one-item", dict(overlap_items=0, pixel_lists=0)),
                     ("lists pipelined 7", dict(overlap_items=1, pixel_lists=1, list_waves=7)),
                     ("lists pipelined 6", dict(overlap_items=1, pixel_lists=1, list_waves=6)),
                     ("lists one-item", dict(overlap_items=0, pixel_lists=1))):
        r.set_tuning(split_items=1, chunks=chunks, **kw)
        img, st = r.render(w, h, spp)
        frames[name] = img.copy()
        print(f"{name}: rays {st.rays}, listed {r.info().pixel_list_permille}", flush=True)
    ref, ref_rays = oracle.OracleScene(s).render(w, h, spp, seed=case_seed)
    print("oracle rays", ref_rays)
    frames["oracle"] = ref
    base = frames["walk pipelined"]
    pix = set()
    for name, img in frames.items():
        d = np.argwhere((img != base).any(axis=-1)) if name != "oracle" else np.argwhere(~np.isclose(img, base, rtol=1e-4, atol=1e-5).all(axis=-1))
        for y, x in d:
            pix.add((int(y), int(x)))
    for y, x in sorted(pix):
        print(f"pixel row {y} col {x}:")
        for name, img in frames.items():
            print(f"   {name:20s} {img[y, x]}")
    r.close()


if __name__ == "__main__":
    main()
