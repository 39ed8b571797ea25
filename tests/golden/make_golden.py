#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (oracle/rrt_oracle.c).

    python tests/golden/make_golden.py

The reference ships no golden vectors and cannot be built here, so these fixtures pin the ORACLE (a regression
guard) and give the HIP path a fixed target that needs no oracle run: ray batches (primary, bounce, adversarial)
with the closest-hit answers, and two small accumulation images.  Scene inputs are the bundled glTF assets and the
deterministic scene builders in tests/helpers.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
import rrt_pkg  # noqa: E402
from oracle import oracle  # noqa: E402

RAY_SCENES = ["default", "suzanne", "mixed", "instanced", "cornell"]
IMAGE_CASES = [("default", 64, 36, 4, 5), ("mixed", 64, 36, 4, 9)]  # scene, width, height, spp, seed


def main():
    rrt = rrt_pkg.load()
    for name in RAY_SCENES:
        scene = helpers.SCENES[name](rrt)
        o = oracle.OracleScene(scene)
        suite = helpers.ray_suite(scene, o, n=384, seed=21)
        out = {}
        for kind, rays in suite.items():
            out[f"{kind}_rays"] = rays
            out[f"{kind}_hits"] = o.intersect(rays)
        np.savez_compressed(os.path.join(HERE, f"rays_{name}.npz"), **out)
        print(name, {k: len(v) for k, v in suite.items()})
    for name, w, h, spp, seed in IMAGE_CASES:
        scene = helpers.SCENES[name](rrt)
        o = oracle.OracleScene(scene)
        accum, rays = o.render(w, h, spp, seed=seed)
        np.savez_compressed(os.path.join(HERE, f"image_{name}.npz"), accum=accum, rays=np.int64(rays), spp=np.int32(spp),
                            seed=np.int32(seed), rgb8=oracle.tonemap(accum, spp))
        print(name, "image", accum.mean(), rays)


if __name__ == "__main__":
    main()
