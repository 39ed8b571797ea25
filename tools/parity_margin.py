"""Prints how closely same-seed GPU renders follow the CPU oracle (the margins behind the tolerances the tests state)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import rrt_pkg  # noqa: E402
from oracle import oracle  # noqa: E402

rrt = rrt_pkg.load()
for name, w, h, spp in [("default", 142, 80, 8), ("suzanne", 160, 90, 4), ("mixed", 96, 54, 8), ("instanced", 96, 54, 4),
                        ("cornell", 64, 64, 8), ("synthetic", 96, 54, 4)]:
    scene = helpers.SCENES[name](rrt)
    r = rrt.Renderer(device=0, seed=11)
    r.upload(scene)
    a, st = r.render(w, h, spp)
    b, rays = oracle.OracleScene(scene).render(w, h, spp, seed=11)
    ok = np.isfinite(a) & np.isfinite(b)
    a, b = np.where(ok, a, 0), np.where(ok, b, 0)
    close4 = np.isclose(a, b, rtol=1e-4, atol=1e-5).all(axis=-1).mean()
    close6 = np.isclose(a, b, rtol=1e-6, atol=1e-7).all(axis=-1).mean()
    print(f"{name:10s} rays {int(st.rays)} vs {rays}  rel_l1 {np.abs(a - b).sum() / np.abs(b).sum():.2e}  "
          f"pixels within 1e-4: {close4:.4f}  within 1e-6: {close6:.4f}", flush=True)
    r.close()
