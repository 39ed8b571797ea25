"""Experiment: how many node visits would a primary ray save if its walk started with the hit distance already known
(e.g. from the same pixel's previous sample)?  Traces the primary rays once, then again with each ray shortened to
just beyond its own hit, and prints nodes and primitive tests per ray of both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import rrt_pkg  # noqa: E402

rrt = rrt_pkg.load()
from oracle import oracle  # noqa: E402

scene, w, h, spp, name = bench.build_scene(rrt, 4)
rc = rrt.Renderer(count_visits=True)
rc.upload(scene)
o = oracle.OracleScene(scene)
g = np.random.default_rng(1)
n = 1 << 21
xy = np.stack([g.uniform(0, w, n), g.uniform(0, h, n)], 1).astype(np.float32)
tile = (xy[:, 1].astype(np.int64) // 8) * ((w + 7) // 8) + xy[:, 0].astype(np.int64) // 8
xy = xy[np.argsort(tile, kind="stable")]
prim = o.camera_rays(w, h, xy)
dev = torch.device("cuda", 0)


def run(rays, tmax, label):
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).to(dev)
    d_hits = torch.empty(len(rays) * 16, dtype=torch.uint8, device=dev)
    rc.intersect_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), t_max=tmax)
    st = rc.render_stats()
    hits = d_hits.cpu().numpy().view(rrt.HIT_DTYPE)
    print(label, "nodes/ray %.2f prims/ray %.2f hit rate %.3f" % (st.node_visits / len(rays), st.prim_tests / len(rays), (hits["prim_id"] >= 0).mean()), flush=True)
    return hits


hits = run(prim, float("inf"), "unbounded      ")
ok = hits["prim_id"] >= 0
short = prim[ok].copy()
short["direction"] = prim["direction"][ok] * (hits["t"][ok, None] * np.float32(1.001))
run(prim[ok], float("inf"), "hits, unbounded")
run(short, 1.0, "hits, bounded  ")
