#!/usr/bin/env python3
"""Experiment helper for profiling the ray-batch kernel: a few launches on tile-ordered primary rays of one config."""
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

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
scene, w, h, _, name = bench.build_scene(rrt, cfg)
r = rrt.Renderer()
r.upload(scene)
o = oracle.OracleScene(scene)
g = np.random.default_rng(1)
xy = np.stack([g.uniform(0, w, n), g.uniform(0, h, n)], 1).astype(np.float32)
tile = (xy[:, 1].astype(np.int64) // 8) * ((w + 7) // 8) + xy[:, 0].astype(np.int64) // 8
xy = xy[np.argsort(tile, kind="stable")]
prim = o.camera_rays(w, h, xy)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
d_rays = torch.from_numpy(prim.view(np.uint8).reshape(-1)).to(dev)
d_hits = torch.empty(n * 16, dtype=torch.uint8, device=dev)
for _ in range(iters):
    r.intersect_device(d_rays.data_ptr(), n, d_hits.data_ptr(), stream=stream.cuda_stream)
torch.cuda.synchronize()
print("done", n, iters)
