#!/usr/bin/env python3
"""Experiment: the render kernel's lane utilisation and time per ray as a function of the path depth bound (depth 1 =
primary rays only, i.e. wavefronts whose 64 rays are all of one kind)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import rrt_pkg  # noqa: E402

rrt = rrt_pkg.load()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scene, w, h, _, name = bench.build_scene(rrt, cfg)
dev = torch.device("cuda", 0)
accum = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream(dev)
for depth in (1, 2, 3, 4):
    r = rrt.Renderer(max_depth=depth)
    r.upload(scene)
    for _ in range(2):
        r.render_device(w, h, spp, 0, accum.data_ptr(), stream.cuda_stream)
        st = r.render_stats()
    r.close()
    c = rrt.Renderer(max_depth=depth, count_visits=True)
    c.upload(scene)
    c.render_device(w, h, 8, 0, accum.data_ptr(), stream.cuda_stream)
    cs = c.render_stats()
    c.close()
    print(f"depth {depth}: {st.kernel_ms:.1f} ms {st.rays / st.kernel_ms / 1e6:.2f} Grays/s rays/sample {st.rays / (w * h * spp):.2f}"
          f" nodes/ray {cs.node_visits / cs.rays:.1f} prims/ray {cs.prim_tests / cs.rays:.2f}"
          f" util node {cs.node_visits / max(64 * cs.node_wave_steps, 1):.3f} prim {cs.prim_tests / max(64 * cs.prim_wave_steps, 1):.3f}"
          f" shade {cs.rays / max(64 * cs.shade_wave_passes, 1):.3f}"
          f" idle done {cs.node_slots_lane_done / max(64 * cs.node_wave_steps, 1):.3f} at leaf {cs.node_slots_lane_at_leaf / max(64 * cs.node_wave_steps, 1):.3f}",
          flush=True)
