"""Time of a 128-spp launch of the headline frame by maximum path depth, with and without pixel lists:
    python tools/depth_split.py   (on the GPU box)"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, rrt_pkg, helpers, bench
rrt = rrt_pkg.load()
scene, w, h, _, workload = bench.build_scene(rrt, 4)
buf = helpers.DeviceArray((h, w, 3))
for depth in (1, 2, 3, 4):
    for lists in (1, 0):
        r = rrt.Renderer(seed=1, max_depth=depth)
        r.upload(scene)
        r.set_tuning(pixel_lists=lists)
        ms = []
        for _ in range(3):
            r.render_device(w, h, 128, 0, buf.ptr, 0)
            st = r.render_stats(); ms.append(st.kernel_ms)
        print(f"max_depth {depth} lists {lists}: {min(ms):8.2f} ms  rays {st.rays/1e9:.3f} G  chunks {st.chunks}", flush=True)
        r.close()
