#!/usr/bin/env python3
"""A mid-size unrolled scene (a 100k-triangle mesh x 16 instances, 4K frame): whole tiles against items of a few pixels
whose primary rays take per-pixel lists.   python tools/exp_mid_unrolled.py [grid]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import rrt_pkg
    import helpers

    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rrt = rrt_pkg.load()
    from rust_ray_tracer_amd import scenes  # (registered by rrt_pkg.load())
    scene = scenes.synthetic_instanced(rrt, helpers.ASSETS, grid=grid)
    r = rrt.Renderer(seed=1)
    info = r.upload(scene)
    w, h = 3840, 2160
    print(f"# grid {grid}: {info.node_count} nodes, unrolled {info.unrolled_instances}", flush=True)
    buf = helpers.DeviceArray((h, w, 3))
    for spp in (128, 1024):
        first = None
        for knobs in ({}, dict(split_items=1, pixel_lists=1, chunks=8), dict(split_items=1, pixel_lists=1, chunks=16), dict(split_items=1, pixel_lists=1, chunks=32)):
            r.set_tuning(**{**dict(split_items=None, pixel_lists=None, chunks=None), **knobs})
            ms = []
            for _ in range(3):
                r.render_device(w, h, spp, 0, buf.ptr, 0)
                st = r.render_stats()
                ms.append(st.kernel_ms)
            print(f"spp {spp} {knobs}: chunks {st.chunks} {min(ms[1:]):.1f} ms  {st.rays / min(ms[1:]) / 1e6:.2f} Grays/s listed {r.info().pixel_list_permille / 10:.1f} %", flush=True)
    r.close()


if __name__ == "__main__":
    main()
