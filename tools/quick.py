#!/usr/bin/env python3
"""Kernel-time sweeps in one process: python tools/quick.py --config 4 --spp 128,1024 --reps 3 --tune "leave_walk=8;leave_walk=16,leave_ready=8"
Every tuning set (rrt_tuning fields, ';' between sets, the empty set = the library's rules) is timed on every sample count;
prints the HIP-event time of the render launch (min and median over the repetitions) and the rate."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--spp", default="128")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--tune", default="")
    ap.add_argument("--count", action="store_true", help="counting variant of the kernel: lane utilisation per pass")
    ap.add_argument("--device-build", action="store_true")
    ap.add_argument("--check", action="store_true", help="compare every tuning set's image and ray count with the first set's (bit for bit)")
    args = ap.parse_args()
    import numpy as np
    import rrt_pkg
    import helpers
    import bench

    rrt = rrt_pkg.load()
    scene, w, h, _, workload = bench.build_scene(rrt, args.config)
    r = rrt.Renderer(seed=1, count_visits=args.count, device_build=args.device_build)
    info = r.upload(scene)
    print(f"# {workload}: {info.node_count} nodes, depth {info.bvh_depth}, build {info.build_ms:.0f} ms, unrolled {info.unrolled_instances}, scaled {info.scaled_slab}", flush=True)
    buf = helpers.DeviceArray((h, w, 3))
    sets = [t for t in args.tune.split(";")] if args.tune else [""]
    for spp in [int(x) for x in args.spp.split(",")]:
        first_img, first_rays = None, None
        for t in sets:
            knobs = {k: None for k in ("chunks", "split_items", "node_share_64ths", "leave_walk", "leave_ready", "overlap_items", "band_tiles", "scaled_slab", "blocks_per_cu", "pixel_lists", "pixel_list_len", "list_waves")}
            for kv in filter(None, t.split(",")):
                k, v = kv.split("=")
                knobs[k] = int(v)
            r.set_tuning(**knobs)
            ms = []
            for _ in range(args.reps + 1):
                r.render_device(w, h, spp, 0, buf.ptr, 0)
                st = r.render_stats()
                ms.append(st.kernel_ms)
            ms = ms[1:]
            extra = ""
            if args.count:
                extra = (f"  util n/p/s {st.node_visits / max(64 * st.node_wave_steps, 1):.3f}/{st.prim_tests / max(64 * st.prim_wave_steps, 1):.3f}/"
                         f"{st.rays / max(64 * st.shade_wave_passes, 1):.3f}  nodes/prims per ray {st.node_visits / st.rays:.2f}/{st.prim_tests / st.rays:.2f}"
                         f"  wave steps per 64 rays n/p/s {64 * st.node_wave_steps / st.rays:.1f}/{64 * st.prim_wave_steps / st.rays:.1f}/{64 * st.shade_wave_passes / st.rays:.2f}")
            if args.check:
                img = buf.numpy()
                if first_img is None:
                    first_img, first_rays = img, st.rays
                else:
                    same = np.array_equal(img, first_img)
                    extra += f"  image {'identical' if same else 'DIFFERS in %d floats' % int((img != first_img).sum())}, rays {'equal' if st.rays == first_rays else 'DIFFER %d vs %d' % (st.rays, first_rays)}"
            extra += f"  listed {r.info().pixel_list_permille / 10:.1f}%"
            print(f"cfg {args.config} spp {spp:5d} [{t or 'default':40s}] chunks {st.chunks:3d}  min {min(ms):9.2f} ms  med {float(np.median(ms)):9.2f} ms  {st.rays / min(ms) / 1e6:7.3f} Grays/s{extra}", flush=True)
    buf.free()
    r.close()


if __name__ == "__main__":
    main()
