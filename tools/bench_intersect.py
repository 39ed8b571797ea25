#!/usr/bin/env python3
"""Throughput of the ray-batch closest-hit kernel (rrt_intersect_device) on primary and on incoherent bounce rays.
Experiment helper: tells how fast pure traversal runs outside the path-tracing megakernel."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import rrt_pkg  # noqa: E402

ITERS = int(os.environ.get("RRT_INTERSECT_ITERS", "5"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--rays", type=int, default=1 << 22)
    args = ap.parse_args()
    rrt = rrt_pkg.load()
    from oracle import oracle

    scene, w, h, spp, name = bench.build_scene(rrt, args.config)
    r = rrt.Renderer()
    r.upload(scene)
    rc = rrt.Renderer(count_visits=True)
    rc.upload(scene)
    o = oracle.OracleScene(scene)
    g = np.random.default_rng(1)
    n = args.rays
    xy = np.stack([g.uniform(0, w, n), g.uniform(0, h, n)], 1).astype(np.float32)
    # coherent order: sort primary rays by 8x8 tile like the render kernel sees them
    tile = (xy[:, 1].astype(np.int64) // 8) * ((w + 7) // 8) + xy[:, 0].astype(np.int64) // 8
    xy = xy[np.argsort(tile, kind="stable")]
    prim = o.camera_rays(w, h, xy)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)

    def run(rays, label):
        d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).to(dev)
        d_hits = torch.empty(len(rays) * 16, dtype=torch.uint8, device=dev)
        for _ in range(2):
            r.intersect_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(ITERS):
            r.intersect_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / ITERS
        hits = d_hits.cpu().numpy().view(rrt.HIT_DTYPE)
        rc.intersect_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), stream=stream.cuda_stream)
        st = rc.render_stats()
        n = len(rays)
        print(f"{label}: {n} rays {ms:.3f} ms {n / ms / 1e3:.0f} Mrays/s hit rate {(hits['prim_id'] >= 0).mean():.3f}"
              f" nodes/ray {st.node_visits / n:.1f} prims/ray {st.prim_tests / n:.2f}"
              f" lane util node {st.node_visits / max(64 * st.node_wave_steps, 1):.3f} prim {st.prim_tests / max(64 * st.prim_wave_steps, 1):.3f}"
              f" refills/ray {st.refill_wave_passes / n:.4f}", flush=True)
        return hits

    hits = run(prim, "primary (tile order)")
    ok = hits["prim_id"] >= 0
    p = prim["origin"][ok] + hits["t"][ok, None] * prim["direction"][ok]
    d = g.normal(size=(len(p), 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:, 1] = np.abs(d[:, 1])  # upwards: leave the terrain like a diffuse bounce
    bounce = np.zeros(len(p), rrt.RAY_DTYPE)
    bounce["origin"], bounce["direction"] = p + d * 1e-3, d
    run(bounce, "bounce (same order)")
    run(bounce[g.permutation(len(bounce))], "bounce (shuffled)")
    # the mix a path tracer's ray queue holds: primary and bounce rays, in runs of 64 and ray by ray
    m = min(len(prim), len(bounce)) // 64 * 64
    mixed = np.empty(2 * m, rrt.RAY_DTYPE)
    mixed.reshape(-1, 2, 64)[:, 0, :] = prim[:m].reshape(-1, 64)
    mixed.reshape(-1, 2, 64)[:, 1, :] = bounce[:m].reshape(-1, 64)
    run(mixed, "primary + bounce, runs of 64")
    mixed[0::2] = prim[:m]
    mixed[1::2] = bounce[:m]
    run(mixed, "primary + bounce, alternating")


if __name__ == "__main__":
    main()
