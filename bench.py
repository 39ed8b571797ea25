#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 4|1|2|3|5|6] [--spp S]

The default workload is the one the headline target is quoted on: the synthetic 1,048,576-triangle mesh at
3840x2160 with 128 samples per pixel per GPU, so that 8 GPUs render the 4K x 1024 spp frame of that target and one GPU
renders its eighth of the samples.  `--config 2` is the single-GPU asset case (suzanne, 1080p, 256 spp).

One *step* = one full pass of the hot path over one frame: render kernel (camera rays -> closest hit ->
scatter -> bounce, `spp` samples per pixel), the RCCL reduce of the radiance accumulation buffers to rank 0
when N > 1, and the tone map on rank 0.  Scene, BVH, camera and materials are resident in HBM before the
timed region.  Multi-GPU: one process per GPU; rank g renders sample indices [g*spp, (g+1)*spp) of every
pixel (weak scaling: per-GPU work is fixed, the image gets N*spp samples).

Prints ONE JSON line (rank 0): metric Mrays/s over the whole job, plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rrt_pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
NODE_BYTES, PRIM_BYTES, ACCUM_BYTES_PER_SAMPLE_AT_SPP1 = 32, 48, 12


def build_scene(rrt, config: int):
    """Returns (scene, width, height, default spp per GPU, workload name)."""
    from rust_ray_tracer_amd import scenes  # noqa: F401  (registered by rrt_pkg.load())

    assets = os.path.join(ROOT, "assets")
    if config == 1:
        s = rrt.Scene.from_gltf_file(os.path.join(assets, "default.gltf"), os.path.join(assets, "suzanne.gltf"))
        h = 400
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 64, "assets/default.gltf, 711x400, 64 spp"
    if config == 2:
        s = rrt.Scene.from_gltf_file(os.path.join(assets, "suzanne.gltf"), os.path.join(assets, "suzanne.gltf"))
        h = 1080
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 256, "assets/suzanne.gltf (968 tris), 1920x1080, 256 spp per GPU"
    if config == 3:
        s = scenes.metal_rough_spheres(rrt, assets)
        h = 1440
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 128, "MetalRoughSpheres (1.04M tris, extended loader), 2560x1440, 128 spp per GPU"
    if config == 4:
        s = scenes.synthetic_mesh(rrt, assets, subdiv=512)
        h = 2160
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 128, "synthetic 1,048,576-triangle displaced mesh, 3840x2160, 128 spp per GPU"
    if config == 5:
        s = scenes.synthetic_instanced(rrt, assets)
        h = 4320
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 16, "synthetic 100k-tri mesh x 100 instances, 7680x4320, 16 spp per GPU"
    if config == 6:
        s = scenes.synthetic_flattened(rrt, assets)
        h = 4320
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 16, "the instanced scene flattened: 10,002,102 unique triangles, 7680x4320, 16 spp per GPU"
    raise SystemExit(f"unknown config {config}")


def cpu_baseline(scene, width, height, spp_full, seed, budget_s=15.0):
    """Times the CPU restatement (reference-style pointer BVH, recursive ray_color, row-parallel threads) on a
    bounded sample of the same workload: the full frame at a reduced number of samples per pixel."""
    from oracle import oracle

    o = oracle.OracleScene(scene)
    threads = oracle.default_threads()
    try:
        o.build_ref_bvh()
        mode = oracle.HIT_REF_BVH
    except ValueError:
        mode = oracle.HIT_BRUTE
    # The sample is the same camera frame at a lower resolution and sample count (the same mix of rays).  It is sized
    # from a tiny calibration frame: the reference's box test (src/bvh/aabb.rs:14-36) prunes so little that on the
    # 1M-triangle scene one ray costs the CPU tens of milliseconds.
    cw, ch = max(16, width // 80), max(9, height // 80)
    t0 = time.perf_counter()
    _, rays = o.render(cw, ch, 1, mode=mode, threads=threads, seed=seed)
    dt = max(time.perf_counter() - t0, 1e-4)
    per_sample = dt / (cw * ch)
    want = budget_s / per_sample  # samples that fit the budget
    scale = min(1.0, (want / (width * height)) ** 0.5)
    sw, sh = max(cw, int(width * scale)), max(ch, int(height * scale))
    spp = int(max(1, min(spp_full, want / (sw * sh))))
    if (sw, sh, spp) != (cw, ch, 1):
        t0 = time.perf_counter()
        _, rays = o.render(sw, sh, spp, mode=mode, threads=threads, seed=seed)
        dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"the frame at {sw}x{sh} (full: {width}x{height}), {spp} of {spp_full} spp: {rays} rays in {dt:.1f} s, "
                      "reference-style pointer BVH and box test",
            "samples_per_s": sw * sh * spp / dt}


def measured_pmc(config: int, width: int, height: int, spp: int):
    """Per-launch hardware-counter figures of k_render from the committed rocprofv3 PMC passes
    (profiles/pmc_per_launch.json, written from tools/prof.sh output); None when this exact workload has not been
    profiled."""
    path = os.path.join(ROOT, "profiles", "pmc_per_launch.json")
    try:
        with open(path) as f:
            row = json.load(f).get(str(config))
    except (OSError, ValueError):
        return None
    if not row or (row["width"], row["height"], row["spp_per_gpu"]) != (width, height, spp):
        return None
    return row


# Issue cost in cycles of one wave-instruction on one SIMD, by instruction class, measured with tools/valu_peak.hip
# (profiles/round1_valu_issue_rate_mi355x.txt): plain add/mul/logic/move, v_fma_f32, and the class of min/max/min3/max3,
# sub-dword converts, bit-field and select forms.  Scalar and branch instructions are not free riders: a SIMD issues about
# one instruction of ANY kind per two cycles (tools/salu_mix.hip, profiles/round1_issue_rate_vector_scalar_mix_mi355x.txt:
# the kernel's own mix of vector and scalar instructions, no memory operations, 2.02 cycles each at seven wavefronts per
# SIMD), so the kernel is rated by all the instructions it issues.
GPU_GHZ = 2.4
VALU_CLASS_CYCLES = {"add_mul_logic": 2.5, "fma": 3.06, "minmax_convert_select": 4.3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per GPU (default: the config's)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-build", action="store_true", help="build large BVHs on the GPU (linear BVH) instead of the host's SAH build")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # Rehearsal on a box with fewer GPUs than ranks (RRT_BENCH_REHEARSAL=1): ranks share the GPUs there are and the
    # reduce goes through host memory with gloo, because RCCL refuses two ranks on one GPU.  Not a measurement.
    rehearsal = os.environ.get("RRT_BENCH_REHEARSAL") == "1"
    gpu_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(gpu_index)
    device = torch.device("cuda", gpu_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def reduce_to_root(t):
        if rehearsal:
            h = t.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                t.copy_(h)
        else:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)

    def all_reduce(t, op):
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    rrt = rrt_pkg.load()
    scene, width, height, spp, workload = build_scene(rrt, args.config)
    if args.spp > 0:
        spp = args.spp
    spp_total = spp * world
    spp_offset = rank * spp

    renderer = rrt.Renderer(device=gpu_index, seed=args.seed, device_build=args.device_build)
    info = renderer.upload(scene)
    accum = torch.empty((height, width, 3), dtype=torch.float32, device=device)
    rgb8 = torch.empty((height, width, 3), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream(device)

    kernel_ms = []
    rays_per_step = 0

    def step(record: bool):
        nonlocal rays_per_step
        renderer.render_device(width, height, spp, spp_offset, accum.data_ptr(), stream.cuda_stream)
        st = renderer.render_stats()  # waits for the kernel; its HIP events are on `stream`
        if world > 1:
            reduce_to_root(accum)
        if rank == 0:
            renderer.tonemap_device(accum.data_ptr(), width, height, spp_total, rgb8.data_ptr(), stream.cuda_stream)
        if record:
            kernel_ms.append(st.kernel_ms)
            rays_per_step = st.rays

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    r = torch.tensor([float(rays_per_step)], dtype=torch.float64, device=device)
    if world > 1:
        all_reduce(t, dist.ReduceOp.MAX)
        all_reduce(r, dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays_all = float(r.item())  # rays of one step, all ranks

    # --- roofline of the dominant kernel (k_render), rank 0's launch
    roof = None
    cpu = None
    if rank == 0:
        counting = rrt.Renderer(device=gpu_index, seed=args.seed, count_visits=True, device_build=args.device_build)
        counting.upload(scene)
        probe_spp = min(spp, int(os.environ.get("RRT_PROBE_SPP", "8")))  # samples of the counting launch
        counting.render_device(width, height, probe_spp, spp_offset, accum.data_ptr(), stream.cuda_stream)
        cs = counting.render_stats()
        counting.close()
        nodes_per_ray = cs.node_visits / max(cs.rays, 1)
        prims_per_ray = cs.prim_tests / max(cs.rays, 1)
        samples = width * height * spp
        bytes_per_ray = nodes_per_ray * NODE_BYTES + prims_per_ray * PRIM_BYTES
        algo_bytes = rays_per_step * bytes_per_ray + width * height * 12.0
        k_ms = float(np.mean(kernel_ms))
        pmc = measured_pmc(args.config, width, height, spp)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": float(pmc["hbm_bytes"]) if pmc else None, "kernel": "k_render", "kernel_ms": k_ms, "nodes_per_ray": nodes_per_ray,
                "prims_per_ray": prims_per_ray, "bytes_per_ray": bytes_per_ray, "rays_per_launch": rays_per_step,
                "samples_per_launch": samples,
                # fraction of the 64 lanes doing useful work in a node step / primitive test / shading pass
                "lane_util_node": cs.node_visits / max(64 * cs.node_wave_steps, 1),
                "lane_util_prim": cs.prim_tests / max(64 * cs.prim_wave_steps, 1),
                "lane_util_shade": cs.rays / max(64 * cs.shade_wave_passes, 1),
                # where the idle lanes of the node passes are: walk already complete / parked at a leaf since the pass began
                "node_slots_idle_done": cs.node_slots_lane_done / max(64 * cs.node_wave_steps, 1),
                "node_slots_idle_at_leaf": cs.node_slots_lane_at_leaf / max(64 * cs.node_wave_steps, 1),
                "wave_steps_per_ray": {"node": 64 * cs.node_wave_steps / max(cs.rays, 1), "prim": 64 * cs.prim_wave_steps / max(cs.rays, 1),
                                       "shade": 64 * cs.shade_wave_passes / max(cs.rays, 1)}}
        roof["note"] = ("achieved = algorithmic bytes (nodes and primitive records the rays touch) / kernel time; the caches serve "
                        "most of them, the HBM-side bytes per launch are `traffic`; the kernel is bound by instruction issue "
                        "(vector and scalar alike) and SIMD divergence (`issue`, `lane_util_*`)")
        if pmc and "sq_insts_valu" in pmc:
            # the kernel is bound by instruction issue, not by HBM: SIMD cycles of the launch per instruction issued
            # (vector, scalar, LDS and vector-memory), against the same figure of a memory-free stream of the same mix
            insts = sum(pmc.get(k, 0.0) for k in ("sq_insts_valu", "sq_insts_salu", "sq_insts_lds", "sq_insts_vmem_rd"))
            cycles = k_ms * 1e-3 * GPU_GHZ * 1e9
            per_inst = cycles * pmc["simds"] / insts
            roof["issue"] = {"insts_per_ray": insts / rays_per_step,
                             "valu_per_ray": pmc["sq_insts_valu"] / rays_per_step,
                             "salu_per_ray": pmc.get("sq_insts_salu", 0.0) / rays_per_step,
                             "simd_cycles_per_inst": per_inst,
                             "peak_simd_cycles_per_inst": pmc.get("issue_peak_cycles_per_inst"),
                             "frac_of_issue_peak": (pmc["issue_peak_cycles_per_inst"] / per_inst) if pmc.get("issue_peak_cycles_per_inst") else None,
                             "valu_class_cycles": VALU_CLASS_CYCLES,
                             "source": "profiles/pmc_per_launch.json (rocprofv3 --pmc SQ_INSTS_*), tools/salu_mix.hip, tools/valu_peak.hip"}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(scene, width, height, spp, args.seed)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_all * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/s (whole node)", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": ("REHEARSAL, ranks share GPUs, not a measurement; " if rehearsal else "") + "synthetic" if args.config >= 4 else "reference asset (bundled glTF), random-free scene; counter-RNG seed %d" % args.seed,
            "config": {"workload": workload, "width": width, "height": height, "spp_per_gpu": spp, "spp_total": spp_total,
                       "max_depth": 4, "prims": info.prim_count, "bvh_nodes": info.node_count, "bvh_depth": info.bvh_depth,
                       "scene_bytes": info.device_bytes, "bvh_build_ms": info.build_ms,
                       "bvh_builder": "gpu linear BVH" if info.device_built else "host binned SAH", "partition": "spp slices per GPU + RCCL reduce(sum) to rank 0"},
            "samples_per_s": width * height * spp * world * args.steps / elapsed,
            "rays_per_sample": rays_all / (width * height * spp * world),
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    renderer.close()
    if world > 1:
        dist.barrier()  # rank 0 is still measuring its roofline figures: nobody tears the group down before it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
