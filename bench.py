#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 4|1|2|3|5|6] [--spp S] [--scaling strong|weak]

The default workload is the frame the headline target is quoted on: the synthetic 1,048,576-triangle mesh at 3840x2160
with 1024 samples per pixel.  Strong scaling (default): the frame is the same for every N, its sample range is split
over the ranks (N = 1 renders all 1024 samples, N = 8 renders 128 per GPU).  `--scaling weak`: every rank renders
`--spp` samples and the frame gets N times as many.  `--config 2` is the single-GPU asset case (suzanne, 1080p, 256 spp).

One *step* = one full pass of the hot path over one frame: render kernel (camera rays -> closest hit -> scatter ->
bounce), the RCCL reduce of the radiance accumulation buffers to rank 0 when N > 1, and the tone map on rank 0.  Scene,
BVH, camera and materials are resident in HBM before the timed region.  Multi-GPU: one process per GPU
(torch.distributed, backend nccl = RCCL).  Started as plain `python bench.py --gpus N` with N > 1, the script launches
its N ranks itself (torch.distributed.run) before anything touches a GPU, and exits with their status.

Prints ONE JSON line (rank 0): metric Mrays/s over the whole job, plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
NODE_BYTES, PRIM_BYTES = 32, 48
# Issue roofline: a SIMD issues one wave64 vector instruction in two cycles at best (four SIMDs per CU, 256 CUs).
GPU_GHZ = 2.4
SIMDS = 1024
VALU_CYCLES_PER_WAVE_INST = 2.0
ISSUE_PEAK_GINST = SIMDS * GPU_GHZ / VALU_CYCLES_PER_WAVE_INST  # 1228.8 G wave-instructions per second
# Vector L1 address path: a CU's one address unit takes 16 cycles per wavefront load instruction of 8 to 16 bytes per
# lane, however many lanes are switched on (measured: tools/l1_rate.hip), and one more per 128-byte line beyond 16.
CUS = 256
L1_CYCLES_PER_WAVE_LOAD = 16.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel: the frame's total under strong scaling, per GPU under weak scaling (default: the config's)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame-check", action="store_true", help="skip the untimed comparison of the frame with its whole-tile rendering")
    ap.add_argument("--device-build", action="store_true", help="build large BVHs on the GPU (linear BVH) instead of the host's SAH build")
    return ap.parse_args()


def launch_ranks(n: int) -> int:
    """Starts this script as n ranks under torch.distributed.run (a child process: the parent has not touched a GPU and
    does not replace itself) and returns the launcher's exit status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def build_scene(rrt, config: int):
    """Returns (scene, width, height, samples per pixel of the frame, workload name)."""
    from rust_ray_tracer_amd import scenes  # noqa: F401  (registered by rrt_pkg.load())

    assets = os.path.join(ROOT, "assets")
    if config == 1:
        s = rrt.Scene.from_gltf_file(os.path.join(assets, "default.gltf"), os.path.join(assets, "suzanne.gltf"))
        h = 400
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 64, "assets/default.gltf (12 triangles), 711x400"
    if config == 2:
        s = rrt.Scene.from_gltf_file(os.path.join(assets, "suzanne.gltf"), os.path.join(assets, "suzanne.gltf"))
        h = 1080
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 256, "assets/suzanne.gltf (968 triangles), 1920x1080"
    if config == 3:
        s = scenes.metal_rough_spheres(rrt, assets, dielectric=True)
        h = 1440
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 1024, "MetalRoughSpheres (1.04M triangles, extended loader, every 7th sphere glass), 2560x1440"
    if config == 4:
        s = scenes.synthetic_mesh(rrt, assets, subdiv=512)
        h = 2160
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 1024, "synthetic 1,048,576-triangle displaced mesh, 3840x2160"
    if config == 5:
        s = scenes.synthetic_instanced(rrt, assets)
        h = 4320
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 4096, "synthetic 100k-triangle mesh x 100 instances, 7680x4320"
    if config == 6:
        s = scenes.synthetic_flattened(rrt, assets)
        h = 4320
        return s, rrt.image_width(h, s.camera.aspect_ratio), h, 4096, "the instanced scene flattened: 10,002,102 unique triangles, 7680x4320"
    raise SystemExit(f"unknown config {config}")


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, width, height, spp_full, seed, mode_name, budget_s=30.0):
    """Times the CPU restatement (reference-style pointer BVH, recursive ray_color, row-parallel threads) on a bounded
    sample of the same workload: the same camera frame at a lower resolution and sample count (the same mix of rays).
    mode_name "reference": the reference's per-axis box test (src/bvh/aabb.rs:14-36); "running_slab": the same tree
    with a box test that carries the interval across the axes, i.e. a hierarchy that culls."""
    from oracle import oracle

    o = oracle.OracleScene(scene)
    threads = oracle.default_threads()
    try:
        o.build_ref_bvh()
        mode = oracle.HIT_REF_BVH if mode_name == "reference" else oracle.HIT_REF_BVH_RUNNING_SLAB
        tree = "reference-style pointer BVH (median split, one primitive per leaf)"
    except ValueError:
        mode = oracle.HIT_BRUTE  # instanced scenes: the oracle has no tree for them
        tree = "exhaustive search (the oracle builds no tree for instanced scenes)"
    # sized from a tiny calibration frame: with the reference's box test one ray of the 1M-triangle scene costs the CPU
    # tens of milliseconds
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
    box = ("the reference's per-axis box test" if mode_name == "reference" else "running-interval slab test (NOT the reference's)") if mode != oracle.HIT_BRUTE else "no boxes"
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
            "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"), "kind": "port",
            "sample": f"the frame at {sw}x{sh} (full: {width}x{height}), {spp} of {spp_full} spp: {rays} rays in {dt:.1f} s; {tree}, {box}",
            "samples_per_s": sw * sh * spp / dt, "rays": rays, "seconds": dt}


def measured_pmc(config: int, width: int, height: int, spp: int):
    """Hardware-counter figures of k_render from the committed rocprofv3 PMC passes (profiles/pmc_per_launch.json, written
    by tools/pmc_to_json.py from tools/prof.sh output), or None when this scene and frame size have not been profiled.
    The figures are per launch of the profiled sample count and are used per ray: the record profiled at this launch's
    sample count ("<config>@<spp>spp") if there is one; for the ten-million-triangle scenes, whose launches of 128
    samples and more run another form of the kernel (primary rays from per-pixel lists), the record of such a launch;
    else the configuration's default record."""
    path = os.path.join(ROOT, "profiles", "pmc_per_launch.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    keys = [f"{config}@{spp}spp"]
    if config in (5, 6) and spp >= 128:
        keys.append(f"{config}@256spp")
    keys.append(str(config))
    for key in keys:
        row = table.get(key)
        if row and (row["width"], row["height"]) == (width, height) and row.get("rays_per_launch"):
            return dict(row, key=key)
    return None


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    import rrt_pkg

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal on a box with fewer GPUs than ranks (RRT_BENCH_REHEARSAL=1): ranks share the GPUs there are and the
    # reduce goes through host memory with gloo, because RCCL refuses two ranks on one GPU.  Not a measurement.
    rehearsal = os.environ.get("RRT_BENCH_REHEARSAL") == "1"
    gpu_index = local_rank % max(torch.cuda.device_count(), 1) if rehearsal else local_rank
    torch.cuda.set_device(gpu_index)
    device = torch.device("cuda", gpu_index)
    # RRT_BENCH_FORCE_GROUP=1 (under a launcher, one rank): the process group is set up although there is nobody to
    # talk to, so that the RCCL calls of the multi-GPU path run on a box with one GPU
    grouped = world > 1 or (os.environ.get("RRT_BENCH_FORCE_GROUP") == "1" and "WORLD_SIZE" in os.environ)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    rrt = rrt_pkg.load()
    from rust_ray_tracer_amd import dist as rdist

    scene, width, height, spp_config, workload = build_scene(rrt, args.config)
    if args.scaling == "strong":
        spp_arg = args.spp if args.spp > 0 else spp_config
    else:
        spp_arg = args.spp if args.spp > 0 else max(1, spp_config // 8)
    spp_offset, spp, spp_total = rdist.rank_slice(rank, world, spp_arg, args.scaling)
    if spp == 0:
        raise SystemExit(f"rank {rank}: {spp_arg} samples leave nothing for {world} ranks")
    if args.spp > 0 and spp_total != spp_config and rank == 0:
        print(f"bench.py: NOTE: the frame gets {spp_total} samples per pixel, not the {spp_config} the configuration is quoted on "
              "(--spp was given); the rate per ray does not depend on the sample count", file=sys.stderr)

    renderer = rrt.Renderer(device=gpu_index, seed=args.seed, device_build=args.device_build)
    info = renderer.upload(scene)
    accum = torch.empty((height, width, 3), dtype=torch.float32, device=device)
    rgb8 = torch.empty((height, width, 3), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream(device)

    kernel_ms = []
    rays_per_step = 0

    last_note = [time.perf_counter()]

    def step(record: bool):
        nonlocal rays_per_step
        renderer.render_device(width, height, spp, spp_offset, accum.data_ptr(), stream.cuda_stream)
        st = renderer.render_stats()  # waits for the kernel; its HIP events are on `stream`
        if rank == 0 and time.perf_counter() - last_note[0] > 30.0:  # (long frames: a sign of life on stderr)
            last_note[0] = time.perf_counter()
            print(f"bench.py: a step of {st.kernel_ms:.0f} ms done", file=sys.stderr, flush=True)
        rdist.reduce_accum(accum, dst=0, via_host=rehearsal)
        if rank == 0:
            renderer.tonemap_device(accum.data_ptr(), width, height, spp_total, rgb8.data_ptr(), stream.cuda_stream)
        if record:
            kernel_ms.append(st.kernel_ms)
            rays_per_step = st.rays

    for _ in range(args.warmup):
        step(False)
    # The library times the first launches of a frame geometry to choose between two builds of its render kernel (six or
    # seven wavefronts per SIMD; rrt_tuning.list_waves): three launches until the choice stands, whatever --warmup says.
    for _ in range(max(0, 3 - args.warmup)):
        step(False)
    # One more untimed step on rank 0: the frame as the timed steps render it (where the library cuts it that way: work items of
    # a few pixels x many sample chunks, primary rays from per-pixel lists) against the same sample slice rendered with whole 8x8 tiles as items.  Every
    # pixel's chunk sums are formed by one lane in sample order either way, so the two must agree in every bit and in
    # the number of rays; a difference ends the benchmark.
    frame_check = None
    if rank == 0 and not args.no_frame_check:
        renderer.render_device(width, height, spp, spp_offset, accum.data_ptr(), stream.cuda_stream)
        st_a = renderer.render_stats()
        listed_permille = renderer.info().pixel_list_permille
        whole = torch.empty_like(accum)
        renderer.set_tuning(chunks=st_a.chunks, split_items=0)
        renderer.render_device(width, height, spp, spp_offset, whole.data_ptr(), stream.cuda_stream)
        st_b = renderer.render_stats()
        renderer.set_tuning(chunks=None, split_items=None)
        same = bool(torch.equal(accum, whole)) and st_a.rays == st_b.rays and bool(torch.isfinite(accum).all())
        del whole
        if not same:
            raise SystemExit("bench.py: the timed kernel variant and the whole-tile form of the kernel disagree on this frame")
        listed = listed_permille / 10.0
        frame_check = (f"bit-identical to whole-tile items whose primary rays walk the tree ({st_a.chunks} sample chunks, {st_a.rays} rays; "
                       f"timed variant: {listed:.1f} % of the pixels resolve their primary rays against candidate lists)")
    if grouped:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize(device)
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    elapsed = rdist.all_reduce_value(elapsed, "max", device, via_host=rehearsal)
    rays_all = rdist.all_reduce_value(rays_per_step, "sum", device, via_host=rehearsal)  # rays of one step, all ranks
    samples_all = rdist.all_reduce_value(width * height * spp, "sum", device, via_host=rehearsal)

    # --- roofline of the dominant kernel (k_render), rank 0's launch
    roof = None
    cpu = None
    cpu_sound = None
    if rank == 0:
        counting = rrt.Renderer(device=gpu_index, seed=args.seed, count_visits=True, device_build=args.device_build)
        counting.upload(scene)
        # samples of the counting launch: enough for the launch to be cut into items the way the timed one is (one GPU's
        # share of the headline frame under eight-way scaling); a launch of a few samples is one chunk and mostly tail
        probe_spp = min(spp, int(os.environ.get("RRT_PROBE_SPP", "128")))
        counting.render_device(width, height, probe_spp, spp_offset, accum.data_ptr(), stream.cuda_stream)
        cs = counting.render_stats()
        counting.close()
        nodes_per_ray = cs.node_visits / max(cs.rays, 1)
        prims_per_ray = cs.prim_tests / max(cs.rays, 1)
        samples = width * height * spp
        # algorithmic bytes: the node and primitive records the rays of one launch touch, plus the accumulation write
        bytes_per_ray = nodes_per_ray * NODE_BYTES + prims_per_ray * PRIM_BYTES
        algo_bytes = rays_per_step * bytes_per_ray + width * height * 12.0
        k_ms = float(np.mean(kernel_ms))
        k_s = k_ms * 1e-3
        pmc = measured_pmc(args.config, width, height, spp)
        roof = {"kernel": "k_render", "kernel_ms": k_ms, "rays_per_launch": rays_per_step, "samples_per_launch": samples,
                "nodes_per_ray": nodes_per_ray, "prims_per_ray": prims_per_ray, "bytes_per_ray": bytes_per_ray,
                # what the rays read and write by the algorithm's own count, served mostly by L1 / L2 / Infinity Cache:
                # a rate, NOT a fraction of the HBM peak
                "algorithmic_GBps": algo_bytes / k_s / 1e9,
                # fraction of the 64 lanes doing useful work in a node step / primitive test / shading pass (software counters)
                # share of the frame's pixels whose primary rays take a candidate list instead of a walk (k_pixel_lists)
                "pixel_list_share": renderer.info().pixel_list_permille / 1000.0,
                "lane_util_node": cs.node_visits / max(64 * cs.node_wave_steps, 1),
                "lane_util_prim": cs.prim_tests / max(64 * cs.prim_wave_steps, 1),
                "lane_util_shade": cs.rays / max(64 * cs.shade_wave_passes, 1),
                # where the idle lanes of the node passes are: walk already complete / parked at a leaf since the pass began
                "node_slots_idle_done": cs.node_slots_lane_done / max(64 * cs.node_wave_steps, 1),
                "node_slots_idle_at_leaf": cs.node_slots_lane_at_leaf / max(64 * cs.node_wave_steps, 1),
                "wave_steps_per_ray": {"node": 64 * cs.node_wave_steps / max(cs.rays, 1), "prim": 64 * cs.prim_wave_steps / max(cs.rays, 1),
                                       "shade": 64 * cs.shade_wave_passes / max(cs.rays, 1)}}
        if pmc:
            per_ray = 1.0 / pmc["rays_per_launch"]
            valu = pmc["sq_insts_valu"] * per_ray * rays_per_step      # vector wave-instructions of this launch
            traffic = pmc["hbm_bytes"] * per_ray * rays_per_step        # HBM-side bytes of this launch (FETCH_SIZE / WRITE_SIZE)
            insts = sum(pmc.get(k, 0.0) for k in ("sq_insts_valu", "sq_insts_salu", "sq_insts_lds", "sq_insts_vmem_rd")) * per_ray
            hbm_gbps = traffic / k_s / 1e9
            hbm_frac = hbm_gbps / HBM_PEAK_GBS
            valu_rate = valu / k_s / 1e9
            roof.update({
                "traffic": traffic, "hbm_GBps": hbm_gbps, "hbm_frac": hbm_frac,
                "valu_Ginst_per_s": valu_rate, "issue_frac": valu_rate / ISSUE_PEAK_GINST,
                "lane_util_hw": (pmc["sq_thread_cycles_valu"] / (64.0 * pmc["sq_active_inst_valu"])) if pmc.get("sq_active_inst_valu") else None,
                "insts_per_ray": insts, "valu_per_ray": pmc["sq_insts_valu"] * per_ray, "salu_per_ray": pmc.get("sq_insts_salu", 0.0) * per_ray,
                "simd_cycles_per_inst": k_s * GPU_GHZ * 1e9 * SIMDS / (insts * rays_per_step),
                # the other resource the kernel nearly fills: the address unit of each CU's vector L1.  floor: wavefront
                # load instructions x 16 cycles / (CUs x kernel cycles); busy: the unit's own busy counter (TA_BUSY)
                "l1_loads_per_ray": pmc.get("sq_insts_vmem_rd", 0.0) * per_ray,
                "l1_addr_floor_frac": pmc.get("sq_insts_vmem_rd", 0.0) * per_ray * rays_per_step * L1_CYCLES_PER_WAVE_LOAD / (CUS * k_s * GPU_GHZ * 1e9),
                "l1_addr_busy_frac": pmc.get("ta_busy_frac"),
                "waves_per_simd": pmc.get("waves_per_simd"),
                "pmc_source": f"profiles/pmc_per_launch.json [{pmc['key']}] ({pmc.get('source', 'rocprofv3 --pmc, tools/prof.sh')}), profiled at {pmc.get('spp_per_gpu')} spp, used per ray"})
            # the unit the kernel fills most is the one that bounds it: the SIMDs' issue slots (vector instructions at two
            # cycles each), the address unit of the vector L1 (16 cycles per wavefront load instruction), or HBM
            l1_rate = pmc.get("sq_insts_vmem_rd", 0.0) * per_ray * rays_per_step / k_s / 1e9  # G wavefront loads/s
            l1_peak = CUS * GPU_GHZ / L1_CYCLES_PER_WAVE_LOAD
            candidates = {
                "issue": (valu_rate / ISSUE_PEAK_GINST, valu_rate, ISSUE_PEAK_GINST, "G wave-instructions/s (VALU)"),
                "l1_addr": (l1_rate / l1_peak, l1_rate, l1_peak, "G wavefront load instructions/s"),
                "hbm": (hbm_frac, hbm_gbps, HBM_PEAK_GBS, "GB/s"),
            }
            bound = max(candidates, key=lambda k: candidates[k][0])
            frac, achieved, peak, unit = candidates[bound]
            roof.update({"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": frac})
            roof["note"] = ("bound = the fullest of three units.  issue: vector wave-instructions x 2 cycles / (1024 SIMDs x kernel cycles at "
                            "2.4 GHz), the share of the SIMDs' issue slots the kernel fills at the two-cycle rate (half of the kernel's vector "
                            "instructions are of the half-rate class -- min/max, compares, selects, rotates -- which cannot reach it: "
                            "profiles/round2_valu_instruction_classes_mi355x.txt).  l1_addr: wavefront load instructions x 16 cycles / (256 CUs x "
                            "kernel cycles), what the loads alone take of each CU's one L1 address unit (16 cycles per load instruction whatever "
                            "the number of active lanes, tools/l1_rate.hip); l1_addr_busy_frac = TA_BUSY_avr / kernel cycles is that unit's own "
                            "busy counter.  hbm: HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE) / kernel time over 8 TB/s.  lane_util_hw = "
                            "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) is the share of lanes that work in an issued vector "
                            "instruction.  algorithmic_GBps is served by the caches and is not a fraction of any peak.  kernel_ms is taken "
                            "between HIP events around the launch: k_render plus the two small kernels of a launch (k_pixel_lists before it, "
                            "k_reduce_chunks after it: 0.5 % of the time on the headline frame, profiles/round3_config4_1024spp_rocprofv3_summary.txt).")
        else:
            roof.update({"bound": "issue", "achieved": None, "peak": ISSUE_PEAK_GINST, "unit": "G wave-instructions/s (VALU)", "frac": None,
                         "traffic": None, "note": "this scene / frame size has no committed PMC profile (profiles/pmc_per_launch.json)"})
        if world == 1 and not args.no_cpu_baseline:
            # the reference's box test hardly culls, so its sample is bounded by time (45 s); the sound hierarchy's
            # sample is millions of rays in 15 s
            cpu = cpu_baseline(scene, width, height, spp_total, args.seed, "reference", budget_s=45.0)
            cpu_sound = cpu_baseline(scene, width, height, spp_total, args.seed, "running_slab", budget_s=15.0)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_all * args.steps / elapsed / 1e6
        data = "synthetic" if args.config >= 4 else "reference asset (bundled glTF); counter-RNG seed %d" % args.seed
        out = {
            "metric": "Mrays/s (whole node)", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32",
            "data": ("REHEARSAL, ranks share GPUs, not a measurement; " if rehearsal else "") + data,
            "process_group": (dist.get_backend() if grouped else None),
            "config": {"workload": f"{workload}, {spp_total} spp" + (f" ({spp} per GPU)" if world > 1 else ""),
                       "width": width, "height": height, "spp_per_gpu": spp, "spp_total": spp_total,
                       "max_depth": 4, "prims": info.prim_count, "bvh_nodes": info.node_count, "bvh_depth": info.bvh_depth,
                       "scene_bytes": info.device_bytes, "bvh_build_ms": info.build_ms,
                       "bvh_builder": "gpu linear BVH" if info.device_built else "host binned SAH",
                       "partition": "spp slices per GPU + RCCL reduce(sum) to rank 0"},
            "frame_check": frame_check,
            "samples_per_s": samples_all * args.steps / elapsed,
            "rays_per_sample": rays_all / samples_all,
            "roofline": roof,
            "cpu_baseline": cpu,
            "cpu_baseline_sound_bvh": cpu_sound,
        }
        print(json.dumps(out), flush=True)
    renderer.close()
    if grouped:
        dist.barrier()  # rank 0 is still measuring its roofline figures: nobody tears the group down before it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
