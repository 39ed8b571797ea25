"""The multi-process path on CPU: world_size 2 over gloo.  Each rank renders its slice of the sample range with the
ORACLE (standing in for the GPU kernel, which cannot run here); the slice arithmetic (`dist.rank_slice`) and the sum on
rank 0 (`dist.reduce_accum`, `dist.all_reduce_value`) are the functions bench.py imports and calls, and the result must
equal the single-process image of the whole range.  (The same path with the HIP library in every rank runs on the GPU
box: tests/test_gpu_bench.py.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spp_partition_is_disjoint_and_complete(rrt):
    from rust_ray_tracer_amd import dist as rdist

    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            off, n, total = rdist.spp_slice(r, world, 128)
            assert total == world * 128
            seen += list(range(off, off + n))
        assert seen == list(range(world * 128))
    parts = rdist.split_spp(1024, 8)
    assert parts == [(i * 128, 128) for i in range(8)]
    parts = rdist.split_spp(10, 4)
    assert [n for _, n in parts] == [3, 3, 2, 2] and [o for o, _ in parts] == [0, 3, 6, 8]
    with pytest.raises(ValueError):
        rdist.spp_slice(2, 2, 8)
    # the two partitions behind one call, as the benchmark uses them
    assert rdist.rank_slice(3, 8, 1024, "strong") == (384, 128, 1024)
    assert rdist.rank_slice(3, 8, 128, "weak") == (384, 128, 1024)
    assert rdist.rank_slice(0, 1, 1024, "strong") == (0, 1024, 1024)
    assert [rdist.rank_slice(r, 3, 10, "strong")[:2] for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    with pytest.raises(ValueError):
        rdist.rank_slice(0, 2, 8, "sideways")
    # outside a process group the collectives are no-ops
    assert rdist.all_reduce_value(2.5, "max") == 2.5 and rdist.reduce_accum("untouched") == "untouched"


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import helpers
    import rrt_pkg
    from oracle import oracle

    rrt = rrt_pkg.load()
    from rust_ray_tracer_amd import dist as rdist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    scene = helpers.scene_default(rrt)
    o = oracle.OracleScene(scene)
    w, h, spp = 48, 27, 3
    off, n, total = rdist.rank_slice(rank, world, spp, "weak")
    accum, rays = o.render(w, h, n, spp_offset=off, seed=4, threads=2)
    t = torch.from_numpy(accum)
    rdist.reduce_accum(t, dst=0, via_host=(rank == 0))  # both forms of the call end in the same gloo reduce on CPU tensors
    all_rays = rdist.all_reduce_value(rays, "sum")
    slowest = rdist.all_reduce_value(float(rank), "max")
    if rank == 0:
        assert slowest == world - 1
        np.savez(out_path, accum=t.numpy(), rays=np.array([int(all_rays)]), total=total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_to_the_single_process_image(rrt, oracle, tmp_path):
    import torch.multiprocessing as mp

    import helpers

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    o = oracle.OracleScene(helpers.scene_default(rrt))
    a0, r0 = o.render(48, 27, 3, spp_offset=0, seed=4, threads=2)
    a1, r1 = o.render(48, 27, 3, spp_offset=3, seed=4, threads=2)
    assert int(got["total"]) == 6 and int(got["rays"][0]) == r0 + r1
    assert np.array_equal(got["accum"], a0 + a1)  # the reduce adds the two slices, nothing else
    whole, rw = o.render(48, 27, 6, spp_offset=0, seed=4, threads=2)
    assert rw == r0 + r1
    assert np.allclose(got["accum"], whole, rtol=1e-5, atol=1e-6)  # same samples, different summation order
